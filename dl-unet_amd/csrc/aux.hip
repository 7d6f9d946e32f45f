// aux.hip — the callers and data formats either side of the hot path, all HBM-bound:
//   N2 overlap-tile front end : per-image min/max + mirror-pad + normalise     (data.py:184-188, :249-277)
//   N2 back end               : centre-crop + argmax + IoU / pixel-error counts (tester.py:29-42, functions.py:174-213)
//   N3 class-balance maps     : per-image class counts -> weight map            (functions.py:82-117)
//   N1 elastic deformation    : separable Gaussian of a uniform field, bilinear warp (data.py:225-245)
#include "common.hpp"
#include <cmath>
#include "../../include/unet_hip.h"

namespace unet {

typedef float float4_ __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float wave_min(float v) { for (int d = 32; d >= 1; d >>= 1) v = fminf(v, __shfl_xor(v, d, 64)); return v; }
__device__ __forceinline__ float wave_max(float v) { for (int d = 32; d >= 1; d >>= 1) v = fmaxf(v, __shfl_xor(v, d, 64)); return v; }

// ---- per-image min / max: one workgroup per image (images are <= a few MB) ------------------------
__global__ __launch_bounds__(1024) void minmax_kernel(const float *__restrict__ x, size_t n, float *__restrict__ out)
{
    const float *p = x + (size_t)blockIdx.x * n;
    float lo = INFINITY, hi = -INFINITY;
    for (size_t i = threadIdx.x; i < n; i += blockDim.x) { const float v = p[i]; lo = fminf(lo, v); hi = fmaxf(hi, v); }
    lo = wave_min(lo); hi = wave_max(hi);
    __shared__ float slo[16], shi[16];
    if ((threadIdx.x & 63) == 0) { slo[threadIdx.x >> 6] = lo; shi[threadIdx.x >> 6] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w) { lo = fminf(lo, slo[w]); hi = fmaxf(hi, shi[w]); }
        out[2 * blockIdx.x] = lo; out[2 * blockIdx.x + 1] = hi;
    }
}

// ---- mirror_transform (+ optional (x-min)/ptp): out[b,Y,X] = in[b, r(Y), r(X)] --------------------
// r() is the reference's map: reflect WITHOUT the edge pixel on the top/left band (row P-Y), reflect
// WITH the edge pixel on the bottom/right band (row n-1-(Y-n-P)) — data.py:266-275 is asymmetric.
__device__ __forceinline__ int mirror_index(int Y, int P, int n)
{
    if (Y < P) return P - Y;
    if (Y < P + n) return Y - P;
    return n - 1 - (Y - n - P);
}
__global__ __launch_bounds__(256) void mirror_pad_kernel(const float *__restrict__ x, float *__restrict__ out, int n, int S, int P,
                                                         const float *__restrict__ minmax, size_t total)
{
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int X = (int)(e % S);
        const size_t t = e / S;
        const int Y = (int)(t % S);
        const int b = (int)(t / S);
        float v = x[((size_t)b * n + mirror_index(Y, P, n)) * n + mirror_index(X, P, n)];
        if (minmax) { const float lo = minmax[2 * b], hi = minmax[2 * b + 1]; v = (v - lo) / (hi - lo); }
        out[e] = v;
    }
}

// ---- crop + argmax + metric counts: stats[b] = {sum(pred&label), sum(pred|label), sum|pred-label|} ----
__global__ __launch_bounds__(256) void eval_masks_kernel(const float *__restrict__ logits, long bs, long ps, long rs, int pad,
                                                         const long long *__restrict__ labels, long long *__restrict__ mask,
                                                         int n, unsigned long long *__restrict__ stats)
{
    const int b = blockIdx.y;
    const size_t npx = (size_t)n * n;
    unsigned long long inter = 0, uni = 0, diff = 0;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < npx; e += (size_t)gridDim.x * blockDim.x) {
        const int xx = (int)(e % n), yy = (int)(e / n);
        const float *p = logits + b * bs + (size_t)(yy + pad) * rs + xx + pad;
        const long long pr = p[ps] > p[0] ? 1 : 0;                       // first maximum on ties -> class 0
        mask[(size_t)b * npx + e] = pr;
        if (labels) {
            const long long lb = labels[(size_t)b * npx + e];
            inter += (pr != 0 && lb != 0); uni += (pr != 0 || lb != 0);
            diff += (unsigned long long)(pr > lb ? pr - lb : lb - pr);
        }
    }
    if (!labels) return;
    for (int d = 32; d >= 1; d >>= 1) {
        inter += __shfl_xor(inter, d, 64); uni += __shfl_xor(uni, d, 64); diff += __shfl_xor(diff, d, 64);
    }
    if ((threadIdx.x & 63) == 0) {       // integer atomics: order-independent, exact
        atomicAdd(&stats[3 * b], inter); atomicAdd(&stats[3 * b + 1], uni); atomicAdd(&stats[3 * b + 2], diff);
    }
}

// ---- class_balance: counts of label==0 / label==1 per image, then w = (v==1) ? 1 : n1/n0 ------------
__global__ __launch_bounds__(256) void count_ones_kernel(const long long *__restrict__ labels, size_t npx, unsigned long long *__restrict__ counts)
{
    const int b = blockIdx.y;
    unsigned long long c = 0;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < npx; e += (size_t)gridDim.x * blockDim.x)
        c += labels[(size_t)b * npx + e] != 0;
    for (int d = 32; d >= 1; d >>= 1) c += __shfl_xor(c, d, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(&counts[b], c);
}
__global__ __launch_bounds__(256) void class_balance_kernel(const long long *__restrict__ labels, size_t npx,
                                                            const unsigned long long *__restrict__ counts, float *__restrict__ w, size_t total)
{
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const size_t b = e / npx;
        const float n1 = (float)counts[b], n0 = (float)(npx - counts[b]);
        w[e] = labels[e] != 0 ? n1 / n1 : n1 / n0;        // counts[1]/counts[pos] as the reference computes it
    }
}

// ---- elastic deformation ----------------------------------------------------------------------------
// 1-D pass of scipy.ndimage.gaussian_filter(mode="constant", cval=0, truncate=4): zero outside the image,
// weights exp(-k^2/(2 sigma^2)) normalised over the full window; `scale` (alpha) applied on the second pass.
__global__ __launch_bounds__(256) void gauss1d_kernel(const float *__restrict__ in, float *__restrict__ out, int H, int W, int axis,
                                                      const float *__restrict__ wts, int radius, float scale, size_t total)
{
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(e % W);
        const size_t t = e / W;
        const int y = (int)(t % H);
        const size_t base = (t / H) * (size_t)H * W;
        float acc = 0.f;
        if (axis == 0) {
            for (int k = -radius; k <= radius; ++k) { const int yy = y + k; if ((unsigned)yy < (unsigned)H) acc = fmaf(wts[k + radius], in[base + (size_t)yy * W + x], acc); }
        } else {
            for (int k = -radius; k <= radius; ++k) { const int xx = x + k; if ((unsigned)xx < (unsigned)W) acc = fmaf(wts[k + radius], in[base + (size_t)y * W + xx], acc); }
        }
        out[e] = acc * scale;
    }
}
// scipy.ndimage.map_coordinates(order=1, mode="constant", cval=0): bilinear inside [0,n-1], 0 for any
// coordinate outside it.  Coordinates are (row + dy, col + dx); same field for every plane of a sample.
__global__ __launch_bounds__(256) void warp_bilinear_kernel(const float *__restrict__ img, const float *__restrict__ dy, const float *__restrict__ dx,
                                                            float *__restrict__ out, int H, int W, size_t total)
{
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(e % W);
        const size_t t = e / W;
        const int y = (int)(t % H);
        const size_t base = (t / H) * (size_t)H * W;
        const float cy = (float)y + dy[e], cx = (float)x + dx[e];
        float v = 0.f;
        if (cy >= 0.f && cy <= (float)(H - 1) && cx >= 0.f && cx <= (float)(W - 1)) {
            int y0 = (int)floorf(cy), x0 = (int)floorf(cx);
            if (y0 > H - 2) y0 = H - 2 < 0 ? 0 : H - 2;
            if (x0 > W - 2) x0 = W - 2 < 0 ? 0 : W - 2;
            const float fy = cy - (float)y0, fx = cx - (float)x0;
            const int y1 = y0 + 1 < H ? y0 + 1 : y0, x1 = x0 + 1 < W ? x0 + 1 : x0;
            const float v00 = img[base + (size_t)y0 * W + x0], v01 = img[base + (size_t)y0 * W + x1];
            const float v10 = img[base + (size_t)y1 * W + x0], v11 = img[base + (size_t)y1 * W + x1];
            v = (1.f - fy) * ((1.f - fx) * v00 + fx * v01) + fy * ((1.f - fx) * v10 + fx * v11);
        }
        out[e] = v;
    }
}


// ---- reflect-pad + rotate + centre crop (data.py:108-125) ----------------------------------------------------------------
//   image_pad = np.pad(image, input_size, mode='reflect'); image_rot = scipy.ndimage.rotate(image_pad, deg)   (order-3 spline,
//   reshape=True, mode='constant'); image = image_rot[t:b, l:r]   — the S x S centre.
// The crop only ever samples a disc of radius S/sqrt(2) around the centre of the padded image, hundreds of pixels away from
// its border, so neither the 'constant' extension of the interpolation nor the boundary initialisation of the spline
// prefilter can reach it (the prefilter's impulse response decays as 0.268^k).  Hence: (1) form the square of the padded image
// around that disc (+2 for the spline support, +28 for the prefilter) by reflect indexing, (2) turn it into cubic B-spline
// coefficients with the prefilter written as the symmetric FIR it is far from boundaries, taps sqrt(3) z^|k|, z = sqrt(3) - 2,
// |k| <= 28 (z^28 = 1e-16), separably, (3) evaluate the spline at the rotated coordinates of the crop's pixels with scipy's
// affine_transform geometry, (4) round like scipy does for integer images: (type)min(t > 0 ? t + 0.5 : 0, max).
constexpr int SPL_R = 28;
struct SplTaps { float w[2 * SPL_R + 1]; };
struct RotGeom { double c[64], s[64], offy[64], offx[64]; int t[64], l[64]; };     // per sample: rotation, affine offset, crop origin

__device__ __forceinline__ int reflect_index(int i, int n)          // numpy 'reflect': ... 2 1 | 0 1 2 ... n-1 | n-2 n-3 ...
{
    const int period = 2 * (n - 1);
    if (period == 0) return 0;
    i %= period;
    if (i < 0) i += period;
    return i < n ? i : period - i;
}

__global__ __launch_bounds__(256) void reflect_region_kernel(const float *__restrict__ img, float *__restrict__ reg, int n, int pad, int W, int y0, int x0, size_t total)
{
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int xx = (int)(e % W);
        const size_t t = e / W;
        const int yy = (int)(t % W);
        const size_t b = t / W;
        reg[e] = img[(b * n + reflect_index(y0 + yy - pad, n)) * n + reflect_index(x0 + xx - pad, n)];
    }
}

__global__ __launch_bounds__(256) void spline_fir_kernel(const float *__restrict__ in, float *__restrict__ out, int W, int axis, const SplTaps taps, size_t total)
{
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(e % W);
        const size_t t = e / W;
        const int y = (int)(t % W);
        const size_t base = (t / W) * (size_t)W * W;
        float acc = 0.f;
        const int p = axis ? x : y;
        const size_t stride = axis ? 1 : (size_t)W;
        const size_t line = axis ? base + (size_t)y * W : base + x;
        for (int k = -SPL_R; k <= SPL_R; ++k) {
            const int q = p + k;
            if ((unsigned)q < (unsigned)W) acc = fmaf(taps.w[k + SPL_R], in[line + (size_t)q * stride], acc);
        }
        out[e] = acc;
    }
}

__global__ __launch_bounds__(256) void rotate_sample_kernel(const float *__restrict__ coef, float *__restrict__ out, int S, int W, int y0, int x0,
                                                            const RotGeom g, float levels, size_t total)
{
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int j = (int)(e % S);
        const size_t t = e / S;
        const int i = (int)(t % S);
        const int b = (int)(t / S);
        // scipy: input coordinate = matrix @ output coordinate + offset, matrix = [[c, s], [-s, c]] — in fp64 like scipy: at
        // coordinates ~1000 an fp32 ulp (1e-4 px) times the slope of a noisy 8-bit image is 1e-2 grey levels, enough to move
        // 0.4 % of the pixels across a rounding boundary
        const double oy = (double)(g.t[b] + i), ox = (double)(g.l[b] + j);
        const double yd = g.c[b] * oy + g.s[b] * ox + g.offy[b] - (double)y0;
        const double xd = -g.s[b] * oy + g.c[b] * ox + g.offx[b] - (double)x0;
        const double fyd = floor(yd), fxd = floor(xd);
        const float fy = (float)fyd, fx = (float)fxd;
        const float ty = (float)(yd - fyd), tx = (float)(xd - fxd);
        float wy[4], wx[4];
        {   // cubic B-spline weights at offsets -1, 0, 1, 2 (scipy ni_interpolation.c, order 3)
            float z = 1.f - ty;
            wy[1] = (ty * ty * (ty - 2.f) * 3.f + 4.f) / 6.f; wy[2] = (z * z * (z - 2.f) * 3.f + 4.f) / 6.f; wy[0] = z * z * z / 6.f; wy[3] = 1.f - wy[0] - wy[1] - wy[2];
            z = 1.f - tx;
            wx[1] = (tx * tx * (tx - 2.f) * 3.f + 4.f) / 6.f; wx[2] = (z * z * (z - 2.f) * 3.f + 4.f) / 6.f; wx[0] = z * z * z / 6.f; wx[3] = 1.f - wx[0] - wx[1] - wx[2];
        }
        const int iy = (int)fy - 1, ix = (int)fx - 1;
        const float *cb = coef + (size_t)b * W * W;
        float acc = 0.f;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            int yy = iy + a; yy = yy < 0 ? 0 : (yy > W - 1 ? W - 1 : yy);
            float r = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                int xx = ix + q; xx = xx < 0 ? 0 : (xx > W - 1 ? W - 1 : xx);
                r = fmaf(wx[q], cb[(size_t)yy * W + xx], r);
            }
            acc = fmaf(wy[a], r, acc);
        }
        if (levels > 0.f) {
            // scipy's conversion to an unsigned integer image (ni_interpolation.c CASE_INTERP_OUT_UINT, scipy 1.15): t > 0 ? t + 0.5 : 0,
            // clamped to the type's maximum (cubic overshoot next to 255 stays 255), then truncated
            float v = acc > 0.f ? acc + 0.5f : 0.f;
            v = v > levels ? levels : v;
            acc = floorf(v);
        }
        out[e] = acc;
    }
}

static inline int grid1(size_t total, int cap = 16384)
{
    size_t g = (total + 255) / 256;
    return (int)(g < 1 ? 1 : (g > (size_t)cap ? cap : g));
}

}  // namespace unet

using namespace unet;

extern "C" {

int unet_minmax(const void *x, int B, size_t n, void *out_minmax, void *stream)
{
    ARG_CHECK(x && out_minmax && B > 0 && n > 0, "unet_minmax: bad argument");
    hipLaunchKernelGGL(minmax_kernel, dim3(B), dim3(1024), 0, (hipStream_t)stream, (const float *)x, n, (float *)out_minmax);
    HIP_TRY(hipGetLastError());
    return 0;
}

int unet_mirror_pad(const void *x, int B, int n, int S, const void *minmax, void *out, void *stream)
{
    ARG_CHECK(x && out && B > 0, "unet_mirror_pad: null argument");
    ARG_CHECK(S >= n && (S - n) % 2 == 0 && (S - n) / 2 <= n - 1, "unet_mirror_pad: cannot mirror %d into %d (pad must be even-split and < n)", n, S);
    const size_t total = (size_t)B * S * S;
    hipLaunchKernelGGL(mirror_pad_kernel, dim3(grid1(total)), dim3(256), 0, (hipStream_t)stream, (const float *)x, (float *)out, n, S, (S - n) / 2,
                       (const float *)minmax, total);
    HIP_TRY(hipGetLastError());
    return 0;
}

int unet_eval_masks(const void *logits, long batch_stride, long plane_stride, long row_stride, int pad, const void *labels_i64,
                    void *mask_i64, int B, int n, void *stats_u64, void *stream)
{
    ARG_CHECK(logits && mask_i64 && B > 0 && n > 0 && pad >= 0, "unet_eval_masks: bad argument");
    ARG_CHECK(!labels_i64 || stats_u64, "unet_eval_masks: stats buffer needed with labels");
    hipStream_t st = (hipStream_t)stream;
    if (labels_i64) HIP_TRY(hipMemsetAsync(stats_u64, 0, (size_t)B * 3 * sizeof(unsigned long long), st));
    int gx = grid1((size_t)n * n, 256);
    hipLaunchKernelGGL(eval_masks_kernel, dim3(gx, B), dim3(256), 0, st, (const float *)logits, batch_stride, plane_stride, row_stride, pad,
                       (const long long *)labels_i64, (long long *)mask_i64, n, (unsigned long long *)stats_u64);
    HIP_TRY(hipGetLastError());
    return 0;
}

int unet_class_balance(const void *labels_i64, int B, int H, int W, void *weights, void *counts_u64, void *stream)
{
    ARG_CHECK(labels_i64 && weights && counts_u64 && B > 0, "unet_class_balance: bad argument");
    hipStream_t st = (hipStream_t)stream;
    const size_t npx = (size_t)H * W;
    HIP_TRY(hipMemsetAsync(counts_u64, 0, (size_t)B * sizeof(unsigned long long), st));
    hipLaunchKernelGGL(count_ones_kernel, dim3(grid1(npx, 256), B), dim3(256), 0, st, (const long long *)labels_i64, npx, (unsigned long long *)counts_u64);
    hipLaunchKernelGGL(class_balance_kernel, dim3(grid1(npx * B)), dim3(256), 0, st, (const long long *)labels_i64, npx,
                       (const unsigned long long *)counts_u64, (float *)weights, npx * B);
    HIP_TRY(hipGetLastError());
    return 0;
}

int unet_gaussian_filter(const void *field, int B, int H, int W, const void *weights, int radius, float scale, void *tmp, void *out, void *stream)
{
    ARG_CHECK(field && weights && tmp && out && radius >= 0, "unet_gaussian_filter: bad argument");
    hipStream_t st = (hipStream_t)stream;
    const size_t total = (size_t)B * H * W;
    hipLaunchKernelGGL(gauss1d_kernel, dim3(grid1(total)), dim3(256), 0, st, (const float *)field, (float *)tmp, H, W, 0, (const float *)weights, radius, 1.f, total);
    hipLaunchKernelGGL(gauss1d_kernel, dim3(grid1(total)), dim3(256), 0, st, (const float *)tmp, (float *)out, H, W, 1, (const float *)weights, radius, scale, total);
    HIP_TRY(hipGetLastError());
    return 0;
}

int unet_warp_bilinear(const void *img, const void *dy, const void *dx, int B, int H, int W, void *out, void *stream)
{
    ARG_CHECK(img && dy && dx && out, "unet_warp_bilinear: null argument");
    const size_t total = (size_t)B * H * W;
    hipLaunchKernelGGL(warp_bilinear_kernel, dim3(grid1(total)), dim3(256), 0, (hipStream_t)stream, (const float *)img, (const float *)dy, (const float *)dx,
                       (float *)out, H, W, total);
    HIP_TRY(hipGetLastError());
    return 0;
}

static int rot_region_w(int S) { return 2 * ((int)ceil(S * 0.70711) + 2 + 2 + SPL_R) + 2; }
size_t unet_rotate_scratch_bytes(int B, int S) { return (size_t)2 * B * rot_region_w(S) * rot_region_w(S) * sizeof(float); }

int unet_reflect_rotate_crop(const void *img, int B, int n, int pad, int S, const float *angles_deg_host, int levels, void *out, void *scratch, void *stream)
{
    ARG_CHECK(img && out && scratch && angles_deg_host, "unet_reflect_rotate_crop: null argument");
    ARG_CHECK(B > 0 && B <= 64 && n >= 2 && pad >= 0 && S > 0 && S % 2 == 0, "unet_reflect_rotate_crop: bad shape (1 <= B <= 64, S even)");
    ARG_CHECK(levels == 0 || levels == 255 || levels == 65535, "unet_reflect_rotate_crop: levels must be 0 (float), 255 or 65535");
    const int N = n + 2 * pad;                                   // padded extent
    const int W = rot_region_w(S), Wh = (W - 2) / 2;
    ARG_CHECK(N >= W, "unet_reflect_rotate_crop: the padded image (%d) is smaller than the region the crop samples (%d)", N, W);
    const int c0 = (N - 1) / 2 - Wh;                             // region origin in padded coordinates (both axes)
    RotGeom g;
    for (int b = 0; b < B; ++b) {
        // scipy.ndimage.rotate: exact sines/cosines of degrees, output shape from the rotated corners, centres mapped onto each other
        const double a = angles_deg_host[b];
        double r = fmod(a, 360.0); if (r < 0) r += 360.0;
        double c, sn;
        if (fmod(r, 30.0) == 0.0) {                              // scipy.special.cosdg / sindg are exact at the reference's 30-degree steps
            static const double C30[12] = {1.0, 0.8660254037844387, 0.5, 0.0, -0.5, -0.8660254037844387, -1.0, -0.8660254037844387, -0.5, 0.0, 0.5, 0.8660254037844387};
            const int q = (int)(r / 30.0);
            c = C30[q]; sn = C30[(q + 9) % 12];                  // sin(x) = cos(x - 90)
        } else { c = cos(r * M_PI / 180.0); sn = sin(r * M_PI / 180.0); }
        const int Nout = (int)(N * (fabs(c) + fabs(sn)) + 0.5);
        const double oc = (Nout - 1) / 2.0, ic = (N - 1) / 2.0;
        g.c[b] = c; g.s[b] = sn;
        g.offy[b] = ic - (c * oc + sn * oc);
        g.offx[b] = ic - (-sn * oc + c * oc);
        g.t[b] = Nout / 2 - S / 2; g.l[b] = Nout / 2 - S / 2;
        ARG_CHECK(g.t[b] >= 0, "unet_reflect_rotate_crop: the rotated image is smaller than the crop");
    }
    SplTaps taps;
    {
        const double z = sqrt(3.0) - 2.0;
        for (int k = -SPL_R; k <= SPL_R; ++k) taps.w[k + SPL_R] = (float)(sqrt(3.0) * pow(z, abs(k)));
    }
    hipStream_t st = (hipStream_t)stream;
    float *reg = (float *)scratch, *tmp = reg + (size_t)B * W * W;
    const size_t rt = (size_t)B * W * W;
    ProfScope ps("N1.rotate");
    prof_begin(PK_ELEMWISE, "reflect_rotate_crop", st, 0.0, 0.0, 4.0 * (5.0 * rt + (double)B * S * S));
    hipLaunchKernelGGL(reflect_region_kernel, dim3(grid1(rt)), dim3(256), 0, st, (const float *)img, reg, n, pad, W, c0, c0, rt);
    hipLaunchKernelGGL(spline_fir_kernel, dim3(grid1(rt)), dim3(256), 0, st, (const float *)reg, tmp, W, 1, taps, rt);
    hipLaunchKernelGGL(spline_fir_kernel, dim3(grid1(rt)), dim3(256), 0, st, (const float *)tmp, reg, W, 0, taps, rt);
    const size_t ot = (size_t)B * S * S;
    hipLaunchKernelGGL(rotate_sample_kernel, dim3(grid1(ot)), dim3(256), 0, st, (const float *)reg, (float *)out, S, W, c0, c0, g, (float)levels, ot);
    prof_end(st);
    HIP_TRY(hipGetLastError());
    return 0;
}

}  // extern "C"
