// aux.hip — the callers and data formats either side of the hot path, all HBM-bound:
//   N2 overlap-tile front end : per-image min/max + mirror-pad + normalise     (data.py:184-188, :249-277)
//   N2 back end               : centre-crop + argmax + IoU / pixel-error counts (tester.py:29-42, functions.py:174-213)
//   N3 class-balance maps     : per-image class counts -> weight map            (functions.py:82-117)
//   N1 elastic deformation    : separable Gaussian of a uniform field, bilinear warp (data.py:225-245)
#include "common.hpp"
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

}  // extern "C"
