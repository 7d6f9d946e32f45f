// direct.hip — the HBM-bound kernels of the path (no dense contraction worth a matrix core):
//   conv11c (1 -> K stencil) fwd / weight-grad, 1x1 head fwd / bwd, 2x2 max-pool fwd / bwd,
//   weight packers, bias gradients, and the step-side kernels (BCE-with-logits, one-hot,
//   argmax, SGD-momentum).  All are written for 16 B per lane coalesced NHWC traffic.
#include "common.hpp"
#include "../../include/unet_hip.h"

namespace unet {

typedef float float4_ __attribute__((ext_vector_type(4)));
typedef unsigned short bf16_t;          // storage type of arithmetic mode 2: bf16 bit patterns

// 4 consecutive channels of a tensor stored as T (float: 16 B, bf16: 8 B), as floats
__device__ __forceinline__ float4_ load4(const float *p) { return *(const float4_ *)p; }
__device__ __forceinline__ float4_ load4(const bf16_t *p)
{
    const uint2 w = *(const uint2 *)p;
    return float4_{__builtin_bit_cast(float, w.x << 16), __builtin_bit_cast(float, w.x & 0xffff0000u),
                   __builtin_bit_cast(float, w.y << 16), __builtin_bit_cast(float, w.y & 0xffff0000u)};
}
__device__ __forceinline__ void put1(float *p, float v) { *p = v; }
__device__ __forceinline__ void put1(bf16_t *p, float v) { *p = __builtin_bit_cast(bf16_t, (__bf16)v); }
__device__ __forceinline__ void store4(float *p, float4_ v) { *(float4_ *)p = v; }
__device__ __forceinline__ void store4(bf16_t *p, float4_ v)
{
    uint2 w;
    w.x = (unsigned)__builtin_bit_cast(bf16_t, (__bf16)v[0]) | ((unsigned)__builtin_bit_cast(bf16_t, (__bf16)v[1]) << 16);
    w.y = (unsigned)__builtin_bit_cast(bf16_t, (__bf16)v[2]) | ((unsigned)__builtin_bit_cast(bf16_t, (__bf16)v[3]) << 16);
    *(uint2 *)p = w;
}

// ============================================================================================
// conv11c: x [B,S,S] (C=1) -> y [B,S-2,S-2,K] NHWC, + bias + ReLU.        network.py:23,131 (A1)
// The one true stencil layer: 4.4 FLOP/B, bound by the output write (83 MB per 572^2 tile).
// A workgroup walks whole output rows: the three input rows of a row are contiguous in memory and are staged into
// LDS once (float4 loads), the 9 taps and the bias live in registers, and CG = K/4 lanes cover a pixel's channels,
// so every store instruction writes 4 (K=64) whole pixels = 1 KiB contiguous.  No per-pixel index arithmetic beyond
// one add: the row decomposition is one scalar division per row.
// ============================================================================================
template <int K, typename T>
__global__ __launch_bounds__(256) void conv1ch_fwd_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                          const float *__restrict__ bias, T *__restrict__ y,
                                                          int S, int nrows)
{
    constexpr int CG = K / 4;                 // lanes per pixel
    constexpr int PPP = 256 / CG;             // pixels per pass per block
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_[];
    float *xs = (float *)smem_;               // [3][S]
    const int So = S - 2;
    const int cg = threadIdx.x % CG, pl = threadIdx.x / CG;
    float wr[9][4], bv[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        bv[c] = bias[cg * 4 + c];
#pragma unroll
        for (int t = 0; t < 9; ++t) wr[t][c] = w[(cg * 4 + c) * 9 + t];
    }
    const int n4 = (3 * S) >> 2;              // S % 4 == 0 (S = 16L + 60)
    for (int row = blockIdx.x; row < nrows; row += gridDim.x) {
        const int img = row / So, oy = row - img * So;
        const float4_ *xp = (const float4_ *)(x + ((size_t)img * S + oy) * S);
        __syncthreads();                      // the previous row's readers are done with xs
        for (int i = threadIdx.x; i < n4; i += 256) ((float4_ *)xs)[i] = xp[i];
        __syncthreads();
        T *yrow = y + (size_t)row * So * K + cg * 4;
        for (int ox = pl; ox < So; ox += PPP) {
            float xv[9];
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int q = 0; q < 3; ++q) xv[r * 3 + q] = xs[r * S + ox + q];
            float4_ o;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float a = bv[c];
#pragma unroll
                for (int t = 0; t < 9; ++t) a = fmaf(xv[t], wr[t][c], a);
                o[c] = a > 0.f ? a : 0.f;
            }
            store4(yrow + (size_t)ox * K, o);
        }
    }
}

// dW[k][t] = sum_p x[p+off_t] * dz[p][k],  db[k] = sum_p dz[p][k]: bound by the one read of dz.  Same row walk as the
// forward; every lane keeps 10 x 4 partial sums in registers over all its pixels, the pixel slots of a wave are then
// combined with wave shuffles and the four waves through 10 KiB of LDS: one partial vector per workgroup, reduced by
// conv1ch_wgrad_reduce_kernel in a fixed order (deterministic).
template <int K, typename T>
__global__ __launch_bounds__(256) void conv1ch_wgrad_kernel(const float *__restrict__ x, const T *__restrict__ dz,
                                                            float *__restrict__ partial, int S, int nrows, int rows_per_block)
{
    constexpr int CG = K / 4, PPP = 256 / CG;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_[];
    float *xs = (float *)smem_;               // [3][S], reused for the cross-wave reduction [4][10][K] (the launch sizes the LDS for both)
    const int So = S - 2;
    const int cg = threadIdx.x % CG, pl = threadIdx.x / CG;
    float acc[10][4];
#pragma unroll
    for (int t = 0; t < 10; ++t)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[t][c] = 0.f;
    const int n4 = (3 * S) >> 2;
    const int row0 = blockIdx.x * rows_per_block;
    int row1 = row0 + rows_per_block; row1 = row1 < nrows ? row1 : nrows;
    for (int row = row0; row < row1; ++row) {
        const int img = row / So, oy = row - img * So;
        const float4_ *xp = (const float4_ *)(x + ((size_t)img * S + oy) * S);
        __syncthreads();
        for (int i = threadIdx.x; i < n4; i += 256) ((float4_ *)xs)[i] = xp[i];
        __syncthreads();
        const T *zrow = dz + (size_t)row * So * K + cg * 4;
#pragma unroll 2
        for (int ox = pl; ox < So; ox += PPP) {
            const float4_ g = load4(zrow + (size_t)ox * K);
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const float xv = xs[r * S + ox + q];
#pragma unroll
                    for (int c = 0; c < 4; ++c) acc[r * 3 + q][c] = fmaf(xv, g[c], acc[r * 3 + q][c]);
                }
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[9][c] += g[c];
        }
    }
    // pixel slots of a wave: lanes cg, cg + CG, ... -> butterfly over the lane bits above CG
#pragma unroll
    for (int t = 0; t < 10; ++t)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float v = acc[t][c];
#pragma unroll
            for (int d = CG; d < 64; d <<= 1) v += __shfl_xor(v, d, 64);
            acc[t][c] = v;
        }
    __syncthreads();                          // xs is free
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float *red = xs;                          // [4 waves][10][K]
    if (lane < CG) {
#pragma unroll
        for (int t = 0; t < 10; ++t)
#pragma unroll
            for (int c = 0; c < 4; ++c) red[(wave * 10 + t) * K + lane * 4 + c] = acc[t][c];
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 10 * K; e += 256)
        partial[(size_t)blockIdx.x * 10 * K + e] = (red[e] + red[10 * K + e]) + (red[20 * K + e] + red[30 * K + e]);
}

// out: dw[k][t] (t<9) and db[k] from partial[nb][10][K]; one wave per output, lanes over the partials
__global__ __launch_bounds__(256) void conv1ch_wgrad_reduce_kernel(const float *__restrict__ partial, int nb, int K,
                                                                   float *__restrict__ dw, float *__restrict__ db)
{
    const int e = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (e >= 10 * K) return;
    float s = 0.f;
    for (int b = lane; b < nb; b += 64) s += partial[(size_t)b * 10 * K + e];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d, 64);
    if (lane == 0) {
        const int t = e / K, kk = e - t * K;
        if (t < 9) { if (dw) dw[kk * 9 + t] = s; } else if (db) db[kk] = s;
    }
}

// ============================================================================================
// head: finalconv 1x1, C -> 2, NHWC in, NCHW logits out.                 network.py:58,190 (A22)
// 16 lanes x float4 cover a pixel's channels (C=64); partial dot products are combined with
// wave shuffles, results staged in LDS so the two class planes are written coalesced.
// ============================================================================================
template <int C, typename T>
__global__ __launch_bounds__(256) void head1x1_fwd_kernel(const T *__restrict__ x, const float *__restrict__ w,
                                                          const float *__restrict__ bias, float *__restrict__ logits,
                                                          int B, int HW)
{
    constexpr int CG = C / 4, PPP = 256 / CG;       // pixels per pass
    constexpr int PASSES = 16;
    constexpr int PPB = PPP * PASSES;               // pixels per block
    __shared__ float outs[2][PPB];
    const int cg = threadIdx.x % CG, pl = threadIdx.x / CG;
    float4_ w0 = *(const float4_ *)(w + cg * 4), w1 = *(const float4_ *)(w + C + cg * 4);
    const float b0 = bias[0], b1 = bias[1];
    const size_t npix = (size_t)B * HW;
    const size_t base = (size_t)blockIdx.x * PPB;
#pragma unroll 4
    for (int ps = 0; ps < PASSES; ++ps) {
        const int lp = ps * PPP + pl;
        const size_t pix = base + lp;
        float s0 = 0.f, s1 = 0.f;
        if (pix < npix) {
            const float4_ v = load4(x + pix * C + cg * 4);
            s0 = v[0] * w0[0] + v[1] * w0[1] + v[2] * w0[2] + v[3] * w0[3];
            s1 = v[0] * w1[0] + v[1] * w1[1] + v[2] * w1[2] + v[3] * w1[3];
        }
#pragma unroll
        for (int d = CG / 2; d >= 1; d >>= 1) { s0 += __shfl_xor(s0, d, 64); s1 += __shfl_xor(s1, d, 64); }
        if (cg == 0) { outs[0][lp] = s0 + b0; outs[1][lp] = s1 + b1; }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 2 * PPB; e += 256) {
        const int k = e / PPB, lp = e - k * PPB;
        const size_t pix = base + lp;
        if (pix < npix) {
            const size_t img = pix / HW, rem = pix - img * HW;
            logits[(img * 2 + k) * HW + rem] = outs[k][lp];
        }
    }
}

// backward: dz[m][c] = (dl0[m]*w[0][c] + dl1[m]*w[1][c]) * (x[m][c] > 0);
//           dw[k][c] = sum_m dl_k[m]*x[m][c];  db[k] = sum_m dl_k[m]   (partials per block)
template <int C, typename T>
__global__ __launch_bounds__(256) void head1x1_bwd_kernel(const T *__restrict__ x, const float *__restrict__ w,
                                                          const float *__restrict__ dlogits, float dls, T *__restrict__ dz,
                                                          float *__restrict__ partial, int B, int HW)
{
    constexpr int CG = C / 4, PPP = 256 / CG;
    const int cg = threadIdx.x % CG, pl = threadIdx.x / CG;
    const float4_ w0 = *(const float4_ *)(w + cg * 4), w1 = *(const float4_ *)(w + C + cg * 4);
    const size_t npix = (size_t)B * HW;
    float4_ a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
    float sb0 = 0.f, sb1 = 0.f;
    // two pixels per trip: both loads are in flight before either is used (the loop is a dependent chain otherwise)
    const size_t step = (size_t)gridDim.x * PPP;
    for (size_t pix = (size_t)blockIdx.x * PPP + pl; pix < npix; pix += 2 * step) {
        const size_t pixb = pix + step;
        const bool okb = pixb < npix;
        const size_t pb = okb ? pixb : pix;
        const size_t img = pix / HW, rem = pix - img * HW;
        const size_t imgb = pb / HW, remb = pb - imgb * HW;
        const float4_ v = load4(x + pix * C + cg * 4);
        const float4_ vb = load4(x + pb * C + cg * 4);
        const float d0 = dlogits[(img * 2) * HW + rem] * dls, d1 = dlogits[(img * 2 + 1) * HW + rem] * dls;
        float d0b = dlogits[(imgb * 2) * HW + remb] * dls, d1b = dlogits[(imgb * 2 + 1) * HW + remb] * dls;
        if (!okb) { d0b = 0.f; d1b = 0.f; }
        float4_ g, gb;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            g[c] = v[c] > 0.f ? d0 * w0[c] + d1 * w1[c] : 0.f;
            a0[c] = fmaf(d0, v[c], a0[c]);
            a1[c] = fmaf(d1, v[c], a1[c]);
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            gb[c] = vb[c] > 0.f ? d0b * w0[c] + d1b * w1[c] : 0.f;
            a0[c] = fmaf(d0b, vb[c], a0[c]);
            a1[c] = fmaf(d1b, vb[c], a1[c]);
        }
        store4(dz + pix * C + cg * 4, g);
        if (okb) store4(dz + pixb * C + cg * 4, gb);
        sb0 += d0 + d0b; sb1 += d1 + d1b;
    }
    // row pitch 2C+4 floats: 16-byte aligned rows, one ds_write_b128 per class instead of four conflicting dword writes
    __shared__ __attribute__((aligned(16))) float red[PPP][2 * C + 4];
    *(float4_ *)&red[pl][cg * 4] = a0;
    *(float4_ *)&red[pl][C + cg * 4] = a1;
    if (cg == 0) { red[pl][2 * C] = sb0; red[pl][2 * C + 1] = sb1; }
    __syncthreads();
    for (int e = threadIdx.x; e < 2 * C + 2; e += 256) {
        float s = 0.f;
        for (int q = 0; q < PPP; ++q) s += red[q][e];
        partial[(size_t)blockIdx.x * (2 * C + 2) + e] = s;
    }
}

__global__ __launch_bounds__(256) void head1x1_bwd_reduce_kernel(const float *__restrict__ partial, int nb, int C,
                                                                 float *__restrict__ dw, float *__restrict__ db)
{
    const int e = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (e >= 2 * C + 2) return;
    float s = 0.f;
    for (int b = lane; b < nb; b += 64) s += partial[(size_t)b * (2 * C + 2) + e];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d, 64);
    if (lane == 0) { if (e < 2 * C) { if (dw) dw[e] = s; } else if (db) db[e - 2 * C] = s; }
}

// ============================================================================================
// 2x2 max-pool, NHWC.                                                  network.py:133-151 (A3)
// ============================================================================================
template <typename T>
__global__ __launch_bounds__(256) void maxpool2_fwd_kernel(const T *__restrict__ x, T *__restrict__ y,
                                                           int B, int H, int W, int C4)
{
    const int Ho = H >> 1, Wo = W >> 1;
    const size_t total = (size_t)B * Ho * Wo * C4;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(e % C4);
        size_t pp = e / C4;
        const int ox = (int)(pp % Wo); pp /= Wo;
        const int oy = (int)(pp % Ho);
        const int img = (int)(pp / Ho);
        const T *src = x + ((((size_t)img * H + 2 * oy) * W + 2 * ox) * C4 + c4) * 4;
        const float4_ v00 = load4(src), v01 = load4(src + 4 * C4), v10 = load4(src + (size_t)W * C4 * 4), v11 = load4(src + ((size_t)W * C4 + C4) * 4);
        float4_ m;
#pragma unroll
        for (int c = 0; c < 4; ++c) m[c] = fmaxf(fmaxf(v00[c], v01[c]), fmaxf(v10[c], v11[c]));
        store4(y + e * 4, m);             // a maximum of stored values: exact in either storage type
    }
}

// dpre = route(dy to the first maximum in row-major window order) * (pre > 0)
template <typename T>
__global__ __launch_bounds__(256) void maxpool2_bwd_kernel(const T *__restrict__ pre, const T *__restrict__ dy,
                                                           T *__restrict__ dpre, int B, int H, int W, int C4)
{
    const int Ho = H >> 1, Wo = W >> 1;
    const size_t total = (size_t)B * Ho * Wo * C4;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(e % C4);
        size_t pp = e / C4;
        const int ox = (int)(pp % Wo); pp /= Wo;
        const int oy = (int)(pp % Ho);
        const int img = (int)(pp / Ho);
        const size_t o00 = ((((size_t)img * H + 2 * oy) * W + 2 * ox) * C4 + c4) * 4;
        const size_t o01 = o00 + 4 * (size_t)C4, o10 = o00 + (size_t)W * C4 * 4, o11 = o10 + 4 * (size_t)C4;
        const float4_ v00 = load4(pre + o00), v01 = load4(pre + o01);
        const float4_ v10 = load4(pre + o10), v11 = load4(pre + o11);
        const float4_ g = load4(dy + e * 4);
        float4_ g00, g01, g10, g11;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float m = v00[c]; int mi = 0;
            if (v01[c] > m) { m = v01[c]; mi = 1; }
            if (v10[c] > m) { m = v10[c]; mi = 2; }
            if (v11[c] > m) { m = v11[c]; mi = 3; }
            const float gv = m > 0.f ? g[c] : 0.f;
            g00[c] = mi == 0 ? gv : 0.f; g01[c] = mi == 1 ? gv : 0.f;
            g10[c] = mi == 2 ? gv : 0.f; g11[c] = mi == 3 ? gv : 0.f;
        }
        store4(dpre + o00, g00); store4(dpre + o01, g01);
        store4(dpre + o10, g10); store4(dpre + o11, g11);
    }
}

// ============================================================================================
// Weight packers: reference layouts (OIHW / IOHW) -> igemm weight matrices [N][Kd].
// ============================================================================================
// conv fwd: wt[k][kd], kd = (c<C1 ? t*C1 + c : 9*C1 + t*C2 + (c-C1));  w[k][c][t]
template <typename T>
__global__ void pack_conv_fwd_kernel(const float *__restrict__ w, T *__restrict__ wt, int K, int C1, int C2)
{
    const int C = C1 + C2;
    const size_t total = (size_t)K * C * 9;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int kd = (int)(e % (9 * C));
        const int k = (int)(e / (9 * C));
        int c, t;
        if (kd < 9 * C1) { t = kd / C1; c = kd - t * C1; }
        else { const int r = kd - 9 * C1; t = r / C2; c = C1 + r - t * C2; }
        put1(wt + e, w[((size_t)k * C + c) * 9 + t]);
    }
}
// conv dgrad: wt[c][t'*K + k] = w[k][c][8 - t']
template <typename T>
__global__ void pack_conv_dgrad_kernel(const float *__restrict__ w, T *__restrict__ wt, int K, int C)
{
    const size_t total = (size_t)K * C * 9;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int kd = (int)(e % (9 * K));
        const int c = (int)(e / (9 * K));
        const int t = kd / K, k = kd - t * K;
        put1(wt + e, w[((size_t)k * C + c) * 9 + (8 - t)]);
    }
}
// up-conv fwd: wt[(ab)*Co + co][ci] = w[ci][co][ab]
template <typename T>
__global__ void pack_upconv_fwd_kernel(const float *__restrict__ w, T *__restrict__ wt, int Ci, int Co)
{
    const size_t total = (size_t)Ci * Co * 4;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int ci = (int)(e % Ci);
        const int n = (int)(e / Ci);
        const int ab = n / Co, co = n - ab * Co;
        put1(wt + e, w[((size_t)ci * Co + co) * 4 + ab]);
    }
}
// up-conv dgrad: wt[ci][(ab)*Co + co] = w[ci][co][ab]
template <typename T>
__global__ void pack_upconv_dgrad_kernel(const float *__restrict__ w, T *__restrict__ wt, int Ci, int Co)
{
    const size_t total = (size_t)Ci * Co * 4;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int kd = (int)(e % (4 * Co));
        const int ci = (int)(e / (4 * Co));
        const int ab = kd / Co, co = kd - ab * Co;
        put1(wt + e, w[((size_t)ci * Co + co) * 4 + ab]);
    }
}

static inline int grid_for(size_t total, int per_block = 256, int cap = 8192)
{
    size_t g = (total + per_block - 1) / per_block;
    return (int)(g < 1 ? 1 : (g > (size_t)cap ? cap : g));
}

// es = element size of the packed matrix: 4 (fp32) or 2 (bf16, arithmetic mode 2)
#define PACK_LAUNCH(kern, total, ...)                                                                           \
    do {                                                                                                         \
        prof_begin(PK_ELEMWISE, #kern, st, 0.0, 0.0, (4.0 + es) * (double)(total));                               \
        if (es == 2) hipLaunchKernelGGL(kern<bf16_t>, dim3(grid_for(total)), dim3(256), 0, st, w, (bf16_t *)wt, __VA_ARGS__); \
        else hipLaunchKernelGGL(kern<float>, dim3(grid_for(total)), dim3(256), 0, st, w, (float *)wt, __VA_ARGS__);          \
        prof_end(st);                                                                                            \
        HIP_TRY(hipGetLastError());                                                                              \
    } while (0)

int pack_conv_fwd(const float *w, void *wt, int K, int C1, int C2, int es, hipStream_t st)
{
    PACK_LAUNCH(pack_conv_fwd_kernel, (size_t)K * (C1 + C2) * 9, K, C1, C2);
    return 0;
}
int pack_conv_dgrad(const float *w, void *wt, int K, int C, int es, hipStream_t st)
{
    PACK_LAUNCH(pack_conv_dgrad_kernel, (size_t)K * C * 9, K, C);
    return 0;
}
int pack_upconv_fwd(const float *w, void *wt, int Ci, int Co, int es, hipStream_t st)
{
    PACK_LAUNCH(pack_upconv_fwd_kernel, (size_t)Ci * Co * 4, Ci, Co);
    return 0;
}
int pack_upconv_dgrad(const float *w, void *wt, int Ci, int Co, int es, hipStream_t st)
{
    PACK_LAUNCH(pack_upconv_dgrad_kernel, (size_t)Ci * Co * 4, Ci, Co);
    return 0;
}

// ============================================================================================
// bias gradient: db[k] = sum_m dz[m][k]  (column sums of an [M][K] matrix), two deterministic passes
// (for the up-conv bias the caller passes M*4 rows of Cout).
// ============================================================================================
constexpr int BG_ROWS = 2048;     // rows per block in pass 1
template <typename T>
__global__ __launch_bounds__(256) void bias_grad_kernel(const T *__restrict__ dz, size_t M, int K,
                                                        float *__restrict__ partial)
{
    // thread -> (channel group of 4, row slot); K/4 groups, 256/(K/4) row slots (K/4 <= 256)
    const int CG = K >> 2;
    const int slots = 256 / CG;
    const int cg = threadIdx.x % CG, sl = threadIdx.x / CG;
    const size_t r0 = (size_t)blockIdx.x * BG_ROWS;
    size_t r1 = r0 + BG_ROWS; r1 = r1 < M ? r1 : M;
    float4_ a = {0, 0, 0, 0};
    if (sl < slots)
        for (size_t r = r0 + sl; r < r1; r += slots) {
            const float4_ v = load4(dz + r * K + cg * 4);
            a[0] += v[0]; a[1] += v[1]; a[2] += v[2]; a[3] += v[3];
        }
    __shared__ float4_ red[256];
    red[threadIdx.x] = a;
    __syncthreads();
    if (threadIdx.x < CG) {
        float4_ s = {0, 0, 0, 0};
        for (int q = 0; q < slots; ++q) { const float4_ v = red[q * CG + threadIdx.x]; s[0] += v[0]; s[1] += v[1]; s[2] += v[2]; s[3] += v[3]; }
        *(float4_ *)(partial + (size_t)blockIdx.x * K + threadIdx.x * 4) = s;
    }
}
__global__ void bias_grad_reduce_kernel(const float *__restrict__ partial, int nb, int K, float *__restrict__ db)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int b = 0;
    for (; b + 4 <= nb; b += 4) {
        s0 += partial[(size_t)b * K + k]; s1 += partial[(size_t)(b + 1) * K + k];
        s2 += partial[(size_t)(b + 2) * K + k]; s3 += partial[(size_t)(b + 3) * K + k];
    }
    for (; b < nb; ++b) s0 += partial[(size_t)b * K + k];
    db[k] = (s0 + s1) + (s2 + s3);
}
size_t bias_grad_scratch_bytes(size_t M, int K) { return ((M + BG_ROWS - 1) / BG_ROWS) * K * sizeof(float); }
int bias_grad(const void *dz, size_t M, int K, float *db, float *scratch, int es, hipStream_t st)
{
    ARG_CHECK(K % 4 == 0 && K / 4 <= 256, "bias_grad: K=%d unsupported", K);
    const int nb = (int)((M + BG_ROWS - 1) / BG_ROWS);
    prof_begin(PK_ELEMWISE, "bias_grad", st, (double)M * K, 0.0, (double)es * M * K);
    if (es == 2) hipLaunchKernelGGL(bias_grad_kernel<bf16_t>, dim3(nb), dim3(256), 0, st, (const bf16_t *)dz, M, K, scratch);
    else hipLaunchKernelGGL(bias_grad_kernel<float>, dim3(nb), dim3(256), 0, st, (const float *)dz, M, K, scratch);
    hipLaunchKernelGGL(bias_grad_reduce_kernel, dim3(cdiv(K, 256)), dim3(256), 0, st, scratch, nb, K, db);
    prof_end(st);
    HIP_TRY(hipGetLastError());
    return 0;
}

// ============================================================================================
// Step-side kernels (L1-L3)
// ============================================================================================
constexpr int BCE_PER_BLOCK = 4096;
__global__ __launch_bounds__(256) void bce_logits_kernel(const float *__restrict__ x, const float *__restrict__ z,
                                                         const float *__restrict__ w, long wsB, long wsC, long wsH, long wsW,
                                                         int H, int W, size_t n, float *__restrict__ dx, float gscale,
                                                         double *__restrict__ partial)
{
    const size_t b0 = (size_t)blockIdx.x * BCE_PER_BLOCK;
    double acc = 0.0;
    const float inv_n = (float)(1.0 / (double)n);
    for (int i = threadIdx.x; i < BCE_PER_BLOCK; i += 256) {
        const size_t e = b0 + i;
        if (e >= n) break;
        const float xv = x[e], zv = z[e];
        float wv = 1.f;
        if (w) {
            size_t r = e;
            const int xx = (int)(r % W); r /= W;
            const int yy = (int)(r % H); r /= H;
            const int cc = (int)(r % 2);
            const long bb = (long)(r / 2);
            wv = w[bb * wsB + cc * wsC + yy * wsH + xx * wsW];
        }
        const float ax = fabsf(xv);
        const float l = fmaxf(xv, 0.f) - xv * zv + log1pf(expf(-ax));
        acc += (double)(wv * l);
        if (dx) {
            const float ex = expf(-ax);
            const float sg = xv >= 0.f ? 1.f / (1.f + ex) : ex / (1.f + ex);
            dx[e] = wv * (sg - zv) * inv_n * gscale;
        }
    }
    __shared__ double red[256];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}
__global__ __launch_bounds__(256) void bce_final_kernel(const double *__restrict__ partial, int nb, size_t n, float *loss)
{
    __shared__ double red[256];
    double a = 0.0;
    for (int i = threadIdx.x; i < nb; i += 256) a += partial[i];
    red[threadIdx.x] = a;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) *loss = (float)(red[0] / (double)n);
}

// L1 + L2 in one pass over the logits (trainer.py:60-82): the one-hot target [1-y, y] is never materialised, the loss
// gradient and the argmax mask come out of the same read.  A thread owns a pixel (both class planes): 2 x 4 B logits + 8 B
// label in, 2 x 4 B dlogits + 8 B mask out.  Logits / weight are addressed through strides (the trainer's centre crop of
// preds is a view); dlogits and the mask are dense.  Loss partials per block in double, fixed-order finish in bce_final_kernel.
constexpr int BCE_PX_PER_BLOCK = 2048;
__global__ __launch_bounds__(256) void bce_step_kernel(const float *__restrict__ x, long xsB, long xsC, long xsH,
                                                       const long long *__restrict__ labels,
                                                       const float *__restrict__ w, long wsB, long wsC, long wsH, long wsW,
                                                       int H, int W, size_t npix, float *__restrict__ dx, float gscale,
                                                       long long *__restrict__ mask, double *__restrict__ partial)
{
    const size_t b0 = (size_t)blockIdx.x * BCE_PX_PER_BLOCK;
    double acc = 0.0;
    const float inv_n = (float)(1.0 / (double)(2 * npix));
    const size_t HW = (size_t)H * W;
    for (int i = threadIdx.x; i < BCE_PX_PER_BLOCK; i += 256) {
        const size_t e = b0 + i;
        if (e >= npix) break;
        const size_t bb = e / HW, rem = e - bb * HW;
        const int yy = (int)(rem / W), xx = (int)(rem - (size_t)yy * W);
        const float *px = x + bb * xsB + (long)yy * xsH + xx;
        const float x0 = px[0], x1 = px[xsC];
        const float y = (float)labels[e];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const float xv = c ? x1 : x0, zv = c ? y : 1.f - y;
            const float wv = w ? w[bb * wsB + c * wsC + yy * wsH + xx * wsW] : 1.f;
            const float ax = fabsf(xv);
            const float ex = expf(-ax);
            const float l = fmaxf(xv, 0.f) - xv * zv + log1pf(ex);
            acc += (double)(wv * l);
            if (dx) {
                const float sg = xv >= 0.f ? 1.f / (1.f + ex) : ex / (1.f + ex);
                dx[(bb * 2 + c) * HW + rem] = wv * (sg - zv) * inv_n * gscale;
            }
        }
        if (mask) mask[e] = x1 > x0 ? 1 : 0;              // first maximum wins on ties -> class 0
    }
    __shared__ double red[256];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s2 = 128; s2 > 0; s2 >>= 1) {
        if (threadIdx.x < s2) red[threadIdx.x] += red[threadIdx.x + s2];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

__global__ void onehot2_kernel(const long long *__restrict__ labels, float *__restrict__ t, int B, size_t HW)
{
    const size_t total = (size_t)B * HW;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const size_t img = e / HW, rem = e - img * HW;
        const float y = (float)labels[e];
        t[(img * 2) * HW + rem] = 1.f - y;
        t[(img * 2 + 1) * HW + rem] = y;
    }
}

__global__ void argmax2_kernel(const float *__restrict__ x, long bs, long ps, long rs, long long *__restrict__ out,
                               int B, int H, int W)
{
    const size_t total = (size_t)B * H * W;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        size_t r = e;
        const int xx = (int)(r % W); r /= W;
        const int yy = (int)(r % H);
        const long b = (long)(r / H);
        const float *p = x + b * bs + yy * rs + xx;
        out[e] = p[ps] > p[0] ? 1 : 0;            // first maximum wins on ties -> class 0
    }
}

// L3: buf = first ? g : mu*buf + g ; p -= lr*buf over <= 46 tensors in one launch.  HBM-bound (3 reads + 2 writes per
// element): a workgroup takes 4096-element chunks of one tensor (16 B per lane per access; the tensors' tails and any
// tensor whose pointers are not 16-byte aligned take 4-byte accesses).  The multiply and the add are rounded separately,
// like the reference's buf.mul_(mu).add_(g); p.add_(buf, alpha=-lr).
constexpr int SGD_CHUNK = 4096;
struct SgdTable { float *p[UNET_N_PARAMS]; const float *g[UNET_N_PARAMS]; float *b[UNET_N_PARAMS];
                  unsigned long long numel[UNET_N_PARAMS]; unsigned cstart[UNET_N_PARAMS + 1]; int n; };
template <bool VEC>
__global__ __launch_bounds__(256) void sgd_momentum_kernel(const SgdTable tb, float lr, float mu, int first)
{
    const unsigned nchunks = tb.cstart[tb.n];
    for (unsigned chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
        int t = 0;
        while (tb.cstart[t + 1] <= chunk) ++t;                      // scalar: chunk is uniform
        const unsigned long long n = tb.numel[t];
        const unsigned long long off = (unsigned long long)(chunk - tb.cstart[t]) * SGD_CHUNK;
        float *pp = tb.p[t] + off; const float *gp = tb.g[t] + off; float *bp = tb.b[t] + off;
        const unsigned long long left = n - off;
        if (VEC && left >= SGD_CHUNK) {
            float4_ g[4], b[4], q[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) g[i] = ((const float4_ *)gp)[i * 256 + threadIdx.x];
#pragma unroll
            for (int i = 0; i < 4; ++i) q[i] = ((const float4_ *)pp)[i * 256 + threadIdx.x];
            if (!first) {
#pragma unroll
                for (int i = 0; i < 4; ++i) b[i] = ((const float4_ *)bp)[i * 256 + threadIdx.x];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float bn = first ? g[i][c] : __fadd_rn(__fmul_rn(mu, b[i][c]), g[i][c]);
                    b[i][c] = bn;
                    q[i][c] = __fsub_rn(q[i][c], __fmul_rn(lr, bn));
                }
                ((float4_ *)bp)[i * 256 + threadIdx.x] = b[i];
                ((float4_ *)pp)[i * 256 + threadIdx.x] = q[i];
            }
        } else {
            const int m = left < SGD_CHUNK ? (int)left : SGD_CHUNK;
            for (int i = threadIdx.x; i < m; i += 256) {
                const float g = gp[i];
                const float bn = first ? g : __fadd_rn(__fmul_rn(mu, bp[i]), g);
                bp[i] = bn;
                pp[i] = __fsub_rn(pp[i], __fmul_rn(lr, bn));
            }
        }
    }
}

// ---- launches of the kernels above; es = element size of the activation tensors (4: fp32, 2: bf16) -------------------------
int conv1ch_fwd(const float *x, int B, int S, const float *w, const float *bias, int K, void *y, int es, hipStream_t st)
{
    ARG_CHECK(K == 64 || K == 32, "conv1ch: K=%d unsupported (32 or 64)", K);
    ARG_CHECK(S >= 3 && S % 4 == 0, "conv1ch: S=%d must be a multiple of 4", S);
    const int So = S - 2, nrows = B * So;
    const int grid = nrows < 2048 ? nrows : 2048;
    const size_t lds = (size_t)3 * S * sizeof(float);
    prof_begin(PK_STENCIL, "conv1ch_fwd", st, 18.0 * nrows * So * K, 0.0, 4.0 * B * S * S + (double)es * nrows * So * K);
    if (es == 2) {
        if (K == 64) hipLaunchKernelGGL((conv1ch_fwd_kernel<64, bf16_t>), dim3(grid), dim3(256), lds, st, x, w, bias, (bf16_t *)y, S, nrows);
        else hipLaunchKernelGGL((conv1ch_fwd_kernel<32, bf16_t>), dim3(grid), dim3(256), lds, st, x, w, bias, (bf16_t *)y, S, nrows);
    } else {
        if (K == 64) hipLaunchKernelGGL((conv1ch_fwd_kernel<64, float>), dim3(grid), dim3(256), lds, st, x, w, bias, (float *)y, S, nrows);
        else hipLaunchKernelGGL((conv1ch_fwd_kernel<32, float>), dim3(grid), dim3(256), lds, st, x, w, bias, (float *)y, S, nrows);
    }
    prof_end(st);
    HIP_TRY(hipGetLastError());
    return 0;
}

// rows per workgroup and workgroup count of the weight-gradient pass (<= 1024 partial vectors)
// conv11c input gradient (only a caller that asks for d loss / d image needs it; the reference's training never does):
//   dx[b][y][x] = sum_{k,ty,tx} dz[b][y - ty][x - tx][k] w[k][ty][tx]      (full correlation with the flipped filter)
// One thread per input pixel, the 9 x K filter taps in LDS, dz read with 16-byte (fp32) / 8-byte (bf16) accesses; neighbouring
// pixels re-read each other's dz rows through L1 / L2.  Not on any timed path.
template <int K, typename T>
__global__ __launch_bounds__(256) void conv1ch_dgrad_kernel(const T *__restrict__ dz, const float *__restrict__ w, float *__restrict__ dx, int B, int S)
{
    __shared__ float ws[9][K];
    for (int i = threadIdx.x; i < 9 * K; i += 256) { const int k = i / 9, t = i - 9 * k; ws[t][k] = w[i]; }
    __syncthreads();
    const int So = S - 2;
    const size_t total = (size_t)B * S * S;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const int x = (int)(e % S);
        const int y = (int)((e / S) % S);
        const int b = (int)(e / ((size_t)S * S));
        float acc = 0.f;
#pragma unroll
        for (int ty = 0; ty < 3; ++ty) {
            const int oy = y - ty;
            if ((unsigned)oy >= (unsigned)So) continue;
#pragma unroll
            for (int tx = 0; tx < 3; ++tx) {
                const int ox = x - tx;
                if ((unsigned)ox >= (unsigned)So) continue;
                const T *src = dz + (((size_t)b * So + oy) * So + ox) * K;
                const float *wt = ws[ty * 3 + tx];
#pragma unroll
                for (int k = 0; k < K; k += 4) {
                    const float4_ v = load4(src + k);
                    acc += v[0] * wt[k] + v[1] * wt[k + 1] + v[2] * wt[k + 2] + v[3] * wt[k + 3];
                }
            }
        }
        dx[e] = acc;
    }
}

int conv1ch_dgrad(const void *dz, int B, int S, int K, const float *w, float *dx, int es, hipStream_t st)
{
    ARG_CHECK(K == 64 || K == 32, "conv1ch: K=%d unsupported (32 or 64)", K);
    const size_t total = (size_t)B * S * S;
    const int nb = (int)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
    prof_begin(PK_STENCIL, "conv1ch_dgrad", st, 18.0 * (double)B * (S - 2) * (S - 2) * K, 0.0, 4.0 * (double)total + (double)es * B * (S - 2) * (S - 2) * K);
    if (es == 2) {
        if (K == 64) hipLaunchKernelGGL((conv1ch_dgrad_kernel<64, bf16_t>), dim3(nb), dim3(256), 0, st, (const bf16_t *)dz, w, dx, B, S);
        else hipLaunchKernelGGL((conv1ch_dgrad_kernel<32, bf16_t>), dim3(nb), dim3(256), 0, st, (const bf16_t *)dz, w, dx, B, S);
    } else {
        if (K == 64) hipLaunchKernelGGL((conv1ch_dgrad_kernel<64, float>), dim3(nb), dim3(256), 0, st, (const float *)dz, w, dx, B, S);
        else hipLaunchKernelGGL((conv1ch_dgrad_kernel<32, float>), dim3(nb), dim3(256), 0, st, (const float *)dz, w, dx, B, S);
    }
    prof_end(st);
    HIP_TRY(hipGetLastError());
    return 0;
}

static void conv1ch_bwd_split(int B, int S, int &rpb, int &nb)
{
    const int nrows = B * (S - 2);
    rpb = cdiv(nrows, 1024);
    nb = cdiv(nrows, rpb);
}
int conv1ch_bwd(const float *x, int B, int S, int K, const void *dz, float *dw, float *db, float *scratch, int es, hipStream_t st)
{
    ARG_CHECK(K == 64 || K == 32, "conv1ch: K=%d unsupported (32 or 64)", K);
    ARG_CHECK(S >= 3 && S % 4 == 0, "conv1ch: S=%d must be a multiple of 4", S);
    int rpb, nb;
    conv1ch_bwd_split(B, S, rpb, nb);
    const int So = S - 2, nrows = B * So;
    size_t lds = (size_t)3 * S * sizeof(float);
    if (lds < (size_t)40 * K * sizeof(float)) lds = (size_t)40 * K * sizeof(float);
    prof_begin(PK_STENCIL, "conv1ch_wgrad", st, 20.0 * nrows * So * K, 0.0, 4.0 * B * S * S + (double)es * nrows * So * K);
    if (es == 2) {
        if (K == 64) hipLaunchKernelGGL((conv1ch_wgrad_kernel<64, bf16_t>), dim3(nb), dim3(256), lds, st, x, (const bf16_t *)dz, scratch, S, nrows, rpb);
        else hipLaunchKernelGGL((conv1ch_wgrad_kernel<32, bf16_t>), dim3(nb), dim3(256), lds, st, x, (const bf16_t *)dz, scratch, S, nrows, rpb);
    } else {
        if (K == 64) hipLaunchKernelGGL((conv1ch_wgrad_kernel<64, float>), dim3(nb), dim3(256), lds, st, x, (const float *)dz, scratch, S, nrows, rpb);
        else hipLaunchKernelGGL((conv1ch_wgrad_kernel<32, float>), dim3(nb), dim3(256), lds, st, x, (const float *)dz, scratch, S, nrows, rpb);
    }
    hipLaunchKernelGGL(conv1ch_wgrad_reduce_kernel, dim3(cdiv(10 * K, 4)), dim3(256), 0, st, (const float *)scratch, nb, K, dw, db);
    prof_end(st);
    HIP_TRY(hipGetLastError());
    return 0;
}

int head1x1_fwd(const void *x, int B, int H, int W, int C, const float *w, const float *bias, float *logits, int es, hipStream_t st)
{
    ARG_CHECK(C == 64 || C == 32, "head1x1: C=%d unsupported (32 or 64)", C);
    const size_t npix = (size_t)B * H * W;
    const int ppb = (256 / (C / 4)) * 16;
    const int grid = (int)((npix + ppb - 1) / ppb);
    prof_begin(PK_ELEMWISE, "head1x1_fwd", st, 4.0 * npix * C, 0.0, (double)npix * (es * C + 8));
    if (es == 2) {
        if (C == 64) hipLaunchKernelGGL((head1x1_fwd_kernel<64, bf16_t>), dim3(grid), dim3(256), 0, st, (const bf16_t *)x, w, bias, logits, B, H * W);
        else hipLaunchKernelGGL((head1x1_fwd_kernel<32, bf16_t>), dim3(grid), dim3(256), 0, st, (const bf16_t *)x, w, bias, logits, B, H * W);
    } else {
        if (C == 64) hipLaunchKernelGGL((head1x1_fwd_kernel<64, float>), dim3(grid), dim3(256), 0, st, (const float *)x, w, bias, logits, B, H * W);
        else hipLaunchKernelGGL((head1x1_fwd_kernel<32, float>), dim3(grid), dim3(256), 0, st, (const float *)x, w, bias, logits, B, H * W);
    }
    prof_end(st);
    HIP_TRY(hipGetLastError());
    return 0;
}

static int head_bwd_blocks(int B, int H, int W) { return grid_for((size_t)B * H * W, 16 * 32, 2048); }
int head1x1_bwd(const void *x, int B, int H, int W, int C, const float *w, const float *dlogits, float dl_scale, void *dz, float *dw, float *db,
                float *scratch, int es, hipStream_t st)
{
    ARG_CHECK(C == 64 || C == 32, "head1x1: C=%d unsupported (32 or 64)", C);
    const int nb = head_bwd_blocks(B, H, W);
    const double npix = (double)B * H * W;
    prof_begin(PK_ELEMWISE, "head1x1_bwd", st, 8.0 * npix * C, 0.0, npix * (2.0 * es * C + 8));
    if (es == 2) {
        if (C == 64) hipLaunchKernelGGL((head1x1_bwd_kernel<64, bf16_t>), dim3(nb), dim3(256), 0, st, (const bf16_t *)x, w, dlogits, dl_scale, (bf16_t *)dz, scratch, B, H * W);
        else hipLaunchKernelGGL((head1x1_bwd_kernel<32, bf16_t>), dim3(nb), dim3(256), 0, st, (const bf16_t *)x, w, dlogits, dl_scale, (bf16_t *)dz, scratch, B, H * W);
    } else {
        if (C == 64) hipLaunchKernelGGL((head1x1_bwd_kernel<64, float>), dim3(nb), dim3(256), 0, st, (const float *)x, w, dlogits, dl_scale, (float *)dz, scratch, B, H * W);
        else hipLaunchKernelGGL((head1x1_bwd_kernel<32, float>), dim3(nb), dim3(256), 0, st, (const float *)x, w, dlogits, dl_scale, (float *)dz, scratch, B, H * W);
    }
    hipLaunchKernelGGL(head1x1_bwd_reduce_kernel, dim3(cdiv(2 * C + 2, 4)), dim3(256), 0, st, (const float *)scratch, nb, C, dw, db);
    prof_end(st);
    HIP_TRY(hipGetLastError());
    return 0;
}

int maxpool2_fwd(const void *x, void *y, int B, int H, int W, int C, int es, hipStream_t st)
{
    ARG_CHECK(H % 2 == 0 && W % 2 == 0 && C % 4 == 0, "maxpool2: H,W must be even and C a multiple of 4");
    const size_t total = (size_t)B * (H / 2) * (W / 2) * (C / 4);
    prof_begin(PK_ELEMWISE, "maxpool2_fwd", st, 0.0, 0.0, 4.0 * es * (double)total * 5.0);
    if (es == 2) hipLaunchKernelGGL(maxpool2_fwd_kernel<bf16_t>, dim3(grid_for(total, 256, 65536)), dim3(256), 0, st, (const bf16_t *)x, (bf16_t *)y, B, H, W, C / 4);
    else hipLaunchKernelGGL(maxpool2_fwd_kernel<float>, dim3(grid_for(total, 256, 65536)), dim3(256), 0, st, (const float *)x, (float *)y, B, H, W, C / 4);
    prof_end(st);
    HIP_TRY(hipGetLastError());
    return 0;
}
int maxpool2_bwd(const void *pre, const void *dy, void *dpre, int B, int H, int W, int C, int es, hipStream_t st)
{
    ARG_CHECK(H % 2 == 0 && W % 2 == 0 && C % 4 == 0, "maxpool2: H,W must be even and C a multiple of 4");
    const size_t total = (size_t)B * (H / 2) * (W / 2) * (C / 4);
    prof_begin(PK_ELEMWISE, "maxpool2_bwd", st, 0.0, 0.0, 4.0 * es * (double)total * 9.0);
    if (es == 2) hipLaunchKernelGGL(maxpool2_bwd_kernel<bf16_t>, dim3(grid_for(total, 256, 65536)), dim3(256), 0, st, (const bf16_t *)pre, (const bf16_t *)dy, (bf16_t *)dpre, B, H, W, C / 4);
    else hipLaunchKernelGGL(maxpool2_bwd_kernel<float>, dim3(grid_for(total, 256, 65536)), dim3(256), 0, st, (const float *)pre, (const float *)dy, (float *)dpre, B, H, W, C / 4);
    prof_end(st);
    HIP_TRY(hipGetLastError());
    return 0;
}

}  // namespace unet

using namespace unet;

extern "C" {

// element size of the activation tensors the per-op entry points take: bf16 in arithmetic mode 2, else fp32
static int op_es() { return get_math_mode() == 2 ? 2 : 4; }

int unet_conv1ch_fwd(const void *x, int B, int S, const void *w, const void *bias, int K, void *y, void *stream)
{
    return conv1ch_fwd((const float *)x, B, S, (const float *)w, (const float *)bias, K, y, op_es(), (hipStream_t)stream);
}
size_t unet_conv1ch_bwd_scratch_bytes(int B, int S, int K) { int rpb, nb; conv1ch_bwd_split(B, S, rpb, nb); return (size_t)nb * 10 * K * sizeof(float); }
int unet_conv1ch_bwd(const void *x, int B, int S, int K, const void *dz, void *dw, void *db, void *scratch, void *stream)
{
    return conv1ch_bwd((const float *)x, B, S, K, dz, (float *)dw, (float *)db, (float *)scratch, op_es(), (hipStream_t)stream);
}
int unet_head1x1_fwd(const void *x, int B, int H, int W, int C, const void *w, const void *bias, void *logits, void *stream)
{
    return head1x1_fwd(x, B, H, W, C, (const float *)w, (const float *)bias, (float *)logits, op_es(), (hipStream_t)stream);
}
size_t unet_head1x1_bwd_scratch_bytes(int B, int H, int W, int C) { return (size_t)head_bwd_blocks(B, H, W) * (2 * C + 2) * sizeof(float); }
int unet_head1x1_bwd(const void *x, int B, int H, int W, int C, const void *w, const void *dlogits, void *dz,
                     void *dw, void *db, void *scratch, void *stream)
{
    return head1x1_bwd(x, B, H, W, C, (const float *)w, (const float *)dlogits, 1.0f, dz, (float *)dw, (float *)db, (float *)scratch, op_es(), (hipStream_t)stream);
}
int unet_maxpool2_fwd(const void *x, void *y, int B, int H, int W, int C, void *stream)
{
    return maxpool2_fwd(x, y, B, H, W, C, op_es(), (hipStream_t)stream);
}
int unet_maxpool2_bwd(const void *pre, const void *dy, void *dpre, int B, int H, int W, int C, void *stream)
{
    return maxpool2_bwd(pre, dy, dpre, B, H, W, C, op_es(), (hipStream_t)stream);
}

size_t unet_bce_scratch_bytes(size_t numel) { return ((numel + BCE_PER_BLOCK - 1) / BCE_PER_BLOCK) * sizeof(double); }
int unet_bce_logits(const void *logits, const void *target, const void *weight, long wsB, long wsC, long wsH, long wsW,
                    int B, int H, int W, void *loss_out, void *dlogits, float grad_scale, void *scratch, void *stream)
{
    const size_t n = (size_t)B * 2 * H * W;
    ARG_CHECK(n > 0 && loss_out && scratch, "bce: bad arguments");
    const int nb = (int)((n + BCE_PER_BLOCK - 1) / BCE_PER_BLOCK);
    hipStream_t st = (hipStream_t)stream;
    ProfScope ps("L1.bce");
    prof_begin(PK_ELEMWISE, "bce_logits", st, 0.0, 0.0, 4.0 * (double)n * (2 + (weight ? 1 : 0) + (dlogits ? 1 : 0)));
    hipLaunchKernelGGL(bce_logits_kernel, dim3(nb), dim3(256), 0, st, (const float *)logits, (const float *)target, (const float *)weight,
                       wsB, wsC, wsH, wsW, H, W, n, (float *)dlogits, grad_scale, (double *)scratch);
    hipLaunchKernelGGL(bce_final_kernel, dim3(1), dim3(256), 0, st, (const double *)scratch, nb, n, (float *)loss_out);
    prof_end(st);
    HIP_TRY(hipGetLastError());
    return 0;
}
size_t unet_bce_step_scratch_bytes(size_t npix) { return ((npix + BCE_PX_PER_BLOCK - 1) / BCE_PX_PER_BLOCK) * sizeof(double); }
int unet_bce_step(const void *logits, long xsB, long xsC, long xsH, const void *labels_i64, const void *weight, long wsB, long wsC,
                  long wsH, long wsW, int B, int H, int W, void *loss_out, void *dlogits, float grad_scale, void *mask_i64, void *scratch,
                  void *stream)
{
    const size_t npix = (size_t)B * H * W;
    ARG_CHECK(npix > 0 && logits && labels_i64 && loss_out && scratch, "bce_step: bad arguments");
    const int nb = (int)((npix + BCE_PX_PER_BLOCK - 1) / BCE_PX_PER_BLOCK);
    hipStream_t st = (hipStream_t)stream;
    ProfScope ps("L1.bce+L2.argmax");
    // logits 8 B + label 8 B in, dlogits 8 B + mask 8 B out per pixel
    prof_begin(PK_ELEMWISE, "bce_step", st, 24.0 * npix, 0.0, (double)npix * (16.0 + (dlogits ? 8.0 : 0.0) + (mask_i64 ? 8.0 : 0.0) + (weight ? 8.0 : 0.0)));
    hipLaunchKernelGGL(bce_step_kernel, dim3(nb), dim3(256), 0, st, (const float *)logits, xsB, xsC, xsH, (const long long *)labels_i64,
                       (const float *)weight, wsB, wsC, wsH, wsW, H, W, npix, (float *)dlogits, grad_scale, (long long *)mask_i64, (double *)scratch);
    hipLaunchKernelGGL(bce_final_kernel, dim3(1), dim3(256), 0, st, (const double *)scratch, nb, 2 * npix, (float *)loss_out);
    prof_end(st);
    HIP_TRY(hipGetLastError());
    return 0;
}

int unet_onehot2(const void *labels_i64, void *target, int B, int H, int W, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    ProfScope ps("L1.onehot");
    prof_begin(PK_ELEMWISE, "onehot2", st, 0.0, 0.0, 16.0 * (double)B * H * W);
    hipLaunchKernelGGL(onehot2_kernel, dim3(grid_for((size_t)B * H * W)), dim3(256), 0, st, (const long long *)labels_i64, (float *)target, B, (size_t)H * W);
    prof_end(st);
    HIP_TRY(hipGetLastError());
    return 0;
}
int unet_argmax2(const void *logits, long batch_stride, long plane_stride, long row_stride, void *out_i64, int B, int H, int W, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    ProfScope ps("L2.argmax");
    prof_begin(PK_ELEMWISE, "argmax2", st, 0.0, 0.0, 16.0 * (double)B * H * W);
    hipLaunchKernelGGL(argmax2_kernel, dim3(grid_for((size_t)B * H * W)), dim3(256), 0, st, (const float *)logits, batch_stride, plane_stride, row_stride, (long long *)out_i64, B, H, W);
    prof_end(st);
    HIP_TRY(hipGetLastError());
    return 0;
}
int unet_sgd_momentum(void *const *params, const void *const *grads, void *const *bufs, const size_t *numel, int n,
                      float lr, float mu, int first_step, void *stream)
{
    ARG_CHECK(n > 0 && n <= UNET_N_PARAMS, "sgd: n=%d out of range (1..%d)", n, UNET_N_PARAMS);
    SgdTable tb;
    tb.n = n;
    unsigned chunks = 0;
    unsigned long long total = 0;
    bool vec = true;
    for (int i = 0; i < n; ++i) {
        tb.p[i] = (float *)params[i]; tb.g[i] = (const float *)grads[i]; tb.b[i] = (float *)bufs[i];
        if (((uintptr_t)params[i] | (uintptr_t)grads[i] | (uintptr_t)bufs[i]) & 15) vec = false;
        tb.numel[i] = numel[i];
        tb.cstart[i] = chunks;
        chunks += (unsigned)((numel[i] + SGD_CHUNK - 1) / SGD_CHUNK);
        total += numel[i];
    }
    tb.cstart[n] = chunks;
    if (chunks == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    const int grid = chunks < 16384u ? (int)chunks : 16384;
    ProfScope ps("L3.sgd");
    prof_begin(PK_ELEMWISE, "sgd_momentum", st, 4.0 * (double)total, 0.0, 4.0 * (double)total * (first_step ? 4 : 5));
    if (vec) hipLaunchKernelGGL(sgd_momentum_kernel<true>, dim3(grid), dim3(256), 0, st, tb, lr, mu, first_step);
    else hipLaunchKernelGGL(sgd_momentum_kernel<false>, dim3(grid), dim3(256), 0, st, tb, lr, mu, first_step);
    prof_end(st);
    HIP_TRY(hipGetLastError());
    return 0;
}

}  // extern "C"
