// wgrad.hip — weight gradients on the fp32 matrix cores (autograd backward of the convs, A23).
//
//   D[t][i][j] = sum_{img,y,x} X[img][(y+oy0)*s + ty - xpad][(x+ox0)*s + tx - xpad][xc0+i] * Y[img][y][x][yc0+j]
//
// conv3x3  : X = layer input (NHWC, optionally the zero-padded skip tensor), Y = dz, s=1, 3x3 taps
//            -> dW[k=j][c=i][ty][tx]                                     (network.py:23-56 weights)
// up-conv  : X = dOut (stride-2 gather, 2x2 taps), Y = layer input, roles transposed
//            -> dW[ci=j][co=i][a][b]                                     (network.py:38-53 weights)
//
// The reduction runs over pixels, the output is tiny, so the kernel is split-K: one workgroup owns
// a 64(i) x 64(j) channel tile for ALL taps and a (image, row-chunk, column-strip) of pixels.  Per
// output row it stages one Y row-strip and the s new X rows of the halo ring into LDS with
// global_load_lds (the 3x3 taps then re-read the same LDS rows: X is fetched once, not 9x), and every
// wave accumulates T 32x32 MFMA tiles (v_mfma_f32_32x32x2_f32, K = a pair of pixels).  Partial tiles
// go to a slab and a second kernel reduces them in a fixed order (deterministic, no atomics).
#include "common.hpp"
#include <array>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <mutex>
#include <type_traits>

namespace unet {

bool wgradw_applicable(const WgradP &p);
size_t wgradw_slab_need(const WgradP &p);
int launch_wgradw(const WgradP &p, hipStream_t st);

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define GLDS16(gptr, lptr)                                                                    \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gptr),  \
                                     (__attribute__((address_space(3))) void *)(lptr), 16, 0, 0)

constexpr int PWMAX = 32;     // pixels per strip (MFMA K = pixel pairs)

__device__ __forceinline__ int fdiv_u(int n, const FastDiv &f) { return (int)(((unsigned long long)(unsigned)n * f.mul) >> f.shift); }

struct WgradK {               // kernel-side copy with the derived decomposition
    WgradP p;
    int pw, nstrips, rows_per_chunk, nchunks, ntile_i, ntile_j;
    int xbytes, ybytes;                // buffer-descriptor sizes of X and Y (bytes; the buffer paths need both below 2 GiB)
    int ci_real, cj_real;              // channel counts of the tensors; p.Ci / p.Cj are these rounded up to whole 64-channel tiles
    int nparts, ngroups;               // pixel partitions, and workgroups per channel tile that share them
    size_t pstride;                    // floats per partition in the slab: T*Ci*Cj weights + Cj bias partials
    // pixel-linear up-conv form (wgrad_up_kernel): Y pixels in total, 64-/32-pixel chunks in total and per workgroup
    int up_npix, up_nchunks, up_per;
    FastDiv d_yw;
};

template <int TY, int TX, int S>
struct WgradGeom {
    static constexpr int T = TY * TX;
    static constexpr int XPX = ((PWMAX - 1) * S + TX + 3) / 4 * 4;   // staged X pixels per row (the bf16 path reads up to (PWMAX-1)*S + TX - 1 too)
    static constexpr int XGROUPS = XPX / 4;
    static constexpr int YGROUPS = PWMAX / 4;
    static constexpr int RING = 4;                                    // X row slots (>= TY + S)
    static constexpr int XSLOT = XPX * 256;                           // bytes: 64 channels fp32 per pixel
    static constexpr int YBUF = PWMAX * 256;
    static constexpr int LDS = RING * XSLOT + 2 * YBUF;
    static constexpr int NI = (S * XGROUPS + YGROUPS + 3) / 4;       // LDS-DMA instructions per wave and steady-state row step
    static_assert(TY + S <= RING, "ring too small");
};

// (a plain function: with a run-time scalar offset the builtin, used directly inside a kernel's nested lambdas, makes the host
//  pass drop the kernel's stub without a diagnostic)
__device__ __forceinline__ void wg_dma16(__amdgpu_buffer_rsrc_t r, unsigned char *lds, int voff, int soff)
{
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void *)lds, 16, voff, soff, 0, 0);
}

// NSPLIT 0: exact fp32 MFMA; 3: bf16x3 split of fp32 operands (see igemmx.hip).  BUF: the steady-state row step stages through
// buffer descriptors with its items worked out once per partition (tensors below 2 GiB); else global_load_lds per item.
template <int TY, int TX, int S, int NSPLIT, bool BUF>
__global__ __launch_bounds__(256, ((TY * TX == 9 && NSPLIT == 0) ? 3 : 2)) void wgrad_f32_kernel(const WgradK k)
{
    using G = WgradGeom<TY, TX, S>;
    constexpr int T = G::T;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *xs = smem;
    unsigned char *ys = smem + G::RING * G::XSLOT;
    const WgradP &p = k.p;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wi = wave >> 1, wj = wave & 1;

    // block -> (partition group grp, channel tile); tiles of one group are neighbours on one XCD.  A workgroup
    // accumulates ALL partitions P = grp, grp+ngroups, ... of its tile in registers and writes one slab, so the
    // slab traffic is set by the number of resident workgroups, not by how finely the pixels are partitioned.
    int logical;
    {
        const int nblk = gridDim.x, q = nblk >> 3, r = nblk & 7, xcd = blockIdx.x & 7;
        logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (blockIdx.x >> 3);
    }
    const int ntile = k.ntile_i * k.ntile_j;
    const int grp = logical / ntile;
    const int tile = logical - grp * ntile;
    const int it = tile / k.ntile_j, jt = tile - it * k.ntile_j;

    const int l15 = lane & 15, lq = lane >> 4;
    const float *zsrc = p.zeros + 4 * l15;

    f32x16 acc[T];
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    // fused bias gradient: the i-tile-0 workgroups also sum their Y strip over pixels (db[j] = sum dz)
    const bool do_bias = p.db != nullptr && !p.db_on_x && it == 0;
    const bool do_xbias = p.db != nullptr && p.db_on_x && jt == 0;
    float bsum = 0.f;
    const int bch = tid & 63, bpg = tid >> 6;

    const int l31 = lane & 31, lh = lane >> 5;
    const int a_lane = (wi * 32 + l31) * 4;      // byte offset of this lane's X channel within a pixel
    const int b_lane = (wj * 32 + l31) * 4;
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void *)p.X, 0, BUF ? k.xbytes : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc((void *)p.Y, 0, BUF ? k.ybytes : 0, 0x00020000);
    static constexpr int OOB = (int)0x80000000;

    for (int P = grp; P < k.nparts; P += k.ngroups) {
    const int strip = P % k.nstrips;
    const int pc = P / k.nstrips;
    const int chunk = pc % k.nchunks;
    const int img = pc / k.nchunks;
    const int x0 = p.xwin0 + strip * k.pw;
    int pwv = p.xwin1 - x0; pwv = pwv < k.pw ? pwv : k.pw;           // valid pixels in this strip
    const int ya = p.ywin0 + chunk * k.rows_per_chunk;
    int yb = ya + k.rows_per_chunk; yb = yb < p.ywin1 ? yb : p.ywin1;
    const int npairs = (pwv + 1) >> 1;
    const int xcol0 = (x0 + p.ox0) * S - p.xpad;

    // stage one X row (global row xr of image img) into its ring slot, pixel group g
    auto stage_x = [&](int xr, int g) {
        const int px = 4 * g + lq;
        const int xc = xcol0 + px;
        // channels past the tensor's (32-channel layers of the base-32 net fill half a tile) read zeros
        const bool ok = (unsigned)xr < (unsigned)p.XH && (unsigned)xc < (unsigned)p.XW && it * 64 + 4 * l15 < k.ci_real;
        const float *src = p.X + ((size_t)((img * p.XH + xr) * p.XW + xc) * p.XC + p.xc0 + it * 64 + 4 * l15);
        GLDS16(ok ? src : zsrc, xs + (xr & (G::RING - 1)) * G::XSLOT + g * 1024);
    };
    auto stage_y = [&](int y, int buf, int g) {
        const int px = 4 * g + lq;
        const bool ok = px < pwv && jt * 64 + 4 * l15 < k.cj_real;
        const float *src = p.Y + ((size_t)((img * p.YH + y) * p.YW + x0 + px) * p.YC + p.yc0 + jt * 64 + 4 * l15);
        GLDS16(ok ? src : zsrc, ys + buf * G::YBUF + g * 1024);
    };
    // items of one step: S new X rows (XGROUPS groups each) then the Y row; round-robin over waves
    auto stage_step = [&](int y, int buf, int first_row, int nrows) {
        const int xr_base = (y + p.oy0) * S - p.xpad;
        const int nx = nrows * G::XGROUPS;
        for (int e = wave; e < nx + G::YGROUPS; e += 4) {
            if (e < nx) {
                const int rr = e / G::XGROUPS;
                stage_x(xr_base + first_row + rr, e - rr * G::XGROUPS);
            } else {
                stage_y(y, buf, e - nx);
            }
        }
    };

    // BUF: this wave's items of a steady-state step (S new X rows + the Y row), worked out once per partition - the lane part
    // (column, channel, validity) is the instruction's vector offset, the row its scalar offset, so a step issues its LDS-DMA
    // without per-lane address arithmetic (the generic stage_step costs an integer division and ~20 VALU per item; in the
    // bf16 twin of this kernel that was a third of a workgroup's time)
    int iv[G::NI], il[G::NI], irr[G::NI];
    if (BUF) {
#pragma unroll
        for (int i = 0; i < G::NI; ++i) {
            const int e = wave + 4 * i;
            iv[i] = OOB; il[i] = 0; irr[i] = -1;
            if (e < S * G::XGROUPS) {
                const int rr = (S == 2 && e >= G::XGROUPS) ? 1 : 0;
                const int g = e - rr * G::XGROUPS;
                const int xc = xcol0 + 4 * g + lq;
                const bool ok = (unsigned)xc < (unsigned)p.XW && it * 64 + 4 * l15 < k.ci_real;
                iv[i] = ok ? (xc * p.XC + p.xc0 + it * 64 + 4 * l15) * 4 : OOB;
                il[i] = g * 1024; irr[i] = rr;
            } else if (e < S * G::XGROUPS + G::YGROUPS) {
                const int g = e - S * G::XGROUPS;
                const int px = 4 * g + lq;
                const bool ok = px < pwv && jt * 64 + 4 * l15 < k.cj_real;
                iv[i] = ok ? ((x0 + px) * p.YC + p.yc0 + jt * 64 + 4 * l15) * 4 : OOB;
                il[i] = g * 1024;
            }
        }
    }
    auto stage_fast = [&](int y, int buf) {
        const int xr_base = (y + p.oy0) * S - p.xpad + (TY - S);
        const int ysoff = ((img * p.YH + y) * p.YW) * p.YC * 4;
#pragma unroll
        for (int i = 0; i < G::NI; ++i) {
            if (wave + 4 * i < S * G::XGROUPS + G::YGROUPS) {
                if (irr[i] >= 0) {
                    const int xr = xr_base + irr[i];
                    const bool rok = (unsigned)xr < (unsigned)p.XH;
                    wg_dma16(rs_x, xs + (xr & (G::RING - 1)) * G::XSLOT + il[i], rok ? iv[i] : OOB, rok ? ((img * p.XH + xr) * p.XW) * p.XC * 4 : 0);
                } else {
                    wg_dma16(rs_y, ys + buf * G::YBUF + il[i], iv[i], ysoff);
                }
            }
        }
    };

    if (ya < yb) {
        stage_step(ya, 0, 0, TY);                // prologue: all TY rows + Y row
        __syncthreads();
        for (int y = ya; y < yb; ++y) {
            const int cur = (y - ya) & 1;
            if (y + 1 < yb) {                    // rows new to the next step
                if (BUF) stage_fast(y + 1, cur ^ 1); else stage_step(y + 1, cur ^ 1, TY - S, S);
            }
            const int xr0 = (y + p.oy0) * S - p.xpad;
            if (do_bias) {
                const float *yb_ = (const float *)(ys + cur * G::YBUF) + bch;
                for (int px = bpg; px < pwv; px += 4) bsum += yb_[px * 64];
            }
            if (do_xbias) {                  // the S rows of X first used by this step, 'S * pwv' pixels each
                for (int r = 0; r < S; ++r) {
                    const float *xb_ = (const float *)(xs + ((xr0 + TY - S + r) & (G::RING - 1)) * G::XSLOT) + bch;
                    for (int px = bpg; px < S * pwv; px += 4) bsum += xb_[px * 64];
                }
            }
            const unsigned char *yrow = ys + cur * G::YBUF + b_lane;
            if constexpr (NSPLIT == 0) {
            for (int q = 0; q < npairs; ++q) {
                const int pix = 2 * q + lh;
                const float b = *(const float *)(yrow + pix * 256);
#pragma unroll
                for (int ty = 0; ty < TY; ++ty) {
                    const unsigned char *xrow = xs + ((xr0 + ty) & (G::RING - 1)) * G::XSLOT + a_lane;
#pragma unroll
                    for (int tx = 0; tx < TX; ++tx) {
                        const float a = *(const float *)(xrow + (pix * S + tx) * 256);
                        acc[ty * TX + tx] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[ty * TX + tx], 0, 0, 0);
                    }
                }
            }
            } else {
                // bf16 matrix cores (v_mfma_f32_32x32x16_bf16, K = 16 pixels): lane half h supplies pixels 8h..8h+7.
                // fp32 values are read from the LDS strips and split into hi (+ lo) bf16 in registers.
                for (int g16 = 0; g16 * 16 < pwv; ++g16) {
                    const int pix0 = g16 * 16 + 8 * lh;
                    float yv[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) yv[j] = *(const float *)(yrow + (pix0 + j) * 256);
                    bf16x8 bh, bl;
#pragma unroll
                    for (int j = 0; j < 8; ++j) { bh[j] = (__bf16)yv[j]; if (NSPLIT == 3) bl[j] = (__bf16)(yv[j] - (float)bh[j]); }
#pragma unroll
                    for (int ty = 0; ty < TY; ++ty) {
                        const unsigned char *xrow = xs + ((xr0 + ty) & (G::RING - 1)) * G::XSLOT + a_lane;
                        // the TX taps of this filter row read a sliding window of 8*S + TX - S pixels
                        constexpr int NV = 7 * S + TX;
                        float xv[NV];
                        __bf16 xh[NV], xl[NV];
#pragma unroll
                        for (int j = 0; j < NV; ++j) {
                            xv[j] = *(const float *)(xrow + (pix0 * S + j) * 256);
                            xh[j] = (__bf16)xv[j];
                            if (NSPLIT == 3) xl[j] = (__bf16)(xv[j] - (float)xh[j]);
                        }
#pragma unroll
                        for (int tx = 0; tx < TX; ++tx) {
                            bf16x8 ah, al;
#pragma unroll
                            for (int j = 0; j < 8; ++j) { ah[j] = xh[j * S + tx]; if (NSPLIT == 3) al[j] = xl[j * S + tx]; }
                            const int t = ty * TX + tx;
                            if (NSPLIT == 3) {
                                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[t], 0, 0, 0);
                                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[t], 0, 0, 0);
                            }
                            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[t], 0, 0, 0);
                        }
                    }
                }
            }
            __syncthreads();
        }
    }

    }   // partitions of this workgroup

    // accumulated tile -> slab of this group (+ [Cj] bias partials behind the T*Ci*Cj weight partials)
    float *slab = p.slab + (size_t)grp * k.pstride;
    if (do_bias || do_xbias) {
        float *red = (float *)smem;             // all LDS reads of the loop are behind its last barrier
        red[tid] = bsum;
        __syncthreads();
        if (tid < 64) slab[(size_t)T * p.Ci * p.Cj + (do_xbias ? it : jt) * 64 + tid] = (red[tid] + red[tid + 64]) + (red[tid + 128] + red[tid + 192]);
    }
    // register-order layout [tile][wave][t][r/4][lane][r%4]: one 16-byte store per lane and (t, r/4), 1 KiB per
    // wave instruction (the 144 dword stores this replaces were store-issue bound); the reduce kernel decodes it.
    f32x4 *dst4 = (f32x4 *)(slab + ((size_t)(tile * 4 + wave) * T) * 1024) + lane;
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq)
            dst4[(t * 4 + rq) * 64] = f32x4{acc[t][4 * rq], acc[t][4 * rq + 1], acc[t][4 * rq + 2], acc[t][4 * rq + 3]};
}

// ---------------------------------------------------------------------------------------------------------------------
// bf16 tensors (arithmetic mode 2): X and Y are bf16 NHWC, the MFMA is v_mfma_f32_32x32x16_bf16 with K = 16 pixels, the
// partial slabs and the result stay fp32.  Same decomposition as above (a 64x64 channel tile for all taps per workgroup,
// split-K over (image, row chunk, pixel strip), one slab per workgroup, deterministic reduce), with what the 16x faster
// pipe needs:
//   * strips of 64 pixels (4 MFMA k-steps per tap and row), so that a row's MFMA work covers the LDS-DMA of the next row;
//   * both operands are K-major in memory (pixel rows of 64 channels = 128 B) but the MFMA wants them K-minor per lane:
//     the fragments are read with ds_read_b64_tr_b16 (4 pixels x 16 channels per 16-lane group, delivered channel-per-lane),
//     two reads per operand fragment, no VALU transposition;
//   * LDS images are filled by LDS-DMA (8 pixels per instruction); 16-byte chunk c of pixel P sits at chunk c ^ 4((P>>1)&1):
//     the four pixel rows a half-wave reads then fall on four disjoint bank-row quarters for every tap shift;
//   * the fused bias gradient is the sum of the fragment values a wave holds anyway (fp32 adds).
constexpr int PWB = 64;
template <int TY, int TX, int S>
struct WgradBGeom {
    static constexpr int T = TY * TX;
    static constexpr int XPX = (PWB + TX - 1 + 7) / 8 * 8;                         // staged X pixel positions per row
    static constexpr int XG = XPX / 8, YG = PWB / 8;                               // LDS-DMA instructions (8 pixels) per row
    // one row step of prefetch: a second one (5-slot X ring, 3 dz buffers, counted vmcnt) was measured 5 % slower with in-kernel
    // cycle stamps - the wait + barrier share is 3 % of a workgroup's time, what costs is the LDS-DMA issue itself (below)
    static constexpr int RING = TY + S, NY = 2;
    static constexpr int XSLOT = XPX * 128, YBUF = PWB * 128;
    static constexpr int LDS = RING * XSLOT + NY * YBUF;
    static constexpr int NI = (S * XG + YG + 3) / 4;                  // LDS-DMA instructions per wave and steady-state row step
    static_assert(2 * LDS <= 160 * 1024, "two workgroups per CU");
    static_assert(S == 1, "stride-1 layers only: the up-conv weight gradient is wgrad_up_kernel");
};

typedef short s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ bf16x8 tr_frag(const unsigned char *p0)       // pixels +0..3 at p0, +4..7 at p0 + 4*128
{
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(p0));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(p0 + 4 * 128));
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}
__device__ __forceinline__ float frag_sum(const bf16x8 &v)
{
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += (float)v[j];
    return s;
}

// (a plain function: with a run-time scalar offset the builtin, used directly inside the kernel's nested lambdas, makes the host
//  pass drop the kernel's stub without a diagnostic)
__device__ __forceinline__ void wgb_dma16(__amdgpu_buffer_rsrc_t r, unsigned char *lds, int voff, int soff)
{
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void *)lds, 16, voff, soff, 0, 0);
}

template <int TY, int TX, int S>
__global__ __launch_bounds__(256, 2) void wgrad_bf16_kernel(const WgradK k)
{
    using G = WgradBGeom<TY, TX, S>;
    constexpr int T = G::T;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *xs = smem;
    unsigned char *ys = smem + G::RING * G::XSLOT;
    const WgradP &p = k.p;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wi = wave >> 1, wj = wave & 1;

    int logical;
    {
        const int nblk = gridDim.x, q = nblk >> 3, r = nblk & 7, xcd = blockIdx.x & 7;
        logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (blockIdx.x >> 3);
    }
    const int ntile = k.ntile_i * k.ntile_j;
    const int grp = logical / ntile;
    const int tile = logical - grp * ntile;
    const int it = tile / k.ntile_j, jt = tile - it * k.ntile_j;

    f32x16 acc[T];
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const bool do_bias = p.db != nullptr && !p.db_on_x && it == 0 && wi == 0;      // db[j] = sum dz: from the B fragments
    const bool do_xbias = p.db != nullptr && p.db_on_x && jt == 0 && wj == 0;      // db[i] = sum X:  from the A fragments
    float bsum = 0.f;

    // DMA role of a lane inside an 8-pixel instruction: pixel lane>>3, LDS chunk position lane&7 <- source chunk (swizzled)
    const int d_px = lane >> 3;
    const int d_c = (lane & 7) ^ (((lane >> 4) & 1) << 2);
    // fragment-read role: lane = 16 g + 4 q + pp: pixel row q of the group's 4x16 block, 8-byte piece pp; the group's channels
    // are 16 (g & 1) .. +15 of the wave's 32, its pixels 8 (g >> 1) .. +7 of the k-step
    const int f_q = (lane >> 2) & 3, f_pp = lane & 3, f_mh = (lane >> 4) & 1, f_h = lane >> 5;
    const int cA = wi * 4 + f_mh * 2 + (f_pp >> 1), cB = wj * 4 + f_mh * 2 + (f_pp >> 1);
    const int yoff = (8 * f_h + f_q) * 128 + ((cB ^ (((f_q >> 1) & 1) << 2)) * 16) + (f_pp & 1) * 8;
    int xoff[TX];
#pragma unroll
    for (int tx = 0; tx < TX; ++tx) {
        xoff[tx] = (8 * f_h + f_q + tx) * 128 + ((cA ^ ((((f_q + tx) >> 1) & 1) << 2)) * 16) + (f_pp & 1) * 8;
    }

    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void *)p.X, 0, k.xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc((void *)p.Y, 0, k.ybytes, 0x00020000);
    static constexpr int OOB = (int)0x80000000;

    // One copy of the whole partition loop per bias role (wave-uniform): the k loop's body must be straight-line code — a
    // branch per tap made the compiler wait for every fragment right before its MFMA (LDS latency exposed 36 times per row) —
    // and role branches INSIDE the row loop made the register allocator spill the accumulators around them.
    auto run = [&](auto role_tag) {
    constexpr int ROLE = decltype(role_tag)::value;        // 0: no bias sum, 1: db from the dz fragments, 2: db from the X fragments
    for (int P = grp; P < k.nparts; P += k.ngroups) {
    const int strip = P % k.nstrips;
    const int pc = P / k.nstrips;
    const int chunk = pc % k.nchunks;
    const int img = pc / k.nchunks;
    const int x0 = p.xwin0 + strip * k.pw;
    int pwv = p.xwin1 - x0; pwv = pwv < k.pw ? pwv : k.pw;           // valid pixels in this strip
    const int ya = p.ywin0 + chunk * k.rows_per_chunk;
    int yb = ya + k.rows_per_chunk; yb = yb < p.ywin1 ? yb : p.ywin1;
    const int nks = (pwv + 15) >> 4;
    const int xcol0 = (x0 + p.ox0) * S - p.xpad;

    auto stage_x = [&](int xr, int slot, int g) {
        const int pos = 8 * g + d_px;
        const int xc = xcol0 + pos;
        const bool ok = (unsigned)xr < (unsigned)p.XH && (unsigned)xc < (unsigned)p.XW;
        const int off = (((img * p.XH + xr) * p.XW + xc) * p.XC + p.xc0 + it * 64 + d_c * 8) * 2;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (__attribute__((address_space(3))) void *)(xs + slot * G::XSLOT + g * 1024), 16,
                                                 ok ? off : OOB, 0, 0, 0);
    };
    auto stage_y = [&](int y, int buf, int g) {
        const int px = 8 * g + d_px;
        const bool ok = px < pwv;
        const int off = (((img * p.YH + y) * p.YW + x0 + px) * p.YC + p.yc0 + jt * 64 + d_c * 8) * 2;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_y, (__attribute__((address_space(3))) void *)(ys + buf * G::YBUF + g * 1024), 16, ok ? off : OOB, 0, 0, 0);
    };
    // items of one step: S new X rows (XG groups each) then the Y row; round-robin over waves.  Groups past the strip's
    // last needed pixel are skipped (their LDS content is never read: k-steps stop at nks).
    const int xg_used = (16 * nks + TX - 1 + 7) >> 3;
    const int yg_used = 2 * nks;
    // step j (row ya + j) reads the X rows j*S .. j*S + TY-1 of the chunk (ring slot = row index mod RING) and Y buffer j mod NY
    auto stage_step = [&](int y, int j, int first_row, int nrows) {
        const int xr_base = (y + p.oy0) * S - p.xpad;
        const int nx = nrows * xg_used;
        const int buf = j % G::NY;
        for (int e = wave; e < nx + yg_used; e += 4) {
            if (e < nx) {
                const int rr = e / xg_used;
                stage_x(xr_base + first_row + rr, (j * S + first_row + rr) % G::RING, e - rr * xg_used);
            } else {
                stage_y(y, buf, e - nx);
            }
        }
    };
    // LDS-DMA instructions this wave issues for one steady-state step (S new X rows + the Y row)
    const int n_items = S * xg_used + yg_used;
    // The steady-state step's items of this wave, worked out once per partition: everything that depends on the lane (column,
    // channel chunk, column validity) sits in the buffer instruction's vector offset, the row goes into its scalar offset -
    // a step then issues its LDS-DMA with no per-lane address arithmetic.  In-kernel cycle stamps (8 x 280^2 x 128 -> 128): the
    // generic stage_step above (an integer division and ~20 VALU instructions per item) held a wave 33 % of its time, this
    // 19 % - what is left is the texture-address unit taking the workgroups' 17 one-KiB instructions per row (~200 cycles per
    // instruction seen from the issuing wave, all eight waves of the CU queueing behind each other right after a barrier).
    // Spreading the items over the k loop instead (one per k-step) costs more than it hides: the branches cut the loop's
    // software pipeline (weight-gradient family 2.87 -> 3.93 ms per step).
    int iv[G::NI], il[G::NI], irr[G::NI];
#pragma unroll
    for (int i = 0; i < G::NI; ++i) {
        const int e = wave + 4 * i;
        iv[i] = OOB; il[i] = 0; irr[i] = -1;
        if (e < S * xg_used) {
            const int rr = 0;
            const int g = e;
            const int pos = 8 * g + d_px;
            const int xc = xcol0 + pos;
            const bool ok = (unsigned)xc < (unsigned)p.XW;
            iv[i] = ok ? (xc * p.XC + p.xc0 + it * 64 + d_c * 8) * 2 : OOB;
            il[i] = g * 1024; irr[i] = rr;
        } else if (e < n_items) {
            const int g = e - S * xg_used;
            const int px = 8 * g + d_px;
            iv[i] = px < pwv ? ((x0 + px) * p.YC + p.yc0 + jt * 64 + d_c * 8) * 2 : OOB;
            il[i] = g * 1024;
        }
    }
    auto stage_fast = [&](int y, int j) {
        const int xr_base = (y + p.oy0) * S - p.xpad + (TY - S);
        unsigned char *yb_ = ys + (j % G::NY) * G::YBUF;
        const int ysoff = ((img * p.YH + y) * p.YW) * p.YC * 2;
#pragma unroll
        for (int i = 0; i < G::NI; ++i) {
            if (wave + 4 * i < n_items) {
                if (irr[i] >= 0) {
                    const int xr = xr_base + irr[i];
                    const bool rok = (unsigned)xr < (unsigned)p.XH;
                    unsigned char *dst = xs + ((j * S + (TY - S) + irr[i]) % G::RING) * G::XSLOT + il[i];
                    wgb_dma16(rs_x, dst, rok ? iv[i] : OOB, rok ? ((img * p.XH + xr) * p.XW) * p.XC * 2 : 0);
                } else {
                    wgb_dma16(rs_y, yb_ + il[i], iv[i], ysoff);
                }
            }
        }
    };

    if (ya < yb) {
        stage_step(ya, 0, 0, TY);
        __syncthreads();
        for (int y = ya, j = 0; y < yb; ++y, ++j) {
            if (y + 1 < yb) stage_fast(y + 1, j + 1);
            const unsigned char *yrow = ys + (j % G::NY) * G::YBUF + yoff;
            const unsigned char *xrow[TY];
#pragma unroll
            for (int ty = 0; ty < TY; ++ty) xrow[ty] = xs + ((j * S + ty) % G::RING) * G::XSLOT;
            // Software pipeline in place: as soon as tap t's MFMA of k-step ks has issued, its fragment register is refilled
            // with the fragment of k-step ks+1 (the last k-step re-reads itself: branch-free), so 9 reads are always in flight
            // behind the MFMAs and a second fragment set is not needed.
            {
                bf16x8 a[T], b;
                b = tr_frag(yrow);
#pragma unroll
                for (int ty = 0; ty < TY; ++ty)
#pragma unroll
                    for (int tx = 0; tx < TX; ++tx) a[ty * TX + tx] = tr_frag(xrow[ty] + xoff[tx]);
                for (int ks = 0; ks < nks; ++ks) {
                    const int kn = (ks + 1 < nks ? ks + 1 : ks) * (16 * 128);
                    const bf16x8 bn = tr_frag(yrow + kn);
                    if (ROLE == 1) bsum += frag_sum(b);
#pragma unroll
                    for (int ty = 0; ty < TY; ++ty)
#pragma unroll
                        for (int tx = 0; tx < TX; ++tx) {
                            const int t = ty * TX + tx;
                            if (ROLE == 2) bsum += frag_sum(a[t]);
                            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[t], b, acc[t], 0, 0, 0);
                            a[t] = tr_frag(xrow[ty] + kn + xoff[tx]);
                            __builtin_amdgcn_sched_barrier(0);      // keep the refill BEHIND its MFMA: hoisted, it needs a second register set
                        }
                    b = bn;
                }
            }
            __syncthreads();            // the next step's rows have landed (vmcnt(0)); every wave is done with this step's slots
        }
    }

    }   // partitions of this workgroup
    };
    if (do_xbias) run(std::integral_constant<int, 2>{});
    else if (do_bias) run(std::integral_constant<int, 1>{});
    else run(std::integral_constant<int, 0>{});

    float *slab = p.slab + (size_t)grp * k.pstride;
    if (do_bias || do_xbias) {
        const float v = bsum + __shfl_xor(bsum, 32, 64);
        if (lane < 32) slab[(size_t)T * p.Ci * p.Cj + (do_xbias ? it * 64 + wi * 32 : jt * 64 + wj * 32) + lane] = v;
    }
    f32x4 *dst4 = (f32x4 *)(slab + ((size_t)(tile * 4 + wave) * T) * 1024) + lane;
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq)
            dst4[(t * 4 + rq) * 64] = f32x4{acc[t][4 * rq], acc[t][4 * rq + 1], acc[t][4 * rq + 2], acc[t][4 * rq + 3]};
}

// ---------------------------------------------------------------------------------------------------------------------
// Up-conv weight gradient (2x2 taps, stride 2; network.py:38-53), PIXEL-LINEAR: the taps of a stride-2 transposed convolution
// do not overlap, so there is no halo and no reason to walk image rows: the reduction runs over the layer-input pixels
// p = (image, y, x) in plain linear order, in chunks of 64 (bf16) / 32 (fp32) consecutive pixels that cross row and image
// boundaries freely.  The row-walking kernels above stage a strip of one row per barrier - for the 28 ... 196-pixel rows of
// the up-convs that is 2 ... 4 MFMA k-steps per tap between barriers with bf16 tensors (MFMA busy 0.12, round 3) and a ragged
// last strip per row in fp32 (0.53-0.56).  Here every chunk is full: per chunk a workgroup stages [4 taps][PIX][64 co] of
// dOut (gathered: pixel (2y + ty, 2x + tx)) and [PIX][64 ci] of the layer input by LDS-DMA (40 KiB, two stages), and every
// wave runs 4 taps x PIX / (16 | 2) MFMAs on its 32 (co) x 32 (ci) block.  Same channel tile per workgroup (64 x 64, all four
// taps), same LDS images and fragment reads (ds_read_b64_tr_b16 / ds_read_b32), same slab format, bias partial (db = sum of
// the dOut fragments) and reduce as the kernels above; a workgroup owns a contiguous range of chunks.
template <bool BF>
struct UpGeom {
    static constexpr int ES = BF ? 2 : 4;
    static constexpr int PIX = BF ? 64 : 32;              // pixels per chunk
    static constexpr int ROWB = 64 * ES;                  // bytes of a pixel's 64 channels
    static constexpr int IMG = PIX * ROWB;                // 8 KiB: one tap's (or Y's) pixels of a chunk
    static constexpr int STAGE = 5 * IMG;                 // X taps 0..3, then Y
    static constexpr int LDS = 2 * STAGE;                 // 80 KiB: two workgroups per CU
    static constexpr int PPI = 1024 / ROWB;               // pixels per 1-KiB LDS-DMA instruction: 8 (bf16) / 4 (fp32)
    static constexpr int NG = PIX / PPI;                  // instructions per image: 8
    static_assert(NG == 8, "a wave stages pixel groups w and w + 4 of all five images");
};

template <bool BF>
__global__ __launch_bounds__(256, 2) void wgrad_up_kernel(const WgradK k)
{
    using G = UpGeom<BF>;
    constexpr int T = 4, ES = G::ES;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const WgradP &p = k.p;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wi = wave >> 1, wj = wave & 1;

    int logical;
    {
        const int nblk = gridDim.x, q = nblk >> 3, r = nblk & 7, xcd = blockIdx.x & 7;
        logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (blockIdx.x >> 3);
    }
    const int ntile = k.ntile_i * k.ntile_j;
    const int grp = logical / ntile;
    const int tile = logical - grp * ntile;
    const int it = tile / k.ntile_j, jt = tile - it * k.ntile_j;

    f32x16 acc[T];
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    // db[co] = sum of dOut over all pixels: every dOut pixel is staged exactly once (stride == taps); the jt == 0 workgroups sum it
    const bool do_xbias = p.db != nullptr && jt == 0 && (BF ? wj == 0 : true);
    float bsum = 0.f;

    // ---- DMA role: pixel (lane / lanes-per-pixel) of an instruction's PPI pixels, 16-byte piece of its 64 channels
    const int d_px = BF ? lane >> 3 : lane >> 4;
    const int d_ch = BF ? (((lane & 7) ^ (((lane >> 4) & 1) << 2)) * 8) : 4 * (lane & 15);        // source channel (bf16: swizzled chunk)
    const bool x_ch_ok = it * 64 + d_ch < k.ci_real, y_ch_ok = jt * 64 + d_ch < k.cj_real;
    const int x_lane = (p.xc0 + it * 64 + d_ch) * ES, y_lane = (p.yc0 + jt * 64 + d_ch) * ES;
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void *)p.X, 0, k.xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc((void *)p.Y, 0, k.ybytes, 0x00020000);
    static constexpr int OOB = (int)0x80000000;
    const int YW = p.YW;
    // tap (ty, tx) of dOut pixel (2y + ty, 2x + tx): a uniform byte offset from tap (0, 0)
    int tap_off[T];
#pragma unroll
    for (int t = 0; t < T; ++t) tap_off[t] = ((t >> 1) * p.XW + (t & 1)) * p.XC * ES;

    // stage chunk c into buffer buf: this wave's pixel groups g = wave, wave + 4 of all five images; one division per group
    auto stage = [&](int c, int buf) {
        unsigned char *sb = smem + buf * G::STAGE;
#pragma unroll
        for (int gg = 0; gg < 2; ++gg) {
            const int g = wave + 4 * gg;
            const int pix = c * G::PIX + g * G::PPI + d_px;
            const bool ok = pix < k.up_npix;
            const int r = fdiv_u(pix, k.d_yw);                 // (image, y) row index of the layer input
            const int x = pix - r * YW;
            const int xo = ok && x_ch_ok ? ((4 * r * YW + 2 * x) * p.XC) * ES + x_lane : OOB;      // dOut pixel (2y, 2x)
            const int yo = ok && y_ch_ok ? pix * p.YC * ES + y_lane : OOB;
#pragma unroll
            for (int t = 0; t < T; ++t) wg_dma16(rs_x, sb + t * G::IMG + g * 1024, xo, tap_off[t]);
            wg_dma16(rs_y, sb + 4 * G::IMG + g * 1024, yo, 0);
        }
    };

    const int c0 = grp * k.up_per;
    int c1 = c0 + k.up_per;
    c1 = c1 < k.up_nchunks ? c1 : k.up_nchunks;

    if constexpr (BF) {
        // fragment-read role (as wgrad_bf16_kernel): lane = 16 g + 4 q + pp
        const int f_q = (lane >> 2) & 3, f_pp = lane & 3, f_mh = (lane >> 4) & 1, f_h = lane >> 5;
        const int cA = wi * 4 + f_mh * 2 + (f_pp >> 1), cB = wj * 4 + f_mh * 2 + (f_pp >> 1);
        const int sw = ((f_q >> 1) & 1) << 2;
        const int xoff = (8 * f_h + f_q) * 128 + ((cA ^ sw) * 16) + (f_pp & 1) * 8;
        const int yoff = 4 * G::IMG + (8 * f_h + f_q) * 128 + ((cB ^ sw) * 16) + (f_pp & 1) * 8;
        if (c0 < c1) {
            stage(c0, 0);
            __syncthreads();
            for (int c = c0; c < c1; ++c) {
                const int cur = (c - c0) & 1;
                if (c + 1 < c1) stage(c + 1, cur ^ 1);
                const unsigned char *sb = smem + cur * G::STAGE;
                bf16x8 a[T], b;
                b = tr_frag(sb + yoff);
#pragma unroll
                for (int t = 0; t < T; ++t) a[t] = tr_frag(sb + t * G::IMG + xoff);
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const int kn = (ks + 1 < 4 ? ks + 1 : ks) * (16 * 128);
                    const bf16x8 bn = tr_frag(sb + yoff + kn);
#pragma unroll
                    for (int t = 0; t < T; ++t) {
                        if (do_xbias) bsum += frag_sum(a[t]);
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[t], b, acc[t], 0, 0, 0);
                        a[t] = tr_frag(sb + t * G::IMG + xoff + kn);
                        __builtin_amdgcn_sched_barrier(0);      // keep the refill BEHIND its MFMA
                    }
                    b = bn;
                }
                __syncthreads();        // the next chunk has landed (vmcnt(0)); every wave is done with this buffer
            }
        }
    } else {
        const int l31 = lane & 31, lh = lane >> 5;
        const int a_lane = (wi * 32 + l31) * 4, b_lane = 4 * G::IMG + (wj * 32 + l31) * 4;
        const int bch = tid & 63, bpg = tid >> 6;
        if (c0 < c1) {
            stage(c0, 0);
            __syncthreads();
            for (int c = c0; c < c1; ++c) {
                const int cur = (c - c0) & 1;
                if (c + 1 < c1) stage(c + 1, cur ^ 1);
                const unsigned char *sb = smem + cur * G::STAGE;
                if (do_xbias) {
                    const float *xb_ = (const float *)sb + bch;
#pragma unroll
                    for (int t = 0; t < T; ++t)
                        for (int px = bpg; px < G::PIX; px += 4) bsum += xb_[(t * G::PIX + px) * 64];
                }
#pragma unroll 4
                for (int q = 0; q < G::PIX / 2; ++q) {
                    const int pix = 2 * q + lh;
                    const float b = *(const float *)(sb + b_lane + pix * 256);
#pragma unroll
                    for (int t = 0; t < T; ++t) {
                        const float a = *(const float *)(sb + t * G::IMG + a_lane + pix * 256);
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
                    }
                }
                __syncthreads();
            }
        }
    }

    // slab of this group: the weight partials in register order, then the bias partials (formats of the kernels above)
    float *slab = p.slab + (size_t)grp * k.pstride;
    if constexpr (BF) {
        if (do_xbias) {
            const float v = bsum + __shfl_xor(bsum, 32, 64);
            if (lane < 32) slab[(size_t)T * p.Ci * p.Cj + it * 64 + wi * 32 + lane] = v;
        }
    } else {
        if (do_xbias) {
            float *red = (float *)smem;             // all LDS reads of the loop are behind its last barrier
            red[tid] = bsum;
            __syncthreads();
            if (tid < 64) slab[(size_t)T * p.Ci * p.Cj + it * 64 + tid] = (red[tid] + red[tid + 64]) + (red[tid + 128] + red[tid + 192]);
        }
    }
    f32x4 *dst4 = (f32x4 *)(slab + ((size_t)(tile * 4 + wave) * T) * 1024) + lane;
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq)
            dst4[(t * 4 + rq) * 64] = f32x4{acc[t][4 * rq], acc[t][4 * rq + 1], acc[t][4 * rq + 2], acc[t][4 * rq + 3]};
}

// out[i*si + j*sj + t*st] = sum_P slab_P(t,i,j)  and  db[j] = sum_P slab[P][T*Ci*Cj + j], in a fixed order.
// Workgroup = 64 consecutive outputs x 4 partition groups (combined through LDS): enough loads in flight
// even when the output is tiny (64x64x9) and the partition count is in the thousands.
// A thread sums 4 consecutive outputs (16-byte loads: 1 KiB per wave instruction) over the slabs of its partition group;
// PG groups (4, or 16 when the launch has many slabs: few outputs, e.g. 64 x 64 x 9, and thousands of partitions) are combined
// through LDS in a fixed order.
template <int PG>
__global__ __launch_bounds__(64 * PG) void wgrad_reduce_kernel(const float *__restrict__ slab, int nP, size_t pstride, int T, int Ci, int Cj,
                                                               int ci_real, int cj_real,
                                                               float *__restrict__ out, long si, long sj, long st, float *__restrict__ db, int ndb)
{
    const size_t nw = (size_t)T * Ci * Cj;
    const size_t total = nw + (db ? (size_t)ndb : 0);          // nw, ndb and pstride are multiples of 4
    const int lane_e = threadIdx.x & 63, grp = threadIdx.x >> 6;
    __shared__ f32x4 red[PG][64];
    for (size_t base = (size_t)blockIdx.x * 256; base < total; base += (size_t)gridDim.x * 256) {
        const size_t e = base + 4 * lane_e;
        f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
        if (e < total) {
            const float *src = slab + e;
            int P = grp;
            for (; P + 3 * PG < nP; P += 4 * PG) {
                s0 += *(const f32x4 *)(src + (size_t)P * pstride);
                s1 += *(const f32x4 *)(src + (size_t)(P + PG) * pstride);
                s2 += *(const f32x4 *)(src + (size_t)(P + 2 * PG) * pstride);
                s3 += *(const f32x4 *)(src + (size_t)(P + 3 * PG) * pstride);
            }
            for (; P < nP; P += PG) s0 += *(const f32x4 *)(src + (size_t)P * pstride);
        }
        red[grp][lane_e] = (s0 + s1) + (s2 + s3);
        __syncthreads();
        if (grp == 0 && e < total) {
            f32x4 v = red[0][lane_e];
#pragma unroll
            for (int g = 1; g < PG; ++g) v += red[g][lane_e];
            if (e < nw) {
                // decode the register-order slab index (see wgrad_f32_kernel): [tile][wave][t][r/4][lane][r%4]; e % 4 == 0
                const int ln = (int)((e >> 2) & 63), rq = (int)((e >> 8) & 3);
                const size_t hi = e >> 10;
                const int t = (int)(hi % T);
                const int wv = (int)((hi / T) & 3);
                const int tile = (int)(hi / T / 4);
                const int ntj = Cj >> 6;
                const int it = tile / ntj, jt = tile - it * ntj;
                const int i0 = it * 64 + (wv >> 1) * 32 + 8 * rq + 4 * (ln >> 5);
                const int j = jt * 64 + (wv & 1) * 32 + (ln & 31);
#pragma unroll
                for (int ri = 0; ri < 4; ++ri)
                    if (i0 + ri < ci_real && j < cj_real) out[(i0 + ri) * si + j * sj + t * st] = v[ri];
            } else {
#pragma unroll
                for (int ri = 0; ri < 4; ++ri)
                    if (e - nw + ri < (size_t)ndb) db[e - nw + ri] = v[ri];
            }
        }
        __syncthreads();
    }
}

static void decompose(const WgradP &p, WgradK &k)
{
    const int wx = p.xwin1 - p.xwin0, wy = p.ywin1 - p.ywin0;
    const int pwmax = p.math == 2 ? PWB : PWMAX;
    k.nstrips = cdiv(wx, pwmax);
    int pw = cdiv(wx, k.nstrips);
    pw = p.math == 2 ? (pw + 15) & ~15 : (pw + 1) & ~1;       // whole MFMA k-steps: 16 pixels (bf16) / a pixel pair (fp32)
    if (pw > pwmax) pw = pwmax;
    k.pw = pw;
    k.nstrips = cdiv(wx, pw);
    k.ntile_i = p.Ci / 64;
    k.ntile_j = p.Cj / 64;
    k.ci_real = k.cj_real = 0;             // set by launch_wgrad
    // Partitioning.  The pixels are cut into partitions (image x row-chunk x strip); `ngroups` workgroups per
    // channel tile share them round-robin, each accumulating its partitions in registers and writing ONE slab.
    // Choose rows-per-chunk r and ngroups to minimise  rounds(ngroups*ntile / resident slots) x ceil(nparts/ngroups) x r
    // (all workgroups of a launch do the same per-row work), preferring fewer groups (slab traffic) on ties.
    const int ntile = k.ntile_i * k.ntile_j;
    const int slots = 256 * ((p.TY == 3 && (p.math == 0 || p.math == 3)) ? 3 : 2);
    const long per_chunk = (long)p.NB * k.nstrips;
    // the search is a few million cheap iterations: memoise per shape (hot calls hit the cache)
    static std::mutex mu;
    static std::map<std::array<long, 4>, std::pair<int, int>> cache;
    const std::array<long, 4> key = {per_chunk, (long)wy, (long)ntile, (long)slots};       // (the strip width is folded into per_chunk)
    {
        std::lock_guard<std::mutex> lk(mu);
        auto itc = cache.find(key);
        if (itc != cache.end()) {
            k.rows_per_chunk = itc->second.first;
            k.nchunks = cdiv(wy, k.rows_per_chunk);
            k.nparts = (int)(per_chunk * k.nchunks);
            k.ngroups = itc->second.second;
            k.pstride = align_up((size_t)p.TY * p.TX * p.Ci * p.Cj + (p.Ci > p.Cj ? p.Ci : p.Cj), 64);
            return;
        }
    }
    double best = 1e300;
    int best_r = wy, best_g = 1;
    for (int r = 4; r <= wy; ++r) {
        const int nch = cdiv(wy, r);
        if (r > 4 && cdiv(wy, r - 1) == nch && r != wy) continue;   // a smaller r gives the same chunk count: skip (keep the first)
        const long nparts = per_chunk * nch;
        long gmax = nparts < 8L * slots / ntile + 1 ? nparts : 8L * slots / ntile + 1;
        if (gmax < 1) gmax = 1;
        for (long g = 1; g <= gmax; ++g) {
            // workgroups beyond the resident slots queue; measured behaviour sits between lockstep rounds
            // (ceil) and perfect refill (fraction), so blend the two; per-partition fixed cost ~ 2 rows' worth
            // (prologue staging of the halo ring, pipeline warm-up)
            const double frac = (double)(g * ntile) / (double)slots;
            const double rounds = frac <= 1.0 ? 1.0 : 0.5 * (frac + (double)((g * ntile + slots - 1) / slots));
            const double t = rounds * (double)((nparts + g - 1) / g) * (double)(r + 2);
            const double score = t * (1.0 + 0.02 * rounds);            // fewer groups on ties: less slab traffic
            if (score < best - 1e-9) { best = score; best_r = r; best_g = (int)g; }
        }
        if (nch == 1) break;
    }
    k.rows_per_chunk = best_r;
    k.nchunks = cdiv(wy, best_r);
    k.nparts = (int)(per_chunk * k.nchunks);
    k.ngroups = best_g;
    {
        std::lock_guard<std::mutex> lk(mu);
        cache[key] = std::make_pair(best_r, best_g);
    }
    k.pstride = align_up((size_t)p.TY * p.TX * p.Ci * p.Cj + (p.Ci > p.Cj ? p.Ci : p.Cj), 64);
}

// the pixel-linear up-conv form: the full-window 2x2 stride-2 weight gradient with its bias gradient on X (or none), tensors
// below 2 GiB (buffer descriptors), exact fp32 or bf16 tensors (the bf16x3 split keeps the row-walking kernel)
static bool up_applicable(const WgradP &p)
{
    static const int on = [] { const char *e = getenv("UNET_WGRAD_UP"); return e ? atoi(e) : 1; }();      // 0: the row-walking kernels (A/B)
    if (!on || p.TY != 2 || p.TX != 2 || p.stride != 2 || p.xpad != 0 || p.oy0 != 0 || p.ox0 != 0) return false;
    if (p.math == 1 || (p.db && !p.db_on_x)) return false;
    if (p.ywin0 != 0 || p.xwin0 != 0 || p.ywin1 != p.YH || p.xwin1 != p.YW || p.XH != 2 * p.YH || p.XW != 2 * p.YW) return false;
    const size_t es = p.math == 2 ? 2 : 4;
    return (size_t)p.NB * p.XH * p.XW * p.XC * es < 0x7FFFFFFFull && (size_t)p.NB * p.YH * p.YW * p.YC * es < 0x7FFFFFFFull && get_lds_dma_mode() != 0;
}

static void up_decompose(const WgradP &p, WgradK &k)
{
    const int pix = p.math == 2 ? 64 : 32;
    k.ntile_i = p.Ci / 64; k.ntile_j = p.Cj / 64;
    k.up_npix = p.NB * p.YH * p.YW;
    k.up_nchunks = cdiv(k.up_npix, pix);
    const int ntile = k.ntile_i * k.ntile_j;
    int g = 512 / ntile;                        // two workgroups per CU (80 KiB of LDS each)
    if (g < 1) g = 1;
    if (g > k.up_nchunks) g = k.up_nchunks;
    k.up_per = cdiv(k.up_nchunks, g);
    k.ngroups = cdiv(k.up_nchunks, k.up_per);
    k.d_yw = make_fastdiv((unsigned)p.YW);
    k.pstride = align_up((size_t)p.TY * p.TX * p.Ci * p.Cj + (p.Ci > p.Cj ? p.Ci : p.Cj), 64);
}

template <bool BF>
static int launch_wgrad_up(WgradK &k, hipStream_t st)
{
    using G = UpGeom<BF>;
    static bool attr_done[64] = {false};
    auto kern = wgrad_up_kernel<BF>;
    if (int rc_ = ensure_dynamic_lds((const void *)kern, G::LDS, attr_done)) return rc_;
    k.xbytes = (int)((size_t)k.p.NB * k.p.XH * k.p.XW * k.p.XC * G::ES);
    k.ybytes = (int)((size_t)k.p.NB * k.p.YH * k.p.YW * k.p.YC * G::ES);
    char tag[96];
    snprintf(tag, sizeof(tag), "wgrad_up<%s> Ci=%d Cj=%d Y=%dx%d chunks=%d per=%d groups=%d", BF ? "bf16" : "f32", k.p.Ci, k.p.Cj, k.p.YH, k.p.YW, k.up_nchunks, k.up_per, k.ngroups);
    prof_begin(PK_WGRAD, tag, st, wgrad_alg_flops(k.p), 2.0 * (double)k.up_nchunks * G::PIX * 4.0 * k.p.Ci * k.p.Cj,
               BF ? wgrad_alg_bytes(k.p) / 2.0 + 2.0 * 4.0 * k.p.Ci * k.p.Cj : wgrad_alg_bytes(k.p));
    hipLaunchKernelGGL(kern, dim3(k.ngroups * k.ntile_i * k.ntile_j), dim3(256), G::LDS, st, k);
    prof_end(st);
    HIP_TRY(hipGetLastError());
    return 0;
}

// whole 64-channel tiles: the kernels' unit; tensors with 32 channels (base-32 net) occupy half a tile
static WgradP padded_tiles(const WgradP &p)
{
    WgradP q = p;
    q.Ci = (p.Ci + 63) / 64 * 64;
    q.Cj = (p.Cj + 63) / 64 * 64;
    return q;
}

size_t wgrad_slab_need(const WgradP &p0)
{
    const WgradP p = padded_tiles(p0);
    WgradK k{};
    decompose(p, k);
    size_t a = (size_t)k.ngroups * k.pstride * sizeof(float);
    const size_t b = wgradw_slab_need(p);
    if (p.TY == 2 && p.TX == 2 && p.stride == 2) {           // the pixel-linear up-conv form may use more groups (sized for either)
        WgradK u{};
        up_decompose(p, u);
        const size_t c = (size_t)u.ngroups * u.pstride * sizeof(float);
        if (c > a) a = c;
    }
    return a > b ? a : b;
}

double wgrad_alg_flops(const WgradP &p)
{
    long cy = 0, cx = 0;
    for (int y = 0; y < p.YH; ++y)
        for (int t = 0; t < p.TY; ++t) { const int i = (y + p.oy0) * p.stride - p.xpad + t; cy += (i >= 0 && i < p.XH); }
    for (int x = 0; x < p.YW; ++x)
        for (int t = 0; t < p.TX; ++t) { const int i = (x + p.ox0) * p.stride - p.xpad + t; cx += (i >= 0 && i < p.XW); }
    return 2.0 * p.NB * (double)cy * (double)cx * p.Ci * p.Cj;
}

// algorithmic HBM bytes: the X pixels the window's taps reach and the Y window, once each; the gradient tensor, once
double wgrad_alg_bytes(const WgradP &p)
{
    int y0 = (p.ywin0 + p.oy0) * p.stride - p.xpad, y1 = (p.ywin1 - 1 + p.oy0) * p.stride - p.xpad + p.TY;
    int x0 = (p.xwin0 + p.ox0) * p.stride - p.xpad, x1 = (p.xwin1 - 1 + p.ox0) * p.stride - p.xpad + p.TX;
    y0 = y0 < 0 ? 0 : y0; x0 = x0 < 0 ? 0 : x0;
    y1 = y1 > p.XH ? p.XH : y1; x1 = x1 > p.XW ? p.XW : x1;
    double b = 0.0;
    if (y1 > y0 && x1 > x0) b += (double)p.NB * (y1 - y0) * (x1 - x0) * p.Ci * 4.0;
    b += (double)p.NB * (p.ywin1 - p.ywin0) * (p.xwin1 - p.xwin0) * p.Cj * 4.0;
    b += (double)p.TY * p.TX * p.Ci * p.Cj * 4.0;
    return b;
}

template <int TY, int TX, int S, int NSPLIT>
static int launch_wgrad_t(WgradK &k, hipStream_t st)
{
    using G = WgradGeom<TY, TX, S>;
    static bool attr_done[64] = {false}, attr_done_b[64] = {false};
    // buffer-descriptor staging needs both tensors below 2 GiB (32-bit offsets, the out-of-range marker); larger ones and
    // unet_set_lds_dma(0) take the global_load_lds instantiation
    const size_t xb = (size_t)k.p.NB * k.p.XH * k.p.XW * k.p.XC * 4, yb = (size_t)k.p.NB * k.p.YH * k.p.YW * k.p.YC * 4;
    // (not for the exact-fp32 3x3 instantiation: at its three waves per SIMD the item registers would spill)
    constexpr bool CAN_BUF = !(TY * TX == 9 && NSPLIT == 0);
    const bool buf = CAN_BUF && get_lds_dma_mode() != 0 && xb < 0x7FFFFFFFull && yb < 0x7FFFFFFFull;
    k.xbytes = buf ? (int)xb : 0; k.ybytes = buf ? (int)yb : 0;
    auto kern = buf ? wgrad_f32_kernel<TY, TX, S, NSPLIT, CAN_BUF> : wgrad_f32_kernel<TY, TX, S, NSPLIT, false>;
    if (int rc_ = ensure_dynamic_lds((const void *)kern, G::LDS, buf ? attr_done_b : attr_done)) return rc_;
    char tag[96];
    snprintf(tag, sizeof(tag), "wgrad<%d;%d;%d;split%d> Ci=%d Cj=%d Y=%dx%d win=%d parts=%d pw=%d rows=%d groups=%d", TY, TX, S, NSPLIT, k.p.Ci, k.p.Cj, k.p.YH, k.p.YW,
             k.p.ywin1 - k.p.ywin0, k.nparts, k.pw, k.rows_per_chunk, k.ngroups);
    {
        // executed: every workgroup runs its partitions' rows x pixel pairs (NSPLIT 0) / 16-pixel groups for all taps of a 64x64 tile
        const double rows = (double)k.p.NB * (k.p.ywin1 - k.p.ywin0) * k.nstrips;
        const double kpix = NSPLIT == 0 ? 2.0 * ((k.pw + 1) / 2) : 16.0 * ((k.pw + 15) / 16);
        prof_begin(PK_WGRAD, tag, st, wgrad_alg_flops(k.p), 2.0 * rows * kpix * G::T * k.p.Ci * k.p.Cj * (NSPLIT == 3 ? 3 : 1), wgrad_alg_bytes(k.p));
    }
    hipLaunchKernelGGL(kern, dim3(k.ngroups * k.ntile_i * k.ntile_j), dim3(256), G::LDS, st, k);
    prof_end(st);
    HIP_TRY(hipGetLastError());
    return 0;
}

template <int TY, int TX, int S>
static int launch_wgrad_b(WgradK &k, hipStream_t st)
{
    using G = WgradBGeom<TY, TX, S>;
    static bool attr_done[64] = {false};
    auto kern = wgrad_bf16_kernel<TY, TX, S>;
    if (int rc_ = ensure_dynamic_lds((const void *)kern, G::LDS, attr_done)) return rc_;
    const size_t xb = (size_t)k.p.NB * k.p.XH * k.p.XW * k.p.XC * 2, yb = (size_t)k.p.NB * k.p.YH * k.p.YW * k.p.YC * 2;
    ARG_CHECK(xb < 0x7FFFFFFFull && yb < 0x7FFFFFFFull, "wgrad (bf16): tensor exceeds 2 GiB");
    k.xbytes = (int)xb; k.ybytes = (int)yb;
    char tag[96];
    snprintf(tag, sizeof(tag), "wgradb<%d;%d;%d> Ci=%d Cj=%d Y=%dx%d win=%d parts=%d pw=%d rows=%d groups=%d", TY, TX, S, k.p.Ci, k.p.Cj, k.p.YH, k.p.YW,
             k.p.ywin1 - k.p.ywin0, k.nparts, k.pw, k.rows_per_chunk, k.ngroups);
    const double rows = (double)k.p.NB * (k.p.ywin1 - k.p.ywin0) * k.nstrips;
    prof_begin(PK_WGRAD, tag, st, wgrad_alg_flops(k.p), 2.0 * rows * k.pw * G::T * k.p.Ci * k.p.Cj, wgrad_alg_bytes(k.p) / 2.0 + 2.0 * G::T * k.p.Ci * k.p.Cj);
    hipLaunchKernelGGL(kern, dim3(k.ngroups * k.ntile_i * k.ntile_j), dim3(256), G::LDS, st, k);
    prof_end(st);
    HIP_TRY(hipGetLastError());
    return 0;
}

int launch_wgrad(WgradP p, hipStream_t st)
{
    ARG_CHECK(p.Ci > 0 && p.Cj > 0 && p.Ci % 32 == 0 && p.Cj % 32 == 0, "wgrad: channel counts must be multiples of 32 (Ci=%d Cj=%d)", p.Ci, p.Cj);
    if (p.math == 2) ARG_CHECK(p.Ci % 64 == 0 && p.Cj % 64 == 0, "wgrad (bf16): channel counts must be multiples of 64 (Ci=%d Cj=%d)", p.Ci, p.Cj);
    const int ci_real = p.Ci, cj_real = p.Cj;
    const bool winograd = p.math == 3 && wgradw_applicable(p);      // (decided on the tensors' own channel counts)
    p = padded_tiles(p);
    ARG_CHECK(p.XC % 4 == 0 && p.YC % 4 == 0 && p.xc0 % 4 == 0 && p.yc0 % 4 == 0, "wgrad: channel pitch/offset must be multiples of 4");
    if (p.math == 2) ARG_CHECK(p.XC % 8 == 0 && p.YC % 8 == 0 && p.xc0 % 8 == 0 && p.yc0 % 8 == 0, "wgrad (bf16): channel pitch/offset must be multiples of 8");
    ARG_CHECK(p.ywin0 >= 0 && p.ywin1 <= p.YH && p.xwin0 >= 0 && p.xwin1 <= p.YW && p.ywin0 < p.ywin1 && p.xwin0 < p.xwin1, "wgrad: bad window");
    ARG_CHECK((size_t)p.NB * p.XH * p.XW * p.XC < 0x7FFFFFFFull * 2 && (size_t)p.NB * p.YH * p.YW * p.YC < 0x7FFFFFFFull * 2, "wgrad: tensor too large");
    p.zeros = zero_page();
    if (!p.zeros) return -2;
    ARG_CHECK(p.math >= 0 && p.math <= 3, "wgrad: bad arithmetic mode %d", p.math);
    if (p.db) ARG_CHECK(p.ywin0 == 0 && p.xwin0 == 0 && p.ywin1 == p.YH && p.xwin1 == p.YW, "wgrad: fused bias gradient needs the full Y window");
    if (p.db && p.db_on_x) ARG_CHECK(p.stride == p.TY && p.stride == p.TX && p.xpad == 0, "wgrad: bias-on-X needs stride == taps (every X pixel staged exactly once)");
    if (winograd) return launch_wgradw(p, st);     // Winograd F(3x3 <- 2x2) (wgradw.hip)
    WgradK k{};
    k.p = p;
    const bool up = up_applicable(p);
    if (up) up_decompose(p, k); else decompose(p, k);
    k.ci_real = ci_real; k.cj_real = cj_real;
    const int nP = k.ngroups;                 // slabs to reduce
    const int T = p.TY * p.TX;
    const size_t need = (size_t)nP * k.pstride * sizeof(float);
    ARG_CHECK(need <= p.slab_bytes, "wgrad: slab scratch too small (%zu < %zu)", p.slab_bytes, need);
    int rc;
    const int mode = p.math == 3 ? 0 : p.math;     // mode 3 (Winograd) falls back to the exact fp32 kernel for the shapes wgradw.hip does not take
    if (up)
        rc = mode == 2 ? launch_wgrad_up<true>(k, st) : launch_wgrad_up<false>(k, st);
    else if (p.TY == 3 && p.TX == 3 && p.stride == 1)
        rc = mode == 0 ? launch_wgrad_t<3, 3, 1, 0>(k, st) : mode == 1 ? launch_wgrad_t<3, 3, 1, 3>(k, st) : launch_wgrad_b<3, 3, 1>(k, st);
    else if (p.TY == 2 && p.TX == 2 && p.stride == 2) {
        // (bf16 tensors: only the pixel-linear kernel - it needs what the bf16 row-walking kernel needed too, tensors below 2 GiB)
        if (mode == 2) { set_error("wgrad (bf16): the up-conv weight gradient needs tensors below 2 GiB, the full window and buffer-descriptor LDS-DMA"); return -4; }
        rc = mode == 0 ? launch_wgrad_t<2, 2, 2, 0>(k, st) : launch_wgrad_t<2, 2, 2, 3>(k, st);
    }
    else { set_error("wgrad: unsupported taps %dx%d stride %d", p.TY, p.TX, p.stride); return -4; }
    if (rc) return rc;
    const int ndb = p.db ? (p.db_on_x ? ci_real : cj_real) : 0;
    const size_t total = (size_t)T * p.Ci * p.Cj + ndb;
    size_t blocks = (total + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    ARG_CHECK(k.pstride % 4 == 0 && ndb % 4 == 0 && ((uintptr_t)p.slab & 15) == 0, "wgrad: the reduce needs 16-byte aligned slabs");
    prof_begin(PK_REDUCE, "wgrad_reduce", st, 0.0, 0.0, (double)nP * k.pstride * 4.0 + (double)total * 4.0);
    if (nP >= 64)
        hipLaunchKernelGGL(wgrad_reduce_kernel<16>, dim3((unsigned)blocks), dim3(1024), 0, st, p.slab, nP, k.pstride, T, p.Ci, p.Cj, ci_real, cj_real, p.out, p.si, p.sj, p.st, p.db, ndb);
    else
        hipLaunchKernelGGL(wgrad_reduce_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, st, p.slab, nP, k.pstride, T, p.Ci, p.Cj, ci_real, cj_real, p.out, p.si, p.sj, p.st, p.db, ndb);
    prof_end(st);
    HIP_TRY(hipGetLastError());
    return 0;
}

}  // namespace unet
