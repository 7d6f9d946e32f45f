// wino.hip — fp32 Winograd F(2x2,3x3) for the stride-1 3x3 convolutions of the path (forward and dgrad;
// network.py:131-188 and their autograd), on the gfx950 matrix cores.
//
//   Y = A^T [ (G g G^T) (.) (B^T d B) ] A            per 2x2 output tile, 4x4 input patch d, 3x3 filter g
//   M[xi][tile][n] = sum_c V[xi][tile][c] * U[xi][c][n]       xi = 0..15: 16 small GEMMs, 16 instead of 36
//                                                              multiplies per (tile, c, n) -> 2.25x fewer MFMA flops
// All arithmetic is fp32 (v_mfma_f32_16x16x4_f32 + fp32 adds); only the evaluation order differs from the
// direct correlation, which costs ~1e-6 relative (tests/test_ops_gpu.py pins it against the oracle).
//
// Work decomposition — one 512-thread workgroup (8 waves, 2 per SIMD) = 64 Winograd tiles x 64 output channels:
//   wave (wm, wn): tiles 16*wm .. +15  x  channels 32*wn .. +31, all 16 xi  -> 16*2 accumulators of 16x16 (128 VGPRs)
//   * tiles are numbered linearly over (image, tile row, tile column): no 2-D edge waste, only the last workgroup
//     of a launch is ragged;
//   * per K step (8 channels) the workgroup stages by LDS-DMA  (a) every tile's private 4x4 pixel patch
//     (64 x 16 x 32 B = 32 KiB) and (b) the U block of its 64 channels (16 x 64 x 8 x 4 B = 32 KiB, stored in
//     global memory already in LDS/fragment order by wino_transform_kernel), double buffered, one barrier per step;
//   * the INPUT transform is done in registers: lane (tile, kg) reads its 16 pixels x 2 channels (ds_read_b64,
//     conflict-free layout), 32 packed adds give V[xi] for those 2 channels = exactly the A operands of the 16x16x4
//     MFMAs (k index = kg); V never exists in memory;
//   * the OUTPUT transform is done in registers too: the accumulator layout gives a lane (channel, 4 tiles) for
//     every xi; 24 adds per tile yield the 2x2 outputs, which go through LDS to 256-B-contiguous float4 stores with
//     the usual fused epilogue (bias, +add, ReLU / deferred-ReLU window, ReLU' mask).
//
// Why this shape: accumulators are 16x the output tile, so the tile is small (64x64) and the staged bytes per MFMA
// cycle are the limit — 64 KiB per 4096 cycles = 16 B/clk/CU through the L1/LDS-DMA path (of ~64), ~10 B/clk from
// L2 (patch overlap hits L1); LDS reads 47 B/clk (of 256).
#include "common.hpp"
#include "igemm_epilogue.hpp"
#include <cstdio>
#include <cstdlib>

namespace unet {

#define GLDS16(gptr, lptr)                                                                    \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gptr),  \
                                     (__attribute__((address_space(3))) void *)(lptr), 16, 0, 0)

typedef float f32x2 __attribute__((ext_vector_type(2)));

struct WinoP {
    IgemmP p;
    const float *U;          // transformed filters, block order (see wino_transform_kernel)
    int tiles_x, tiles_y;    // Winograd tiles per image
    int MT;                  // NB * tiles_y * tiles_x
    int nsteps;              // total channels / 8
    FastDiv d_tpi, d_tx;
};

constexpr int WINO_STAGE = 65536;      // 32 KiB patches + 32 KiB U
constexpr int WINO_LDS = 2 * WINO_STAGE;

// --------------------------------------------------------------------------------------------------------
// U = G g G^T for every (n, c), written in the order the main kernel stages and reads it:
//   block (nt = n/64, step = c/8)  [32 KiB]:  piece (n16 = (n/16)%4, xg = xi/4, h = c%2) [1 KiB]:
//   lane (kg = (c%8)/2, nl = n%16) [16 B]: xi%4
// Source: packed igemm weights wt[n][ldw], column = base(src) + tap*nch(src) + c  (pack_conv_fwd / pack_conv_dgrad).
// --------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void wino_transform_kernel(const float *__restrict__ wt, int ldw, int Nn, int nch0, int nch1,
                                                             float *__restrict__ U)
{
    const int Kc = nch0 + nch1;
    const size_t total = (size_t)Nn * Kc;
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int n = (int)(idx / Kc), c = (int)(idx - (size_t)n * Kc);
    const float *row = wt + (size_t)n * ldw;
    int base, nch, cc;
    if (c < nch0) { base = 0; nch = nch0; cc = c; } else { base = 9 * nch0; nch = nch1; cc = c - nch0; }
    float g[3][3];
#pragma unroll
    for (int t = 0; t < 9; ++t) g[t / 3][t % 3] = row[base + t * nch + cc];
    float r[4][3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        r[0][j] = g[0][j];
        r[1][j] = 0.5f * (g[0][j] + g[1][j] + g[2][j]);
        r[2][j] = 0.5f * (g[0][j] - g[1][j] + g[2][j]);
        r[3][j] = g[2][j];
    }
    float u[16];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        u[4 * i + 0] = r[i][0];
        u[4 * i + 1] = 0.5f * (r[i][0] + r[i][1] + r[i][2]);
        u[4 * i + 2] = 0.5f * (r[i][0] - r[i][1] + r[i][2]);
        u[4 * i + 3] = r[i][2];
    }
    const int nsteps = Kc >> 3;
    const int nt = n >> 6, n16 = (n >> 4) & 3, nl = n & 15;
    const int step = c >> 3, kk = c & 7, kg = kk >> 1, h = kk & 1;
    float *blk = U + ((size_t)nt * nsteps + step) * 8192;
#pragma unroll
    for (int xg = 0; xg < 4; ++xg) {
        f32x4 v = {u[4 * xg], u[4 * xg + 1], u[4 * xg + 2], u[4 * xg + 3]};
        *(f32x4 *)(blk + ((n16 * 4 + xg) * 2 + h) * 256 + (kg * 16 + nl) * 4) = v;
    }
}

int wino_transform(const float *wt, int ldw, int Nn, int nch0, int nch1, float *U, hipStream_t st)
{
    ARG_CHECK(wt && U && Nn % 64 == 0 && nch0 % 8 == 0 && nch1 % 8 == 0 && nch0 > 0, "wino_transform: bad shape");
    const size_t total = (size_t)Nn * (nch0 + nch1);
    hipLaunchKernelGGL(wino_transform_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, wt, ldw, Nn, nch0, nch1, U);
    HIP_TRY(hipGetLastError());
    return 0;
}

// --------------------------------------------------------------------------------------------------------
template <int DBG>
__global__ __launch_bounds__(512, 1) void wino_f32_kernel(const WinoP k)
{
    const IgemmP &p = k.p;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l15 = lane & 15, kg = lane >> 4;

    // XCD-aware order: every XCD gets a contiguous run of logical workgroups; M tiles of one N tile are neighbours
    // (they share the 32 KiB/step U stream, which is 3x the unique patch bytes).
    int logical;
    {
        const int nblk = gridDim.x, q = nblk >> 3, r = nblk & 7, xcd = blockIdx.x & 7;
        logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (blockIdx.x >> 3);
    }
    const int nt = logical / p.mtiles, mt = logical - nt * p.mtiles;
    const int T0 = mt * 64, n0 = nt * 64;

    // ---- DMA role.  Patch image per stage: [wm 4][q 16][half 2][tile 16][16 B]; one instruction = 2 q's of one wm.
    // This wave fills wm = wave>>1, q = 8*(wave&1) + 2*jj + (lane>>5)  ->  qy = 2*(wave&1) + (jj>>1), qx = 2*(jj&1) + (lane>>5).
    int d_img, d_ty, d_tx;
    {
        int T = T0 + (wave >> 1) * 16 + l15;
        T = T < k.MT ? T : k.MT - 1;
        d_img = fdiv(T, k.d_tpi);
        const int rem = T - d_img * (k.tiles_x * k.tiles_y);
        d_ty = fdiv(rem, k.d_tx);
        d_tx = rem - d_ty * k.tiles_x;
    }
    const int d_half4 = ((lane >> 4) & 1) * 4;
    int poff[4];
    const float *sp = nullptr;
    int snch = 0;
    auto setup_source = [&](int si) {
        const GSrc &g = p.src[si];
        sp = g.p; snch = g.nch;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int qy = 2 * (wave & 1) + (jj >> 1), qx = 2 * (jj & 1) + (lane >> 5);
            const int iy = 2 * d_ty + p.oy0 - g.pad + qy, ix = 2 * d_tx + p.ox0 - g.pad + qx;
            const bool ok = (unsigned)iy < (unsigned)g.H && (unsigned)ix < (unsigned)g.W;
            poff[jj] = ok ? ((d_img * g.H + iy) * g.W + ix) * g.C + g.c0 + d_half4 : -1;
        }
    };
    const float *ublk = k.U + (size_t)nt * k.nsteps * 8192 + (4 * wave) * 256 + lane * 4;
    auto stage = [&](int buf, int kc, int step) {
        unsigned char *pb = smem + buf * WINO_STAGE + ((wave >> 1) * 16 + 8 * (wave & 1)) * 512;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const float *g = poff[jj] >= 0 ? sp + (poff[jj] + kc) : p.zeros;
            GLDS16(g, pb + jj * 1024);
        }
        unsigned char *ub = smem + buf * WINO_STAGE + 32768 + (4 * wave) * 1024;
        const float *us = ublk + (size_t)step * 8192;
#pragma unroll
        for (int i = 0; i < 4; ++i) GLDS16(us + i * 256, ub + i * 1024);
    };

    f32x4 acc[16][2];
#pragma unroll
    for (int x = 0; x < 16; ++x)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[x][j][r] = 0.f;

    const int a_rd = (wm * 16) * 512 + (kg >> 1) * 256 + l15 * 16 + (kg & 1) * 8;
    const int b_rd = 32768 + (2 * wn) * 8192 + lane * 16;

    int s = 0, kc = 0;
    setup_source(0);
    stage(0, 0, 0);
    __syncthreads();
    for (int st = 0; st < k.nsteps; ++st) {
        const int cur = st & 1;
        if (st + 1 < k.nsteps) {
            kc += 8;
            if (kc == snch) { kc = 0; ++s; setup_source(s); }
            if (DBG != 1) stage(cur ^ 1, kc, st + 1);
        }
        const unsigned char *sb = smem + cur * WINO_STAGE;
        f32x2 d[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) d[q] = *(const f32x2 *)(sb + a_rd + (DBG == 3 ? 0 : q * 512));
        f32x2 t[16], v[16];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            t[j] = d[j] - d[8 + j];
            t[4 + j] = d[4 + j] + d[8 + j];
            t[8 + j] = d[8 + j] - d[4 + j];
            t[12 + j] = d[4 + j] - d[12 + j];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[4 * i + 0] = t[4 * i] - t[4 * i + 2];
            v[4 * i + 1] = t[4 * i + 1] + t[4 * i + 2];
            v[4 * i + 2] = t[4 * i + 2] - t[4 * i + 1];
            v[4 * i + 3] = t[4 * i + 1] - t[4 * i + 3];
        }
#pragma unroll
        for (int xg = 0; xg < 4; ++xg)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const f32x4 b0 = *(const f32x4 *)(sb + b_rd + (xg * 2 + h) * 1024);
                const f32x4 b1 = *(const f32x4 *)(sb + b_rd + 8192 + (xg * 2 + h) * 1024);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (DBG == 2) { acc[4 * xg + e][0][0] += v[4 * xg + e][h] * b0[e]; acc[4 * xg + e][1][0] += v[4 * xg + e][h] * b1[e]; continue; }
                    acc[4 * xg + e][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[4 * xg + e][h], b0[e], acc[4 * xg + e][0], 0, 0, 0);
                    acc[4 * xg + e][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[4 * xg + e][h], b1[e], acc[4 * xg + e][1], 0, 0, 0);
                }
            }
        __syncthreads();
    }

    // ---- epilogue.  LDS: out[256 rows = tile*4 + 2*py + px][64 n] floats (64 KiB) | rowoff[256] | flags[256]
    float *outp = (float *)smem;
    unsigned *rowoff = (unsigned *)(smem + 65536);
    unsigned char *rflag = smem + 65536 + 1024;      // bit0: row outside the output domain, bit1: inside the deferred-ReLU window
    if (tid < 256) {
        const int tl = tid >> 2, py = (tid >> 1) & 1, px = tid & 1;
        int T = T0 + tl;
        const bool tok = T < k.MT;
        T = tok ? T : k.MT - 1;
        const int img = fdiv(T, k.d_tpi);
        const int rem = T - img * (k.tiles_x * k.tiles_y);
        const int ty = fdiv(rem, k.d_tx);
        const int tx = rem - ty * k.tiles_x;
        int oy = 2 * ty + py, ox = 2 * tx + px;
        const bool ok = tok && oy < p.OH && ox < p.OW;
        oy = oy < p.OH ? oy : p.OH - 1; ox = ox < p.OW ? ox : p.OW - 1;
        unsigned off;
        if (p.scatter == 2) off = (unsigned)((img * p.DH + oy + p.dwy0) * p.DW + ox + p.dwx0) * (unsigned)p.DC;
        else off = (unsigned)((img * p.OH + oy) * p.OW + ox) * (unsigned)p.DC;
        rowoff[tid] = off;
        const bool inwin = (p.rw1 > p.rw0) && oy >= p.rw0 && oy < p.rw1 && ox >= p.rw0 && ox < p.rw1;
        rflag[tid] = (ok ? 0 : 1) | (inwin ? 2 : 0);
    }
#pragma unroll
    for (int nn = 0; nn < 2; ++nn) {
        const int n = wn * 32 + nn * 16 + l15;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float s0[4], s1[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                s0[j] = acc[j][nn][r] + acc[4 + j][nn][r] + acc[8 + j][nn][r];
                s1[j] = acc[4 + j][nn][r] - acc[8 + j][nn][r] - acc[12 + j][nn][r];
            }
            const int tile = wm * 16 + 4 * kg + r;
            float *o = outp + (tile * 4) * 64 + n;
            o[0] = s0[0] + s0[1] + s0[2];
            o[64] = s0[1] - s0[2] - s0[3];
            o[128] = s1[0] + s1[1] + s1[2];
            o[192] = s1[1] - s1[2] - s1[3];
        }
    }
    __syncthreads();

    const int c4 = tid & 15;
    const int ncol = n0 + 4 * c4;
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (p.bias) {
#pragma unroll
        for (int c = 0; c < 4; ++c) { const int n = ncol + c; bv[c] = p.bias[p.cout ? n % p.cout : n]; }
    }
    const bool relu_win = p.rw1 > p.rw0;
#pragma unroll
    for (int it = 0; it < 8; it += 4) {
        f32x4 v[4];
        size_t o[4];
        unsigned char fl[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int row = (it + u) * 32 + (tid >> 4);
            v[u] = *(const f32x4 *)(outp + row * 64 + 4 * c4) + bv;
            o[u] = (size_t)rowoff[row] + (size_t)(p.dn0 + ncol);
            fl[u] = rflag[row];
        }
        if (p.add) {
            f32x4 t[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) t[u] = *(const f32x4 *)(p.add + o[u]);
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] += t[u];
        }
        if (p.relu) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const bool defer = relu_win && (fl[u] & 2);
#pragma unroll
                for (int c = 0; c < 4; ++c) v[u][c] = (v[u][c] > 0.f || defer) ? v[u][c] : 0.f;
            }
        }
        if (p.mask) {
            f32x4 t[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) t[u] = *(const f32x4 *)(p.mask + o[u]);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int c = 0; c < 4; ++c) v[u][c] = t[u][c] > 0.f ? v[u][c] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (!(fl[u] & 1)) *(f32x4 *)(p.dst + o[u]) = v[u];
    }
}

double igemm_alg_flops(const IgemmP &p);

bool wino_applicable(const IgemmP &p)
{
    if (p.T != 9 || p.TX != 3 || p.stride != 1 || p.scatter == 1) return false;
    if (p.Nn % 64 != 0) return false;
    for (int i = 0; i < p.nsrc; ++i)
        if (p.src[i].nch % 8 != 0) return false;
    return true;
}

size_t wino_u_floats(int Kc, int Nn) { return (size_t)16 * Kc * Nn; }

// p must have passed launch_igemm's argument checks (launch_igemm calls this)
int launch_wino(const IgemmP &p, const float *U, hipStream_t st)
{
    static bool attr_done[64] = {false};
    static const int dbg = [] { const char *e = getenv("UNET_WINO_DBG"); return e ? atoi(e) : 0; }();
    auto kern = dbg == 1 ? wino_f32_kernel<1> : dbg == 2 ? wino_f32_kernel<2> : dbg == 3 ? wino_f32_kernel<3> : wino_f32_kernel<0>;
    static bool attr_done1[64] = {false}, attr_done2[64] = {false}, attr_done3[64] = {false};
    if (int rc_ = ensure_dynamic_lds((const void *)kern, WINO_LDS, dbg == 1 ? attr_done1 : dbg == 2 ? attr_done2 : dbg == 3 ? attr_done3 : attr_done)) return rc_;
    WinoP q;
    q.p = p;
    q.U = U;
    q.tiles_x = cdiv(p.OW, 2);
    q.tiles_y = cdiv(p.OH, 2);
    q.MT = p.NB * q.tiles_x * q.tiles_y;
    int kc = 0;
    for (int i = 0; i < p.nsrc; ++i) kc += p.src[i].nch;
    q.nsteps = kc / 8;
    q.d_tpi = make_fastdiv((unsigned)(q.tiles_x * q.tiles_y));
    q.d_tx = make_fastdiv((unsigned)q.tiles_x);
    q.p.mtiles = cdiv(q.MT, 64);
    q.p.ntiles = p.Nn / 64;
    char tag[96];
    snprintf(tag, sizeof(tag), "wino M=%d N=%d Kd=%d nsrc=%d tiles=%d", p.M, p.Nn, p.Kd, p.nsrc, q.MT);
    prof_begin(0, igemm_alg_flops(p), st, tag);
    hipLaunchKernelGGL(kern, dim3(q.p.mtiles * q.p.ntiles), dim3(512), WINO_LDS, st, q);
    prof_end(st);
    HIP_TRY(hipGetLastError());
    return 0;
}

}  // namespace unet
