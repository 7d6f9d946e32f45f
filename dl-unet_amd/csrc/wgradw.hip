// wgradw.hip — weight gradient of the stride-1 3x3 convolutions as fp32 Winograd F(3x3 <- 4x4 input, 2x2 dz) on the
// gfx950 matrix cores (autograd backward of network.py:131-188's convs; the direct version is wgrad.hip).
//
//   forward  Y = A^T [ (G g G^T) (.) (B^T d B) ] A     per 2x2 output tile
//   =>       dg = G^T [ sum_tiles (B^T d B) (.) (A dY A^T) ] G
//   dU[xi][ci][cj] = sum_tiles V[xi][tile][ci] * Z[xi][tile][cj]      16 GEMMs whose K dimension is the tile index:
//                                                                     16 instead of 36 multiplies per (tile, ci, cj)
// All arithmetic is fp32 (v_mfma_f32_16x16x4_f32 + fp32 adds).
//
// One 512-thread workgroup (8 waves, 2 per SIMD) owns a 64(ci) x 64(cj) channel tile for all 16 xi — 128 accumulator
// VGPRs per wave (wave = 16 ci x 32 cj), the same budget as wino.hip — and a contiguous range of tile slots (split K).
// Tile slots are numbered linearly over (image, tile row, tile column 0..TX) where column TX is a GHOST tile: it only
// carries the two pixel columns right of the row's last tile and has dz = 0.  With it every tile's pixel columns 2,3
// are simply the next slot's columns 0,1, so a K step of 16 slots stages [8 (qy,par) rows][4 channel blocks][16 slots]
// [64 B] of input = 32 KiB (+ 2 KiB for the slot after the step's last one) and [4 px][4 blocks][16 slots][64 B] of dz
// = 16 KiB by LDS-DMA, 6 (7) instructions per wave, into a 3-deep ring; per step a wave runs 128 MFMAs.  Both
// transforms happen in registers, two tiles per lane in packed fp32 (v_pk_add_f32):
//   lane (channel l15, k = kg) reads its tiles' pixels channel-wise (ds_read_b32, 64 lanes = 256 contiguous bytes),
//   V = B^T d B (32 packed adds), Z = A dY A^T up to signs (12 packed adds; the signs are applied in the reduce),
//   and V[xi] / Z[xi] are exactly the A / B operands of the 16x16x4 MFMA (M = ci, N = cj, K = 4 tiles).
// Each workgroup applies the signs and G^T . G to its partial dU itself; the partial 3x3 slabs are reduced in a fixed order by
// wgradw_reduce_kernel, which writes the gradient in the caller's layout; the fused bias gradient is the sum of the dz values a wave reads anyway.
#include "common.hpp"
#include "igemm_epilogue.hpp"
#include <cstdio>
#include <cstdlib>

namespace unet {

#define GLDS16(gptr, lptr)                                                                    \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gptr),  \
                                     (__attribute__((address_space(3))) void *)(lptr), 16, 0, 0)

typedef float f32x2 __attribute__((ext_vector_type(2)));

struct WgradWK {
    WgradP p;
    int TXn, TXg, TYn;        // real tiles per row, slots per row (TXn + 1), tile rows
    int wy0, wx0;             // Y-domain origin of tile (0,0)
    int NTg;                  // NB * TYn * TXg slots
    int nsteps, steps_per;    // K steps (16 slots) in total / per workgroup
    int ntile_i, ntile_j, nsplit;
    FastDiv d_spi, d_txg;     // slots per image, slots per row
    size_t pstride;           // floats per slab
    int xbytes, ybytes;       // buffer-descriptor sizes of X and Y (bytes); the buffer path needs both below 2 GiB
};

#ifndef WW_DBG
#define WW_DBG 0
#endif
constexpr int WW_PATCH = 32768, WW_EXTRA = 2048, WW_DY = 16384;
constexpr int WW_STAGE = WW_PATCH + WW_EXTRA + WW_DY;       // 51200
constexpr int WW_NST = 3;
constexpr int WW_LDS = WW_NST * WW_STAGE;                   // 153600
constexpr int WW_SLAB = 9 * 64 * 64 + 64;                   // G^T dU G partial (3x3 taps) + bias partial (floats)

template <int N> __device__ __forceinline__ void ww_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// plain float2 arithmetic (selected as v_pk_add_f32; inline asm would hide the VALU->MFMA hazards from the compiler)
__device__ __forceinline__ f32x2 wpk_add(f32x2 a, f32x2 b) { return a + b; }
__device__ __forceinline__ f32x2 wpk_sub(f32x2 a, f32x2 b) { return a - b; }

template <bool BUF>
__global__ __launch_bounds__(512, 1) void wgradw_f32_kernel(const WgradWK k)
{
    const WgradP &p = k.p;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wi = wave >> 1, wj = wave & 1;
    const int l15 = lane & 15, kg = lane >> 4;

    // XCD-aware order: consecutive logical workgroups = the channel tiles of one K range (they read the same pixels)
    int logical;
    {
        const int nblk = gridDim.x, q = nblk >> 3, r = nblk & 7, xcd = blockIdx.x & 7;
        logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (blockIdx.x >> 3);
    }
    const int ntile = k.ntile_i * k.ntile_j;
    const int part = logical / ntile, tile = logical - part * ntile;
    const int it = tile / k.ntile_j, jt = tile - it * k.ntile_j;
    const int s0 = part * k.steps_per;
    int s1 = s0 + k.steps_per;
    s1 = s1 < k.nsteps ? s1 : k.nsteps;
    const int ns = s1 > s0 ? s1 - s0 : 0;

    // ---- DMA role.  lane -> (slot = lane>>2, 16-B chunk = lane&3); this wave stages channel block cb = wave&3 of
    // patch rows (qy = 0..3, par = (wave>>2)&1) and of dz pixels (py = 0..1, px = (wave>>2)&1); waves 0,1 also stage the
    // "extra" slot (the one after the step's last): lane -> (row = (lane>>4) + 4*wave, cb = (lane>>2)&3, chunk).
    const int d_cb = wave & 3, d_par = (wave >> 2) & 1;
    const int spi = k.TYn * k.TXg;

    // BUF: LDS-DMA through buffer descriptors (buffer_load_dwordx4 ... offen lds): 32-bit byte offsets, the row offset as
    // scalar offset, pixels outside the tensors as offsets beyond num_records (the range check returns zeros).
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void *)p.X, 0, BUF ? k.xbytes : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc((void *)p.Y, 0, BUF ? k.ybytes : 0, 0x00020000);
    constexpr int OOB = (int)0x80000000;
    auto stage = [&](int buf, int step) {
        unsigned char *sb = smem + buf * WW_STAGE;
        // the lane-derived roles are recomputed per call (a handful of VALU) instead of living in VGPRs across the loop:
        // the opaque copy keeps the compiler from hoisting them
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const int d_slot = ln >> 2, d_chunk = ln & 3;
        const int xch = p.xc0 + it * 64 + d_cb * 16 + d_chunk * 4;
        const int ych = p.yc0 + jt * 64 + d_cb * 16 + d_chunk * 4;
        // the step's first slot is decomposed once, on the scalar unit (uniform); a lane then only walks d_slot slots on
        const int Tb = step * 16;
        const bool tok = Tb + d_slot < k.NTg;
        const int img_b = fdiv(Tb, k.d_spi);
        const int rem_b = Tb - img_b * spi;
        const int ty_b = fdiv(rem_b, k.d_txg);
        int img = img_b, ty = ty_b, txg = rem_b - ty_b * k.TXg + d_slot;
        while (txg >= k.TXg) { txg -= k.TXg; ++ty; }
        while (ty >= k.TYn) { ty -= k.TYn; ++img; }
        img = img < p.NB ? img : p.NB - 1;
        // input patch: pixel (qy, par) of the slot
        {
            const int iy0 = (k.wy0 + 2 * ty + p.oy0) - p.xpad, ix = (k.wx0 + 2 * txg + p.ox0) - p.xpad + d_par;
            const bool xok = tok && (unsigned)ix < (unsigned)p.XW;
            const int base = ((img * p.XH + iy0) * p.XW + ix) * p.XC + xch;
#pragma unroll
            for (int qy = 0; qy < 4; ++qy) {
                const bool ok = xok && (unsigned)(iy0 + qy) < (unsigned)p.XH;
                unsigned char *dst = sb + ((qy * 2 + d_par) * 4 + d_cb) * 1024;
                if (BUF) {
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (__attribute__((address_space(3))) void *)dst, 16, ok ? (base + qy * p.XW * p.XC) * 4 : OOB, 0, 0, 0);      // (base alone can be negative under zero padding)
                } else {
                    const float *g = ok ? p.X + (base + qy * p.XW * p.XC) : p.zeros;
                    GLDS16(g, dst);
                }
            }
        }
        // dz: pixel (py, px = d_par) of the slot; ghost slots and pixels outside the tensor read zeros
        {
            const int yy0 = k.wy0 + 2 * ty, xx = k.wx0 + 2 * txg + d_par;
            const bool xok = tok && txg < k.TXn && xx < p.YW;
            const int base = ((img * p.YH + yy0) * p.YW + xx) * p.YC + ych;
#pragma unroll
            for (int py = 0; py < 2; ++py) {
                const bool ok = xok && yy0 + py < p.YH;
                unsigned char *dst = sb + WW_PATCH + WW_EXTRA + ((py * 2 + d_par) * 4 + d_cb) * 1024;
                if (BUF) {
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_y, (__attribute__((address_space(3))) void *)dst, 16, ok ? (base + py * p.YW * p.YC) * 4 : OOB, 0, 0, 0);
                } else {
                    const float *g = ok ? p.Y + (base + py * p.YW * p.YC) : p.zeros;
                    GLDS16(g, dst);
                }
            }
        }
        if (wave < 2) {
            // extra slot = slot 16 of the step: patch rows only.  (It is slot 0 of the next step; if that starts a new
            // tile row the current row's last slot was a ghost and nobody reads the extra slot.)
            const int Te = step * 16 + 16;
            const bool eok = Te < k.NTg;
            const int Tec = eok ? Te : k.NTg - 1;
            const int eimg = fdiv(Tec, k.d_spi);
            const int erem = Tec - eimg * spi;
            const int ety = fdiv(erem, k.d_txg);
            const int etxg = erem - ety * k.TXg;
            const int e_row = (ln >> 4) + 4 * wave, e_cb = (ln >> 2) & 3;
            const int e_xch = p.xc0 + it * 64 + e_cb * 16 + d_chunk * 4;
            const int qy = e_row >> 1, par = e_row & 1;
            const int iy = (k.wy0 + 2 * ety + p.oy0) - p.xpad + qy, ix = (k.wx0 + 2 * etxg + p.ox0) - p.xpad + par;
            const bool ok = eok && (unsigned)iy < (unsigned)p.XH && (unsigned)ix < (unsigned)p.XW;
            const int eoff = ((eimg * p.XH + iy) * p.XW + ix) * p.XC + e_xch;
            if (BUF) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (__attribute__((address_space(3))) void *)(sb + WW_PATCH + wave * 1024), 16, ok ? eoff * 4 : OOB, 0, 0, 0);
            } else {
                const float *g = ok ? p.X + eoff : p.zeros;
                GLDS16(g, sb + WW_PATCH + wave * 1024);
            }
        }
    };
    // wait until only the newest batch (one stage) of this wave is still in flight, or nothing
    auto wait_landed = [&](bool one_in_flight) {
        if (!one_in_flight) ww_wait_vmcnt<0>();
        else if (wave < 2) ww_wait_vmcnt<7>();
        else ww_wait_vmcnt<6>();
    };

    f32x4 acc[16][2];
#pragma unroll
    for (int x = 0; x < 16; ++x)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[x][j][r] = 0.f;
    f32x2 dbacc[2] = {{0.f, 0.f}, {0.f, 0.f}};

    // ---- read role: lane (channel l15, k = kg).  Tile pairs (kg, kg+4) and (kg+8, kg+12): .x / .y of the packed values.
    // patch[(row)*4096 + cb*1024 + slot*64 + ch*4], row = 2*qy + par; columns 2,3 of slot t are columns 0,1 of slot t+1;
    // slot 16 lives in the extra region [row][cb][64 B].
    const int a_base = wi * 1024 + l15 * 4;
    const int e_base = WW_PATCH + wi * 64 + l15 * 4;
    const bool last = kg == 3;                      // this lane's tile kg+12 = 15: its neighbour is the extra slot
    const int y_base = WW_PATCH + WW_EXTRA + l15 * 4;

    // The loop below is written in CHUNKS separated by scheduling barriers — a few LDS reads, a few packed adds, a few MFMAs —
    // in the order they are to be issued: left to itself the compiler lumps a transform's 32 packed adds together and waits
    // for every read right behind its issue, and then a wave feeds the matrix pipe for two thirds of the time only.
#define WW_SB() __builtin_amdgcn_sched_barrier(0)
    // raw pixels of column j of the 4x4 patches of tile pair pr (.x tile kg + 8 pr, .y tile +4): 8 ds_read_b32
    // (addresses: one per-lane base per call + compile-time offsets that fit the instructions' offset fields; the extra slot of
    // the lanes with kg = 3 through a per-lane base and row stride instead of a select per read)
    const int lv = a_base + kg * 64;
    const int lx = last ? e_base : lv + 9 * 64 + 256;       // pair 1, columns 2,3, .y: tile kg + 13 or the extra slot
    const int lxs = last ? 512 : 8192;                      // ... and its stride per qy (two rows)
    const int ly = y_base + 2 * wj * 1024 + kg * 64;
    auto v_read = [&](int buf, int pr, int j, f32x2 (&a)[4]) {
        if (WW_DBG & 4) { WW_SB(); return; }
        const unsigned char *pv = smem + buf * WW_STAGE + lv;
        if (j < 2 || pr == 0) {
#pragma unroll
            for (int qy = 0; qy < 4; ++qy) {
                const unsigned char *src = pv + ((2 * qy + (j & 1)) * 4096 + (8 * pr + (j >> 1)) * 64);
                a[qy][0] = *(const float *)(src);
                a[qy][1] = *(const float *)(src + 256);
            }
        } else {
            const unsigned char *px = smem + buf * WW_STAGE + lx + (j & 1) * (lxs >> 1);
#pragma unroll
            for (int qy = 0; qy < 4; ++qy) {
                a[qy][0] = *(const float *)(pv + ((2 * qy + (j & 1)) * 4096 + 9 * 64));
                a[qy][1] = *(const float *)(px + qy * lxs);
            }
        }
        WW_SB();
    };
    // V = B^T d B: column pass of column j (4 packed adds) ...
    auto v_col = [&](int j, const f32x2 (&a)[4], f32x2 (&v)[16]) {
        if (WW_DBG & 8) { WW_SB(); return; }
        v[j] = wpk_sub(a[0], a[2]);
        v[4 + j] = wpk_add(a[1], a[2]);
        v[8 + j] = wpk_sub(a[2], a[1]);
        v[12 + j] = wpk_sub(a[1], a[3]);
        WW_SB();
    };
    // ... and row pass of row i (4 packed adds)
    auto v_row = [&](int i, f32x2 (&v)[16]) {
        if (WW_DBG & 8) { WW_SB(); return; }
        const f32x2 t0 = v[4 * i], t1 = v[4 * i + 1], t2 = v[4 * i + 2], t3 = v[4 * i + 3];
        v[4 * i + 0] = wpk_sub(t0, t2);
        v[4 * i + 1] = wpk_add(t1, t2);
        v[4 * i + 2] = wpk_sub(t2, t1);
        v[4 * i + 3] = wpk_sub(t1, t3);
        WW_SB();
    };
    // dY = [[a, b], [c, d]] = d[0..3] of tile pair pr, cj block c: 8 ds_read_b32
    auto z_read = [&](int buf, int pr, int c, f32x2 (&d)[4]) {
        if (WW_DBG & 4) { WW_SB(); return; }
        const unsigned char *ys = smem + buf * WW_STAGE + ly;
#pragma unroll
        for (int px = 0; px < 4; ++px) {
            d[px][0] = *(const float *)(ys + (px * 4096 + c * 1024 + pr * 512));
            d[px][1] = *(const float *)(ys + (px * 4096 + c * 1024 + pr * 512 + 256));
        }
        WW_SB();
    };
    // One GROUP = the 32 MFMAs of (tile pair, cj block): V[xi] (registers, transformed) x Z[xi], where Z = A dY A^T up to signs
    // (applied in the reduce) is NOT kept as a 16-entry set: a lane holds the raw dY = [[d0, d1], [d2, d3]] (8 registers) and forms
    // the four Z[xi] of a block of xi right in front of their MFMAs (12 packed adds per group) — 24 registers less per operand
    // set, which is what lets the next operands' reads and transform ride in the shadow of the groups without spilling.
    // MFMA order inside a block: xi 4b..4b+3 for tile half 0, then for half 1 (dependent accumulations four MFMAs apart).
    // Under every group: dn <- raw dY(bufz, prz, cz), the NEXT group's.  RD: also the 32 raw patch reads of V(bufv, prv) into r
    // (used a whole group later: the CU's eight waves run in step and send their reads at the same moments, so a read takes
    // several hundred cycles to come back); XF: transform r into vn.
    auto group = [&](const f32x2 (&v)[16], const f32x2 (&d)[4], int c, bool RD, bool XF, int bufz, int prz, int cz, f32x2 (&dn)[4],
                     int bufv, int prv, f32x2 (&r)[4][4], f32x2 (&vn)[16]) {
        f32x2 zt[4];
        auto quad = [&](int b, int h) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                acc[4 * b + e][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[4 * b + e][h], zt[e][h], acc[4 * b + e][c], 0, 0, 0);
            WW_SB();
        };
        // block 0: Z0..3 = d0, d0+d1, d0-d1, d1 (sign(3) = -)
        zt[0] = d[0]; zt[1] = wpk_add(d[0], d[1]); zt[2] = wpk_sub(d[0], d[1]); zt[3] = d[1];
        WW_SB();
        quad(0, 0);
        z_read(bufz, prz, cz, dn);
        quad(0, 1);
        if (RD) v_read(bufv, prv, 0, r[0]);
        if (XF) { v_col(0, r[0], vn); v_col(1, r[1], vn); }
        // block 1: Z4..7 = s1, s1+s2, s1-s2, s2 with s1 = d0+d2, s2 = d1+d3 (sign(7) = -)
        zt[0] = wpk_add(d[0], d[2]); zt[3] = wpk_add(d[1], d[3]); zt[1] = wpk_add(zt[0], zt[3]); zt[2] = wpk_sub(zt[0], zt[3]);
        WW_SB();
        quad(1, 0);
        if (RD) v_read(bufv, prv, 1, r[1]);
        if (XF) { v_col(2, r[2], vn); v_col(3, r[3], vn); }
        quad(1, 1);
        if (RD) v_read(bufv, prv, 2, r[2]);
        if (XF) v_row(0, vn);
        // block 2: Z8..11 = s3, s3+s4, s3-s4, s4 with s3 = d0-d2, s4 = d1-d3 (sign(11) = -)
        zt[0] = wpk_sub(d[0], d[2]); zt[3] = wpk_sub(d[1], d[3]); zt[1] = wpk_add(zt[0], zt[3]); zt[2] = wpk_sub(zt[0], zt[3]);
        WW_SB();
        quad(2, 0);
        if (RD) v_read(bufv, prv, 3, r[3]);
        if (XF) v_row(1, vn);
        quad(2, 1);
        if (XF) v_row(2, vn);
        // block 3: Z12..15 = d2, d2+d3, d2-d3, d3 (sign(12,13,14) = -); the bias partial (sum of the four dY) rides here
        zt[0] = d[2]; zt[1] = wpk_add(d[2], d[3]); zt[2] = wpk_sub(d[2], d[3]); zt[3] = d[3];
        dbacc[c] = wpk_add(dbacc[c], wpk_add(wpk_add(d[0], d[1]), zt[1]));
        WW_SB();
        quad(3, 0);
        if (XF) {
            v_row(3, vn);
            // pin the transform here: without a use in this block the compiler sinks the packed adds towards their use
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("" : "+v"(vn[i]));
        }
        quad(3, 1);
    };
    // ---- pipeline over a 3-deep ring.  Step s = four groups on buffer s%3; the barrier B_s sits in its middle: behind it stage
    // s+1 is visible to everybody (groups 3, 4 already read the next step's first operands from it) and every wave has left
    // step s-1, whose buffer stage s+2 refills — issued right after B_s by the waves 0-3 and at the end of the step by the waves
    // 4-7 (the two waves of a SIMD, w and w+4, issue their LDS-DMA batch half a step apart: one wave's ~200 cycles per
    // instruction then overlap the partner's MFMAs); either way it has until B_(s+1) to land.
    if (ns > 0) {
        stage(0, s0);
        if (ns > 1) stage(1, s0 + 1);
        wait_landed(ns > 1);
        __builtin_amdgcn_s_barrier();
        const bool early = wave < 4;
        f32x2 va[16], vb[16], da[4], db_[4], r[4][4];
        {
#pragma unroll
            for (int j = 0; j < 4; ++j) { v_read(0, 0, j, r[j]); v_col(j, r[j], va); }
#pragma unroll
            for (int i = 0; i < 4; ++i) v_row(i, va);
        }
        z_read(0, 0, 0, da);
        for (int s = 0; s < ns; ++s) {
            const bool more = s + 2 < ns;
            const int buf = s % WW_NST, nbuf = (s + 1) % WW_NST;
            WW_SB();
            group(va, da, 0, true, false, buf, 0, 1, db_, buf, 1, r, vb);       // (pair 0, c 0); reads dY(pair 0, c 1), raw V(pair 1)
            group(va, db_, 1, false, true, buf, 1, 0, da, buf, 1, r, vb);       // (pair 0, c 1); V(pair 1); reads dY(pair 1, c 0)
            // stage s+1 has landed (nothing newer is in flight) and every wave is past step s-1
            if (!(WW_DBG & 1)) ww_wait_vmcnt<0>();
            if (!(WW_DBG & 2)) __builtin_amdgcn_s_barrier();
            WW_SB();
            if (more && early && !(WW_DBG & 1)) stage((s + 2) % WW_NST, s0 + s + 2);
            WW_SB();
            // (past the last step these read stale LDS that nobody uses)
            group(vb, da, 0, true, false, buf, 1, 1, db_, nbuf, 0, r, va);      // (pair 1, c 0); reads dY(pair 1, c 1), next raw V(pair 0)
            group(vb, db_, 1, false, true, nbuf, 0, 0, da, nbuf, 0, r, va);     // (pair 1, c 1); next V(pair 0); reads next dY(pair 0, c 0)
            if (more && !early && !(WW_DBG & 1)) stage((s + 2) % WW_NST, s0 + s + 2);
            WW_SB();
        }
    }
#undef WW_SB

    // ---- slab: [tap 9][cj 64][ci 64] | db[64]: the workgroup applies the signs and G^T . G to its partial itself (the slabs
    // are 9/16 of the Winograd-domain ones: 29 MB less written by every launch's last wave of stores and read again by the
    // reduce).  D lane layout: ci = 4*kg + r, cj = l15: ci is the fast index, so the four accumulator registers of a (tap, c)
    // are one 16-byte store and the reduce, with its threads along ci, writes the caller's OIHW gradient as consecutive
    // 36-byte runs.   r = G^T m (3x4), dg = r G (3x3);  G^T = [[1,.5,.5,0],[0,.5,-.5,0],[0,.5,.5,1]]
    float *slab = p.slab + (size_t)(tile * k.nsplit + part) * k.pstride;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        f32x4 r[3][4];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            // signs of Z (load side): xi 3, 7, 11, 12, 13, 14 are negated
            const f32x4 m0 = b == 3 ? -acc[b][c] : acc[b][c];
            const f32x4 m1 = b == 3 ? -acc[4 + b][c] : acc[4 + b][c];
            const f32x4 m2 = b == 3 ? -acc[8 + b][c] : acc[8 + b][c];
            const f32x4 m3 = b == 3 ? acc[12 + b][c] : -acc[12 + b][c];
            const f32x4 h = 0.5f * (m1 + m2);
            r[0][b] = m0 + h;
            r[1][b] = 0.5f * (m1 - m2);
            r[2][b] = h + m3;
        }
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const f32x4 h = 0.5f * (r[a][1] + r[a][2]);
            const f32x4 o0 = r[a][0] + h, o1 = 0.5f * (r[a][1] - r[a][2]), o2 = h + r[a][3];
            float *dst = slab + ((a * 3) * 64 + (2 * wj + c) * 16 + l15) * 64 + wi * 16 + 4 * kg;
            *(f32x4 *)(dst) = o0;
            *(f32x4 *)(dst + 4096) = o1;
            *(f32x4 *)(dst + 8192) = o2;
        }
    }
    if (wi == 0) {
        // bias partial: sum over this lane's tiles (.x + .y) and over the 4 k groups (lanes l15 + 16*kg)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            float v = dbacc[c][0] + dbacc[c][1];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            if (kg == 0) slab[9 * 64 * 64 + (2 * wj + c) * 16 + l15] = v;
        }
    }
}

// out[(i0+i)*si + (j0+j)*sj + t*st] = (G^T (sigma (.) sum_parts dU) G)[t],  db[j] = sum_parts
// PG part groups of 64 threads (4, or 16 for the launches with one to four channel tiles: 64 blocks per tile are too few
// to keep the memory system busy with 4 waves each)
template <int PG>
__global__ __launch_bounds__(64 * PG) void wgradw_reduce_kernel(const float *__restrict__ slab, int nsplit, size_t pstride, int ntile_j,
                                                            float *__restrict__ out, long si, long sj, long st,
                                                            float *__restrict__ db, int db_c0)
{
    // block = one (tile, cj); 64 threads along ci (the slab's fast index) x PG part groups
    const int tile = blockIdx.x >> 6, jb = blockIdx.x & 63;
    const int j = threadIdx.x & 63, grp = threadIdx.x >> 6;          // j: this thread's ci within the tile
    const float *src = slab + (size_t)tile * nsplit * pstride + (size_t)jb * 64 + j;
    float m[9];
#pragma unroll
    for (int x = 0; x < 9; ++x) m[x] = 0.f;
    for (int P = grp; P < nsplit; P += PG) {
        const float *s = src + (size_t)P * pstride;
#pragma unroll
        for (int x = 0; x < 9; ++x) m[x] += s[x * 4096];
    }
    __shared__ float red[PG][3][64];
    // three passes of three taps through a 3 (12) KiB buffer
#pragma unroll
    for (int xq = 0; xq < 3; ++xq) {
        if (xq) __syncthreads();
#pragma unroll
        for (int x = 0; x < 3; ++x) red[grp][x][j] = m[3 * xq + x];
        __syncthreads();
        if (grp == 0) {
#pragma unroll
            for (int x = 0; x < 3; ++x) {
                float v = red[0][x][j];
#pragma unroll
                for (int g = 1; g < PG; ++g) v += red[g][x][j];
                m[3 * xq + x] = v;
            }
        }
    }
    if (grp == 0) {
        const int it = tile / ntile_j, jt = tile - it * ntile_j;
        float *o = out + (size_t)(it * 64 + j) * si + (size_t)(jt * 64 + jb) * sj;
#pragma unroll
        for (int t = 0; t < 9; ++t) o[t * st] = m[t];
    }
    if (db && jb == 0 && (tile / ntile_j) == 0) {
        // bias gradient of channel tile jt: slabs of the tiles (it = 0, jt)
        __syncthreads();
        const int jt = tile % ntile_j;
        float v = 0.f;
        for (int P = grp; P < nsplit; P += PG) v += slab[((size_t)tile * nsplit + P) * pstride + 9 * 64 * 64 + j];
        red[grp][0][j] = v;
        __syncthreads();
        if (grp == 0) {
            float t = red[0][0][j];
#pragma unroll
            for (int g = 1; g < PG; ++g) t += red[g][0][j];
            db[db_c0 + jt * 64 + j] = t;
        }
    }
}

bool wgradw_applicable(const WgradP &p)
{
    if (p.TY != 3 || p.TX != 3 || p.stride != 1) return false;
    if (p.Ci % 64 != 0 || p.Cj % 64 != 0 || p.db_on_x) return false;
    return true;
}

static int ww_cus()
{
    static int ncu[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (!ncu[dev] && hipDeviceGetAttribute(&ncu[dev], hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) ncu[dev] = 256;
    return ncu[dev];
}

static void ww_decompose(const WgradP &p, WgradWK &k, int cus)
{
    k.p = p;
    k.wy0 = p.ywin0; k.wx0 = p.xwin0;
    k.TYn = cdiv(p.ywin1 - p.ywin0, 2);
    k.TXn = cdiv(p.xwin1 - p.xwin0, 2);
    k.TXg = k.TXn + 1;
    k.NTg = p.NB * k.TYn * k.TXg;
    k.nsteps = cdiv(k.NTg, 16);
    k.ntile_i = p.Ci / 64; k.ntile_j = p.Cj / 64;
    const int ntile = k.ntile_i * k.ntile_j;
    int nsplit = cus / ntile;
    if (nsplit < 1) nsplit = 1;
    if (nsplit > k.nsteps) nsplit = k.nsteps;
    k.nsplit = nsplit;
    k.steps_per = cdiv(k.nsteps, nsplit);
    k.d_spi = make_fastdiv((unsigned)(k.TYn * k.TXg));
    k.d_txg = make_fastdiv((unsigned)k.TXg);
    k.pstride = WW_SLAB;
}

size_t wgradw_slab_need(const WgradP &p)
{
    if (!wgradw_applicable(p)) return 0;
    WgradWK k;
    ww_decompose(p, k, 256);          // sized for the largest device (the split never exceeds the CU count)
    return (size_t)k.ntile_i * k.ntile_j * k.nsplit * k.pstride * sizeof(float);
}

int launch_wgradw(const WgradP &p, hipStream_t st)
{
    const size_t xb = (size_t)p.NB * p.XH * p.XW * p.XC * sizeof(float), yb = (size_t)p.NB * p.YH * p.YW * p.YC * sizeof(float);
    const bool buf = get_lds_dma_mode() != 0 && xb < 0x7FFFFFFFull && yb < 0x7FFFFFFFull;
    auto kern = buf ? wgradw_f32_kernel<true> : wgradw_f32_kernel<false>;
    static bool attr_done[2][64] = {{false}};
    if (int rc_ = ensure_dynamic_lds((const void *)kern, WW_LDS, attr_done[buf ? 1 : 0])) return rc_;
    WgradWK k;
    int cus = ww_cus();
    if (cus > 256) cus = 256;
    ww_decompose(p, k, cus);
    k.xbytes = buf ? (int)xb : 0; k.ybytes = buf ? (int)yb : 0;
    const int ntile = k.ntile_i * k.ntile_j;
    const size_t need = (size_t)ntile * k.nsplit * k.pstride * sizeof(float);
    ARG_CHECK(need <= p.slab_bytes, "wgradw: slab scratch too small (%zu < %zu)", p.slab_bytes, need);
    ARG_CHECK((size_t)p.NB * p.XH * p.XW * p.XC < 0x7FFFFFFFull && (size_t)p.NB * p.YH * p.YW * p.YC < 0x7FFFFFFFull,
              "wgradw: tensor exceeds 31-bit element offsets");
    if (p.db) ARG_CHECK(p.ywin0 == 0 && p.xwin0 == 0 && p.ywin1 == p.YH && p.xwin1 == p.YW, "wgradw: fused bias gradient needs the full Y window");
    char tag[96];
    snprintf(tag, sizeof(tag), "wgradw<%d> Ci=%d Cj=%d Y=%dx%d tiles=%dx%d steps=%d split=%d", (int)buf, p.Ci, p.Cj, p.YH, p.YW, k.TYn, k.TXn, k.nsteps, k.nsplit);
    prof_begin(PK_WGRAD, tag, st, wgrad_alg_flops(p), 2.0 * k.nsteps * 16.0 * 16.0 * p.Ci * p.Cj, wgrad_alg_bytes(p));
    hipLaunchKernelGGL(kern, dim3(ntile * k.nsplit), dim3(512), WW_LDS, st, k);
    prof_end(st);
    HIP_TRY(hipGetLastError());
    prof_begin(PK_REDUCE, "wgradw_reduce", st, 0.0, 0.0, (double)ntile * k.nsplit * k.pstride * 4.0 + 9.0 * p.Ci * p.Cj * 4.0);
    if (ntile <= 4 && k.nsplit >= 32)
        hipLaunchKernelGGL(wgradw_reduce_kernel<16>, dim3(ntile * 64), dim3(1024), 0, st, p.slab, k.nsplit, k.pstride, k.ntile_j,
                           p.out, p.si, p.sj, p.st, p.db, p.yc0);
    else
        hipLaunchKernelGGL(wgradw_reduce_kernel<4>, dim3(ntile * 64), dim3(256), 0, st, p.slab, k.nsplit, k.pstride, k.ntile_j,
                           p.out, p.si, p.sj, p.st, p.db, p.yc0);
    prof_end(st);
    HIP_TRY(hipGetLastError());
    return 0;
}

}  // namespace unet
