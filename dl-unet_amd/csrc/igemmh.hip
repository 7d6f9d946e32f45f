// igemmh.hip — implicit GEMM for stride-1 3x3 convolutions (forward and dgrad) with a 2-D output tile and an
// LDS-resident input HALO tile: the A operand of all 9 taps is read from one (TH+2) x (TW+2) pixel patch that is
// fetched ONCE per 32-channel chunk, instead of 9 im2col row sets (igemm.hip).  Per chunk a 128x128 tile stages
// 23 KiB of input + 9 x 16 KiB of filter taps (167 LDS-DMA instructions) instead of 9 x 32 KiB (288); a 256x64
// tile 41 + 72 KiB instead of 360 KiB.  Everything else — fp32 MFMA 32x32x2, 4 waves of 64x64, swizzled 128-B LDS
// rows, one barrier per (chunk, tap) step, the shared epilogue — is as in igemm.hip.
//
// Output tile: TH x TW pixels of ONE image (TW = 16; TH = 8 for BM=128, 16 for BM=256); tile row i <-> pixel
// (i / TW, i % TW).  Halo pixel hp = hy*(TW+2)+hx holds input pixel (tile_y0 + oy0 - pad + hy, tile_x0 + ox0 - pad + hx);
// pixels outside the source tensor come from the zero page (virtual padding).  K order: source, 32-channel chunk,
// tap — only the summation order differs from igemm.hip (same packed weights, column t*C + c).
#include "common.hpp"
#include "igemm_epilogue.hpp"
#include <cstdio>
#include <cstdlib>

namespace unet {

#define GLDS16(gptr, lptr)                                                                    \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gptr),  \
                                     (__attribute__((address_space(3))) void *)(lptr), 16, 0, 0)

struct IgemmHP { IgemmP p; int tiles_x, tiles_y; };     // tiles per image along x / y

template <int BM, int BN>
struct HaloGeom {
    static constexpr int TW = 16, TH = BM / TW;
    static constexpr int HW = TW + 2, HH = TH + 2, HPIX = HW * HH;
    static constexpr int HGROUPS = (HPIX + 7) / 8;             // LDS-DMA instructions per halo (8 pixels each)
    static constexpr int HALO_BYTES = HGROUPS * 1024;
    static constexpr int B_BYTES = BN * 128;
    static constexpr int HPT = (HGROUPS + 3) / 4;              // halo instructions per wave
    static constexpr int LDS = 2 * HALO_BYTES + 2 * B_BYTES;
};

template <int BM, int BN>
__global__ __launch_bounds__(256, 2) void igemmh_f32_kernel(const IgemmHP k)
{
    using G = HaloGeom<BM, BN>;
    constexpr int TW = G::TW, HW = G::HW;
    constexpr int WN = BN / 64, WM = 4 / WN;
    static_assert(WM * 64 == BM, "4 waves of 64x64");
    constexpr int RB = BN / 32;
    const IgemmP &p = k.p;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *halo = smem;                                // 2 x HALO_BYTES
    unsigned char *bbuf = smem + 2 * G::HALO_BYTES;            // 2 x B_BYTES

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;

    int logical;
    {
        const int nblk = gridDim.x, q = nblk >> 3, r = nblk & 7, xcd = blockIdx.x & 7;
        logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (blockIdx.x >> 3);
    }
    const int mt = logical / p.ntiles, nt = logical - mt * p.ntiles;
    const int n0 = nt * BN;
    const int tpi = k.tiles_x * k.tiles_y;
    const int img = mt / tpi;
    const int trem = mt - img * tpi;
    const int tyi = trem / k.tiles_x, txi = trem - tyi * k.tiles_x;
    const int ty0 = tyi * G::TH, tx0 = txi * TW;               // tile origin in the output domain

    // ---- halo staging role: instruction j of this wave covers halo pixels 8*(4j + wave) .. +7
    const int hrow = lane >> 3, hpos = lane & 7;
    int h_off[G::HPT];            // element offset of (pixel, channel 0 of the current source) + swizzled chunk; -1 = zero page
    int h_sw[G::HPT];
    // ---- B staging role (as igemm.hip)
    const int srow = tid >> 3;
    const int schunk = (tid & 7) ^ ((srow >> 1) & 7);
    const int coff = schunk * 4;
    int b_off[RB];
#pragma unroll
    for (int j = 0; j < RB; ++j) {
        int n = n0 + srow + 32 * j;
        n = n < p.Nn ? n : p.Nn - 1;
        b_off[j] = n * p.ldw + coff;
    }

    int s = 0, c0 = 0, tap = 0, kcol0 = 0;      // source, channel chunk start, tap; kcol0 = column of (source, tap 0, channel 0)
    const float *sp = nullptr;
    int snch = 0;
    auto setup_source = [&](int si) {
        const GSrc &g = p.src[si];
        sp = g.p; snch = g.nch;
#pragma unroll
        for (int j = 0; j < G::HPT; ++j) {
            const int hp = 8 * (4 * j + wave) + hrow;
            const int hy = hp / HW, hx = hp - hy * HW;
            const int iy = (ty0 + p.oy0) - g.pad + hy, ix = (tx0 + p.ox0) - g.pad + hx;
            const bool ok = hp < G::HPIX && (unsigned)iy < (unsigned)g.H && (unsigned)ix < (unsigned)g.W;
            h_sw[j] = 4 * (hpos ^ ((hp >> 1) & 7));
            h_off[j] = ok ? ((img * g.H + iy) * g.W + ix) * g.C + g.c0 + h_sw[j] : -1;
        }
    };
    auto stage_halo = [&](int buf, int cch) {
#pragma unroll
        for (int j = 0; j < G::HPT; ++j) {
            const int grp = 4 * j + wave;
            if (grp < G::HGROUPS) {
                const float *g = h_off[j] >= 0 ? sp + (h_off[j] + cch) : p.zeros + h_sw[j];
                GLDS16(g, halo + buf * G::HALO_BYTES + grp * 1024);
            }
        }
    };
    auto stage_b = [&](int buf, int kcol) {
        unsigned char *bb = bbuf + buf * G::B_BYTES + wave * (8 * 128);
#pragma unroll
        for (int j = 0; j < RB; ++j) GLDS16(p.wt + (b_off[j] + kcol), bb + j * (32 * 128));
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int l31 = lane & 31, lh = lane >> 5;
    // halo pixel of this lane's two A rows (tile rows wm*64 + tm*32 + l31) at tap (0,0)
    int hp_lane[2];
#pragma unroll
    for (int tm = 0; tm < 2; ++tm) {
        const int i = wm * 64 + tm * 32 + l31;
        hp_lane[tm] = (i / TW) * HW + (i % TW);
    }
    const int bswz = (l31 >> 1) & 7;
    const int b_rd = (wn * 64 + l31) * 128;

    // ---- pipeline over steps (source, chunk, tap): B double-buffered per step, halo double-buffered per chunk
    const int nk = p.Kd >> 5;
    setup_source(0);
    stage_halo(0, 0);
    stage_b(0, 0);
    __syncthreads();
    int hbuf = 0;
    for (int ks = 0; ks < nk; ++ks) {
        const int cur = ks & 1;
        // state of step ks: (s, c0, tap); prefetch for step ks+1
        int ns = s, nc0 = c0, ntap = tap + 1, nkcol0 = kcol0;
        bool new_chunk = false;
        if (ntap == 9) {
            ntap = 0; nc0 = c0 + 32; new_chunk = true;
            if (nc0 == snch) { nc0 = 0; nkcol0 = kcol0 + 9 * snch; ns = s + 1; }
        }
        if (ks + 1 < nk) {
            if (new_chunk) {
                if (ns != s) setup_source(ns);
                stage_halo(hbuf ^ 1, nc0);
            }
            const int nn = (ns != s) ? p.src[ns].nch : snch;
            stage_b(cur ^ 1, nkcol0 + ntap * nn + nc0);
        }
        // compute step ks
        const int tyy = tap / 3, txx = tap - tyy * 3;
        const int tapoff = tyy * HW + txx;
        const unsigned char *hb = halo + hbuf * G::HALO_BYTES;
        const unsigned char *sb = bbuf + cur * G::B_BYTES;
        int a_rd[2], a_sw[2];
#pragma unroll
        for (int tm = 0; tm < 2; ++tm) {
            const int hp = hp_lane[tm] + tapoff;
            a_rd[tm] = hp * 128;
            a_sw[tm] = (hp >> 1) & 7;
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int q = 2 * g + lh;
            const f32x4 a0 = *(const f32x4 *)(hb + a_rd[0] + ((q ^ a_sw[0]) * 16));
            const f32x4 a1 = *(const f32x4 *)(hb + a_rd[1] + ((q ^ a_sw[1]) * 16));
            const int pos = (q ^ bswz) * 16;
            const f32x4 b0 = *(const f32x4 *)(sb + b_rd + pos);
            const f32x4 b1 = *(const f32x4 *)(sb + b_rd + 32 * 128 + pos);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[t], b0[t], acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[t], b1[t], acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[t], b0[t], acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[t], b1[t], acc[1][1], 0, 0, 0);
            }
        }
        __syncthreads();
        if (new_chunk) hbuf ^= 1;
        s = ns; c0 = nc0; tap = ntap; kcol0 = nkcol0;
        if (ns < p.nsrc) snch = p.src[ns].nch;
    }

    // ---- epilogue: row tables for the 2-D tile, then the shared store path
    unsigned *rowoff = (unsigned *)smem;
    unsigned char *inwin = smem + BM * 4 + 4 * EPI_WAVE_BYTES;
    if (tid < BM) {
        const int py = tid / TW, px = tid - py * TW;
        int oy = ty0 + py, ox = tx0 + px;
        const bool ok = oy < p.OH && ox < p.OW;
        oy = oy < p.OH ? oy : p.OH - 1; ox = ox < p.OW ? ox : p.OW - 1;
        unsigned off;
        if (p.scatter == 2) off = (unsigned)((img * p.DH + oy + p.dwy0) * p.DW + ox + p.dwx0) * (unsigned)p.DC;
        else off = (unsigned)((img * p.OH + oy) * p.OW + ox) * (unsigned)p.DC;
        rowoff[tid] = ok ? off : (off | 0x80000000u);          // bit 31: row outside the output domain (no store)
        inwin[tid] = (p.rw1 > p.rw0) && oy >= p.rw0 && oy < p.rw1 && ox >= p.rw0 && ox < p.rw1;
    }
    __syncthreads();
    igemm_epilogue_store_flagged<BM, BN>(p, acc, n0, tid, smem);
}

double igemm_alg_flops(const IgemmP &p);

template <int BM, int BN>
static int launch_cfgh(const IgemmP &p, hipStream_t st)
{
    using G = HaloGeom<BM, BN>;
    static bool attr_done[64] = {false};
    auto kern = igemmh_f32_kernel<BM, BN>;
    if (int rc_ = ensure_dynamic_lds((const void *)kern, G::LDS, attr_done)) return rc_;
    IgemmHP q;
    q.p = p;
    q.tiles_x = cdiv(p.OW, G::TW);
    q.tiles_y = cdiv(p.OH, G::TH);
    q.p.mtiles = p.NB * q.tiles_x * q.tiles_y;
    q.p.ntiles = cdiv(p.Nn, BN);
    char tag[96];
    snprintf(tag, sizeof(tag), "igemmh<%d;%d> M=%d N=%d Kd=%d nsrc=%d tiles=%dx%d", BM, BN, p.M, p.Nn, p.Kd, p.nsrc, q.tiles_y, q.tiles_x);
    prof_begin(0, igemm_alg_flops(p), st, tag);
    hipLaunchKernelGGL(kern, dim3(q.p.mtiles * q.p.ntiles), dim3(256), G::LDS, st, q);
    prof_end(st);
    HIP_TRY(hipGetLastError());
    return 0;
}

// usable for: 9 taps (3x3), stride 1, linear or window stores; worthwhile when the 2-D tiling wastes little
bool igemmh_applicable(const IgemmP &p)
{
    if (p.T != 9 || p.TX != 3 || p.stride != 1 || p.scatter == 1) return false;
    static const int n64 = [] { const char *e = getenv("UNET_HALO64"); return e ? atoi(e) : 0; }();
    if (p.Nn % 128 != 0 && !n64) return false;     // the 256x64 tile needs 98 KiB of LDS (1 workgroup/CU)
    const int BM = p.Nn % 128 == 0 ? 128 : 256;
    const int TH = BM / 16;
    const double waste = (double)(cdiv(p.OH, TH) * TH) * (cdiv(p.OW, 16) * 16) / ((double)p.OH * p.OW);
    return waste < 1.12;
}

int launch_igemmh(const IgemmP &p, hipStream_t st)
{
    if (p.Nn % 128 == 0) return launch_cfgh<128, 128>(p, st);
    return launch_cfgh<256, 64>(p, st);
}

}  // namespace unet
