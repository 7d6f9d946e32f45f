// igemm2.hip — second-generation fp32-MFMA implicit GEMM: K step 16, three-stage LDS ring, ONE barrier
// per K step placed in the MIDDLE of the step's MFMAs, fragments of the next stage prefetched into
// registers behind that barrier.
//
// Why: in the first kernel (igemm.hip: K step 32, two buffers, barrier at the END of a step) every step
// opens with barrier -> ds_read -> first MFMA, a bubble the matrix pipe cannot fill (measured: MFMA
// busy 75%, best-case layer 85%).  Here the pipe never waits on LDS:
//
//   step s:   MFMA(stage s, k 0..7)                       <- operands already in registers
//             s_waitcnt vmcnt(G)      ; stage s+1 landed (this wave's share), stage s+2 stays in flight
//             s_barrier               ; ... for every wave; every wave is done reading buffer s%3
//             LDS-DMA stage s+3 -> buffer s%3             (two full steps to land)
//             ds_read fragments of stage s+1              (latency hidden by ...)
//             MFMA(stage s, k 8..15)
//
// LDS per workgroup: 3 x (BM+BN) x 64 B = 48 KiB for 128x128 -> 3 workgroups per CU (3 waves per SIMD).
// LDS rows are 64 B (16 floats of K); the 16-B chunk index is XOR-swizzled with (row>>2)&3 on the
// SOURCE address of the LDS-DMA and on the fragment read -> conflict-free ds_read_b128.
#include "common.hpp"
#include "igemm_epilogue.hpp"
#include <cstdio>

namespace unet {

#define GLDS16(gptr, lptr)                                                                    \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gptr),  \
                                     (__attribute__((address_space(3))) void *)(lptr), 16, 0, 0)

template <int N> __device__ __forceinline__ void wait_vmcnt()
{
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// wait until at most `stages` whole stages (G LDS-DMA instructions each) are still in flight
template <int G> __device__ __forceinline__ void wait_stages(int stages)
{
    switch (stages) {
    case 0: wait_vmcnt<0>(); break;
    case 1: wait_vmcnt<G>(); break;
    case 2: wait_vmcnt<2 * G>(); break;
    case 3: wait_vmcnt<3 * G>(); break;
    case 4: wait_vmcnt<4 * G>(); break;
    default: wait_vmcnt<5 * G>(); break;
    }
}

struct Frags { f32x4 a[2][2]; f32x4 b[2][2]; };      // [tile][k8 half]

template <int BM, int BN, bool PAD, int NST>
__global__ __launch_bounds__(256, 2) void igemm2_f32_kernel(const IgemmP p)
{
    constexpr int WN = BN / 64, WM = 4 / WN;
    static_assert(WM * 64 == BM, "4 waves of 64x64");
    constexpr int RA = BM / 64, RB = BN / 64;       // staging passes (64 rows each)
    constexpr int G = RA + RB;                      // LDS-DMA instructions per thread per stage
    constexpr int A_BYTES = BM * 64, B_BYTES = BN * 64, STAGE = A_BYTES + B_BYTES;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;

    int logical;
    {
        const int nblk = gridDim.x, q = nblk >> 3, r = nblk & 7, xcd = blockIdx.x & 7;
        logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (blockIdx.x >> 3);
    }
    const int mt = logical / p.ntiles, nt = logical - mt * p.ntiles;
    const int m0 = mt * BM, n0 = nt * BN;

    const int srow = tid >> 2;                                   // 0..63
    const int schunk = (tid & 3) ^ ((srow >> 2) & 3);
    const int coff = schunk * 4;

    int a_off[RA], a_iy[RA], a_ix[RA], b_off[RB];
#pragma unroll
    for (int j = 0; j < RB; ++j) {
        int n = n0 + srow + 64 * j;
        n = n < p.Nn ? n : p.Nn - 1;
        b_off[j] = n * p.ldw + coff;
    }
    int s_src = 0, ty = 0, tx = 0, kc = 0, kglob = 0;
    const float *sp = nullptr;
    int sH = 0, sW = 0, sC = 0, snch = 0, toff = 0;

    auto setup_source = [&](int si) {
        const GSrc &g = p.src[si];
        sp = g.p; sH = g.H; sW = g.W; sC = g.C; snch = g.nch;
        const int ohw = p.OH * p.OW;
#pragma unroll
        for (int i = 0; i < RA; ++i) {
            int m = m0 + srow + 64 * i;
            m = m < p.M ? m : p.M - 1;
            const int img = fdiv(m, p.d_ohw);
            const int rem = m - img * ohw;
            const int oy = fdiv(rem, p.d_ow);
            const int ox = rem - oy * p.OW;
            const int iy = (oy + p.oy0) * p.stride - g.pad;
            const int ix = (ox + p.ox0) * p.stride - g.pad;
            a_iy[i] = iy; a_ix[i] = ix;
            a_off[i] = ((img * g.H + iy) * g.W + ix) * g.C + g.c0 + coff;
        }
        toff = 0;
    };
    auto stage = [&](int buf) {
        unsigned char *abase = smem + buf * STAGE + wave * (16 * 64);
#pragma unroll
        for (int i = 0; i < RA; ++i) {
            const float *g = sp + (a_off[i] + toff + kc);
            if (PAD) {
                const bool inb = (unsigned)(a_iy[i] + ty) < (unsigned)sH && (unsigned)(a_ix[i] + tx) < (unsigned)sW;
                g = inb ? g : p.zeros + coff;
            }
            GLDS16(g, abase + i * (64 * 64));
        }
        unsigned char *bbase = abase + A_BYTES;
#pragma unroll
        for (int j = 0; j < RB; ++j) GLDS16(p.wt + (b_off[j] + kglob), bbase + j * (64 * 64));
    };
    auto advance = [&]() {
        kglob += 16;
        kc += 16;
        if (kc == snch) {
            kc = 0;
            ++tx;
            if (tx == p.TX) { tx = 0; ++ty; }
            if (ty * p.TX + tx == p.T) {
                ty = 0; tx = 0;
                ++s_src;
                if (s_src < p.nsrc) setup_source(s_src);
            } else {
                toff = (ty * sW + tx) * sC;
            }
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int l31 = lane & 31, lh = lane >> 5;
    const int swz = (l31 >> 2) & 3;
    const int a_rd = (wm * 64 + l31) * 64;
    const int b_rd = A_BYTES + (wn * 64 + l31) * 64;
    const int pos0 = ((0 + lh) ^ swz) * 16, pos1 = ((2 + lh) ^ swz) * 16;

    auto read_frags = [&](Frags &f, int buf) {
        const unsigned char *sb = smem + buf * STAGE;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            f.a[t][0] = *(const f32x4 *)(sb + a_rd + t * (32 * 64) + pos0);
            f.a[t][1] = *(const f32x4 *)(sb + a_rd + t * (32 * 64) + pos1);
            f.b[t][0] = *(const f32x4 *)(sb + b_rd + t * (32 * 64) + pos0);
            f.b[t][1] = *(const f32x4 *)(sb + b_rd + t * (32 * 64) + pos1);
        }
    };
    auto mfma_half = [&](const Frags &f, int g) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[0][g][t], f.b[0][g][t], acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[0][g][t], f.b[1][g][t], acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[1][g][t], f.b[0][g][t], acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[1][g][t], f.b[1][g][t], acc[1][1], 0, 0, 0);
        }
    };

    const int nk = p.Kd >> 4;
    setup_source(0);
    int issued = 0;
    for (; issued < NST && issued < nk; ++issued) { stage(issued); advance(); }
    wait_stages<G>(issued - 1);
    __builtin_amdgcn_s_barrier();
    Frags f0, f1;
    read_frags(f0, 0);

    // one pipeline step; `cur` holds stage s, `nxt` receives stage s+1
    auto step = [&](int s, const Frags &cur, Frags &nxt) {
        mfma_half(cur, 0);
        if (s + 1 < nk) {
            // in flight: stages s+1 .. min(s+NST-1, nk-1); stage s+1 must have landed
            const int inflight = (nk - 1 - s) < (NST - 1) ? (nk - 1 - s) : (NST - 1);
            wait_stages<G>(inflight - 1);
            __builtin_amdgcn_s_barrier();
            if (s + NST < nk) { stage(s % NST); advance(); }
            read_frags(nxt, (s + 1) % NST);
        }
        mfma_half(cur, 1);
    };
    int s = 0;
    for (; s + 1 < nk; s += 2) {
        step(s, f0, f1);
        step(s + 1, f1, f0);
    }
    if (s < nk) step(s, f0, f1);

    __syncthreads();            // every wave is done with the LDS ring before it is reused below

    // ---- epilogue (same as igemm.hip): per-row destination offsets through LDS, then
    // bias / add / ReLU / mask / store; only the store is predicated.
    unsigned *rowoff = (unsigned *)smem;
    if (tid < BM) {
        int m = m0 + tid;
        m = m < p.M ? m : p.M - 1;
        unsigned off;
        if (!p.scatter) {
            off = (unsigned)m * (unsigned)p.DC;
        } else {
            const int ohw = p.OH * p.OW;
            const int img = fdiv(m, p.d_ohw);
            const int rem = m - img * ohw;
            const int oy = fdiv(rem, p.d_ow);
            const int ox = rem - oy * p.OW;
            off = (unsigned)((img * p.DH + 2 * oy) * p.DW + 2 * ox) * (unsigned)p.DC;
        }
        rowoff[tid] = off;
    }
    __syncthreads();

#pragma unroll
    for (int tn = 0; tn < 2; ++tn) {
        const int n_raw = n0 + wn * 64 + tn * 32 + l31;
        const bool n_ok = n_raw < p.Nn;
        const int n = n_ok ? n_raw : p.Nn - 1;
        int coloff, bidx;
        if (!p.scatter) {
            coloff = p.dn0 + n;
            bidx = p.cout ? n % p.cout : n;
        } else {
            const int ab = n / p.cout;
            bidx = n - ab * p.cout;
            coloff = ((ab >> 1) * p.DW + (ab & 1)) * p.DC + p.dn0 + bidx;
        }
        const float bv = p.bias ? p.bias[bidx] : 0.f;
#pragma unroll
        for (int tm = 0; tm < 2; ++tm) {
            size_t o[16];
            float v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm * 64 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                o[r] = (size_t)rowoff[row] + (size_t)coloff;
                v[r] = acc[tm][tn][r] + bv;
            }
            if (p.add) {
                float t[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) t[r] = p.add[o[r]];
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] += t[r];
            }
            if (p.relu) {
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = v[r] > 0.f ? v[r] : 0.f;
            }
            if (p.mask) {
                float t[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) t[r] = p.mask[o[r]];
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = t[r] > 0.f ? v[r] : 0.f;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm * 64 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (n_ok && m0 + row < p.M) p.dst[o[r]] = v[r];
            }
        }
    }
}

double igemm_alg_flops(const IgemmP &p);

template <int BM, int BN, bool PAD, int NST>
static int launch_cfg2(const IgemmP &p, hipStream_t st)
{
    constexpr int LDS = NST * (BM + BN) * 64;
    static bool attr_done[64] = {false};
    auto kern = igemm2_f32_kernel<BM, BN, PAD, NST>;
    if (int rc_ = ensure_dynamic_lds((const void *)kern, LDS, attr_done)) return rc_;
    IgemmP q = p;
    q.mtiles = cdiv(p.M, BM);
    q.ntiles = cdiv(p.Nn, BN);
    char tag[96];
    snprintf(tag, sizeof(tag), "igemm2<%d;%d;%d> M=%d N=%d Kd=%d T=%d s=%d nsrc=%d", BM, BN, (int)PAD, p.M, p.Nn, p.Kd, p.T, p.stride, p.nsrc);
    prof_begin(0, igemm_alg_flops(p), st, tag);
    hipLaunchKernelGGL(kern, dim3(q.mtiles * q.ntiles), dim3(256), LDS, st, q);
    prof_end(st);
    HIP_TRY(hipGetLastError());
    return 0;
}

int launch_igemm2(const IgemmP &p, bool pad, hipStream_t st)
{
    // K step 16: 5 stages of 16 KiB (128x128) / 4 stages of 20 KiB (256x64) = 80 KiB -> 2 workgroups per CU,
    // LDS-DMA issued 4 (3) steps ahead of its consumer
    if (p.Nn % 128 == 0) return pad ? launch_cfg2<128, 128, true, 5>(p, st) : launch_cfg2<128, 128, false, 5>(p, st);
    return pad ? launch_cfg2<256, 64, true, 4>(p, st) : launch_cfg2<256, 64, false, 4>(p, st);
}

}  // namespace unet
