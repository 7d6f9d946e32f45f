// igemm_epilogue.hpp — epilogue shared by the implicit-GEMM kernels (igemm.hip, igemmx.hip).
#pragma once
#include "common.hpp"

namespace unet {

__device__ __forceinline__ int fdiv(int n, const FastDiv &f) { return (int)(((unsigned long long)(unsigned)n * f.mul) >> f.shift); }

// The epilogue's (and the K loop's) kernel arguments are copied into locals up front and pinned in SGPRs (IGB_PIN): fetched where
// they are used, every one of them is a scalar-load round trip behind the branch that needs it — dozens in a row per workgroup,
// some of them once per K step (found with in-kernel stamps on the fp32 Winograd kernel).
#define IGB_PIN(x) asm volatile("" : "+s"(x))
struct IgEp {
    int rw0, rw1, scatter, DC, OH, OW, DH, DW, dwy0, dwx0, M, cout, Nn, dn0, relu;
    FastDiv d_ohw, d_ow;
    const float *bias, *mask, *add;
    float *dst;
};
__device__ __forceinline__ IgEp igb_epilogue_args(const IgemmP &p)
{
    IgEp e;
    e.rw0 = p.rw0; e.rw1 = p.rw1; e.scatter = p.scatter; e.DC = p.DC; e.OH = p.OH; e.OW = p.OW; e.DH = p.DH; e.DW = p.DW;
    e.dwy0 = p.dwy0; e.dwx0 = p.dwx0; e.M = p.M; e.cout = p.cout; e.Nn = p.Nn; e.dn0 = p.dn0; e.relu = p.relu;
    e.d_ohw = p.d_ohw; e.d_ow = p.d_ow;
    e.bias = p.bias; e.mask = p.mask; e.add = p.add; e.dst = p.dst;
    IGB_PIN(e.rw0); IGB_PIN(e.rw1); IGB_PIN(e.scatter); IGB_PIN(e.DC); IGB_PIN(e.OH); IGB_PIN(e.OW); IGB_PIN(e.DH); IGB_PIN(e.DW);
    IGB_PIN(e.dwy0); IGB_PIN(e.dwx0); IGB_PIN(e.M); IGB_PIN(e.cout); IGB_PIN(e.Nn); IGB_PIN(e.dn0); IGB_PIN(e.relu);
    IGB_PIN(e.d_ohw.mul); IGB_PIN(e.d_ohw.shift); IGB_PIN(e.d_ow.mul); IGB_PIN(e.d_ow.shift);
    // (the pointers are not pinned: behind the asm they would be generic pointers, i.e. FLAT instructions)
    return e;
}


// ---- epilogue: bias / add / ReLU / mask / store with 16-byte accesses.
// The MFMA accumulator layout gives a lane one column and 16 rows, i.e. dword stores (64 per lane; measured
// ~4 us of store issue per workgroup, 18 % of a short-K layer).  Each wave therefore transposes its 32x32
// sub-tiles through a private 4.5 KiB LDS patch (row pitch 36 floats: 16-B aligned, conflict-free) and
// writes/reads global memory as float4: 16 stores per lane, each wave instruction covering 8 rows x 128 B.
// Rows past M use row M-1's (valid) offset so mask/add loads are unconditional; only the store is predicated.
constexpr int EPI_PITCH = 36;
constexpr int EPI_WAVE_BYTES = 32 * EPI_PITCH * 4;

// destination offset (+ deferred-ReLU flag) of tile row `i` into the LDS tables
template <int BM, class P>
__device__ __forceinline__ void igemm_rowoff_entry(const P &p, int m0, int i, unsigned char *lds)
{
    unsigned *rowoff = (unsigned *)lds;
    unsigned char *inwin = lds + BM * 4 + 4 * EPI_WAVE_BYTES;     // per-row flag: pixel inside the deferred-ReLU window
    const bool relu_win = p.rw1 > p.rw0;
    int m = m0 + i;
    m = m < p.M ? m : p.M - 1;
    unsigned off;
    unsigned char flag = 0;
    if (!p.scatter && !relu_win) {
        off = (unsigned)m * (unsigned)p.DC;
    } else {
        const int ohw = p.OH * p.OW;
        const int img = fdiv(m, p.d_ohw);
        const int rem = m - img * ohw;
        const int oy = fdiv(rem, p.d_ow);
        const int ox = rem - oy * p.OW;
        if (p.scatter == 1) off = (unsigned)((img * p.DH + 2 * oy) * p.DW + 2 * ox) * (unsigned)p.DC;
        else if (p.scatter == 2) off = (unsigned)((img * p.DH + oy + p.dwy0) * p.DW + ox + p.dwx0) * (unsigned)p.DC;
        else off = (unsigned)m * (unsigned)p.DC;
        flag = relu_win && oy >= p.rw0 && oy < p.rw1 && ox >= p.rw0 && ox < p.rw1;
    }
    rowoff[i] = off;
    inwin[i] = flag;
}

// stores of one consumer thread (tid in [0,256)); the row tables must be complete (barrier) before the call
template <int BM, int BN, class P>
__device__ __forceinline__ void igemm_epilogue_store(const P &p, f32x16 (&acc)[2][2], int m0, int n0, int tid,
                                                     unsigned char *lds /* >= BM*5 + 4*EPI_WAVE_BYTES bytes */)
{
    constexpr int WN = BN / 64;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, lh = lane >> 5;
    unsigned *rowoff = (unsigned *)lds;
    float *patch = (float *)(lds + BM * 4 + wave * EPI_WAVE_BYTES);
    unsigned char *inwin = lds + BM * 4 + 4 * EPI_WAVE_BYTES;
    const bool relu_win = p.rw1 > p.rw0;
    const int rrow = lane >> 3, cg = lane & 7;           // read-back role: row rrow + 8k, columns 4cg..4cg+3
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) {
        const int nb = n0 + wn * 64 + tn * 32;
        float bv = 0.f;
        if (p.bias) {
            int n = nb + l31;
            n = n < p.Nn ? n : p.Nn - 1;
            bv = p.bias[p.cout ? n % p.cout : n];
        }
        const int n4 = nb + 4 * cg;
        const bool n_ok = n4 < p.Nn;
        const int nc = n_ok ? n4 : 0;
        int coloff;
        if (p.scatter != 1) {
            coloff = p.dn0 + nc;
        } else {
            const int ab = nc / p.cout;
            coloff = ((ab >> 1) * p.DW + (ab & 1)) * p.DC + p.dn0 + (nc - ab * p.cout);
        }
#pragma unroll
        for (int tm = 0; tm < 2; ++tm) {
            // +add / ReLU' mask operands first: their latency runs under the LDS transpose
            size_t o[4];
            f32x4 ta[4], tk[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = (size_t)rowoff[wm * 64 + tm * 32 + rrow + 8 * k] + (size_t)coloff;
            if (p.add) {
#pragma unroll
                for (int k = 0; k < 4; ++k) ta[k] = *(const f32x4 *)(p.add + o[k]);
            }
            if (p.mask) {
#pragma unroll
                for (int k = 0; k < 4; ++k) tk[k] = *(const f32x4 *)(p.mask + o[k]);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r)
                patch[((r & 3) + 8 * (r >> 2) + 4 * lh) * EPI_PITCH + l31] = acc[tm][tn][r] + bv;
            f32x4 v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = *(const f32x4 *)(patch + (rrow + 8 * k) * EPI_PITCH + 4 * cg);
            if (p.add) {
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] += ta[k];
            }
            if (p.relu) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const bool defer = relu_win && inwin[wm * 64 + tm * 32 + rrow + 8 * k];
#pragma unroll
                    for (int c = 0; c < 4; ++c) v[k][c] = (v[k][c] > 0.f || defer) ? v[k][c] : 0.f;
                }
            }
            if (p.mask) {
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int c = 0; c < 4; ++c) v[k][c] = tk[k][c] > 0.f ? v[k][c] : 0.f;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (n_ok && m0 + wm * 64 + tm * 32 + rrow + 8 * k < p.M) *(f32x4 *)(p.dst + o[k]) = v[k];
        }
    }
}

template <int BM, int BN, class P>
__device__ __forceinline__ void igemm_epilogue(const P &p, f32x16 (&acc)[2][2], int m0, int n0, int tid, unsigned char *lds)
{
    if (tid < BM) igemm_rowoff_entry<BM>(p, m0, tid, lds);
    __syncthreads();
    igemm_epilogue_store<BM, BN>(p, acc, m0, n0, tid, lds);
}

}  // namespace unet
