// igemmb.hip — implicit GEMM on the bf16 matrix cores with bf16 tensors in HBM (arithmetic mode 2, BASELINE config #3):
// activations, their gradients and the packed filters are bf16 (2 B/element), accumulation is fp32
// (v_mfma_f32_16x16x32_bf16 here, v_mfma_f32_32x32x16_bf16 in convb64), bias is fp32, the epilogue rounds once to bf16 (RNE).
//
// Same contraction family as igemm.hip (conv3x3 fwd / dgrad over the virtual concat, up-conv fwd / dgrad; network.py:131-188
// and their autograd) and the same descriptor (IgemmP; tensor pointers are bf16 here), but built for the 16x faster pipe:
//   * operands go HBM -> LDS by LDS-DMA through buffer descriptors, bf16 as stored: no VGPR round trip, no conversion;
//   * an LDS row is 128 B = 64 channels of one pixel (or 64 K entries of one filter row), 16-byte chunks XOR-swizzled with
//     (row >> 1) & 7 on the DMA source address and on the fragment read: a ds_read_b128 IS one 16x16x32 operand fragment
//     (lane (r, q) holds k = 8q .. 8q+7 of row r of the k32 step) and the reads are bank-conflict free;
//   * K step = 64 channels of one tap (2 MFMA k-steps; taps innermost), double buffered, one barrier per step;
//   * epilogue: each wave transposes its 32 x 64 accumulator slabs through LDS and stores 16 bytes per lane (8 channels),
//     128 contiguous bytes per pixel row, with bias / +add / ReLU (deferred-ReLU window) / ReLU' mask fused.
// Measured (B = 8 layers of the net, rocprofv3 PMC): MFMA pipe 0.32-0.37 busy, LDS bank-conflict ratio 0.02-0.07, 550-925
// TFLOP/s per layer.  A halo-tile variant (8 x 32 pixel output tile, the 10 x 34 input pixels staged once per 64-channel chunk
// for all 9 taps, 3x less L2 -> LDS traffic, one 512-thread workgroup per CU) was built, was correct, and was measured
// interleaved in one process against this kernel on every 3x3 layer: 0.93x-1.35x the time (faster only on conv42c) — its
// single workgroup per CU exposes the halo prologue and the epilogue, and its 8 lockstep waves all read their fragments right
// after each barrier.  It was removed; DESIGN.md section 4c has the numbers.  Also measured and dropped: issuing the next stage's
// LDS-DMA in quarters between the MFMA groups instead of in one burst after the barrier (+6 % time), a second stage of prefetch,
// and 32-channel K steps with 64-byte LDS rows (tools/ldsdma_depth.hip: the L2 -> LDS path moves 64-byte pieces at 2/3 the rate).
// Round 3, the epilogue: it costs 19 % of the family's time (skipping it: 5.28 -> 4.25 ms per step).  Two replacements built on the
// transposed product (A = filter rows, B = pixels: a lane then holds 4-channel groups of ITS pixel, see convb64 below), both
// correct on every bf16 test, neither kept: (a) all in registers - v_cvt_pk_bf16_f32, v_permlane32_swap, 16-byte stores, no LDS,
// no barrier: +5 % time (conv22c dgrad 0.219 -> 0.247 ms, upconv1 0.082 -> 0.114): a store instruction then touches 32 pixel rows
// with 32 bytes each instead of 8 whole rows; (b) packed bf16 through an 8 KiB LDS patch per wave in [pixel][channel] order, read
// back in memory order - 8 ds_write_b128 + 8 ds_read_b128 per lane instead of 64 ds_write_b32 + 32 ds_read_b128, the same
// whole-row stores, mask applied to the packed rows: -0.6 % on the family in a same-box A/B (3x3 forwards -2...-4 %, up-conv
// forwards +3...+10 %, some dgrads +6...+9 %).  With two workgroups per CU what the epilogue costs is its place at the end of a
// workgroup's life, not its LDS instruction count.
#include "common.hpp"
#include "igemm_epilogue.hpp"
#include <cstdio>
#include <cstdlib>

namespace unet {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;

__device__ __forceinline__ void bbuf_lds16(__amdgpu_buffer_rsrc_t r, unsigned char *lds, int voff, int soff)
{
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void *)lds, 16, voff, soff, 0, 0);
}

__device__ __forceinline__ float bf2f(u16 v) { return __builtin_bit_cast(float, (unsigned)v << 16); }
__device__ __forceinline__ u16 f2bf(float f) { return __builtin_bit_cast(u16, (__bf16)f); }

// ---- epilogue -----------------------------------------------------------------------------------------------------------
// acc[i][j]: 16x16 tiles of the wave's 64x64 block (rows = pixels, columns = output channels; C/D layout: column = lane & 15,
// row = 4 (lane >> 4) + r).  Per slab tm (tile rows 2 tm, 2 tm + 1) the wave writes 32 x 64 values to a private LDS patch (row pitch 68
// floats: 16-B aligned rows, conflict-free b128 read-back) and reads it back as rows: lane -> (row = lane >> 3 + 8k, 8 channels).
constexpr int EPB_PITCH = 68;
constexpr int EPB_WAVE_BYTES = 32 * EPB_PITCH * 4;          // 8704

// row tables of a linear-M tile: element offset of each tile row's pixel in dst | flags (bit 0: inside the deferred-ReLU
// window, bit 1: row outside the output domain)
template <int BM, class P>
__device__ __forceinline__ void igemmb_rows_linear(const P &p, int m0, int tid, unsigned *rowoff, unsigned char *rflag)
{
    if (tid < BM) {
        const bool relu_win = p.rw1 > p.rw0;
        int m = m0 + tid;
        const bool valid = m < p.M;
        m = valid ? m : p.M - 1;
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
        rowoff[tid] = off;
        rflag[tid] = flag | (valid ? 0 : 2);
    }
}

// TN = 32-column tiles per wave; the wave's tile rows start at `wrow0` of the workgroup tile, its columns at n0w.
// The row tables must be complete (barrier) before the call; `patch` is the wave's private LDS area (EPB_WAVE_BYTES).
// PF = row passes whose +add / mask operands are prefetched together (all of a slab's by default; the register-resident-filter
// kernel below has fewer registers to spare)
// The accumulators reach the store loop through a writer: write(tm, patch) puts slab tm (32 rows x 32 TN columns, bias added)
// of the wave's block into the patch.  Acc16: 16x16x32 MFMAs (column = lane & 15, row = 4 (lane >> 4) + r), two 16-row tiles per slab.
template <int TN, class P>
struct Acc16 {
    f32x4 (&acc)[4][2 * TN];
    float bv[2 * TN];
    __device__ __forceinline__ Acc16(const P &p, f32x4 (&a)[4][2 * TN], int n0w, int lane) : acc(a)
    {
        const int l15 = lane & 15;
#pragma unroll
        for (int j = 0; j < 2 * TN; ++j) {
            bv[j] = 0.f;
            if (p.bias) {
                int n = n0w + j * 16 + l15;
                n = n < p.Nn ? n : p.Nn - 1;
                bv[j] = p.bias[p.cout ? n % p.cout : n];
            }
        }
    }
    __device__ __forceinline__ void write(int tm, float *patch, int lane) const
    {
        const int l15 = lane & 15, kq = lane >> 4;
#pragma unroll
        for (int ii = 0; ii < 2; ++ii)
#pragma unroll
            for (int j = 0; j < 2 * TN; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    patch[(16 * ii + 4 * kq + r) * EPB_PITCH + j * 16 + l15] = acc[2 * tm + ii][j][r] + bv[j];
    }
};

template <int TN, int PF = 0, class P = IgemmP, class ACC = Acc16<TN, P>>
__device__ __forceinline__ void igemmb_store(const P &p, const ACC &accw, int wrow0, int n0w, int lane, float *patch,
                                             const unsigned *rowoff, const unsigned char *rflag)
{
    constexpr int NL = 4 * TN;                 // lanes per row on the read-back side (8 channels each)
    constexpr int RPP = 64 / NL;               // rows per pass
    const bool relu_win = p.rw1 > p.rw0;
    const int rrow = lane / NL, cg = lane % NL;
    const u16 *addp = (const u16 *)p.add, *maskp = (const u16 *)p.mask;
    u16 *dstp = (u16 *)p.dst;
    const int n8 = n0w + 8 * cg;
    const bool n_ok = n8 < p.Nn;
    const int nc = n_ok ? n8 : 0;
    int coloff;
    if (p.scatter != 1) {
        coloff = p.dn0 + nc;
    } else {
        const int ab = nc / p.cout;
        coloff = ((ab >> 1) * p.DW + (ab & 1)) * p.DC + p.dn0 + (nc - ab * p.cout);
    }
#pragma unroll
    for (int tm = 0; tm < 2; ++tm) {
        // the slab's +add / ReLU' mask operands first, all of them: their latency runs under the LDS transpose (issued one per
        // row pass inside the store loop they serialise against the stores - the compiler cannot prove dst != add/mask)
        constexpr int NPA = 32 / RPP;                         // row passes of the slab
        constexpr int NP = PF > 0 && PF < NPA ? PF : NPA;     // ... handled per chunk
        size_t o[NP];
        unsigned char fl[NP];
        uint4 ta[NP], tk[NP];
#pragma unroll
      for (int k0 = 0; k0 < NPA; k0 += NP) {
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const int trow = wrow0 + tm * 32 + rrow + RPP * (k0 + k);
            o[k] = (size_t)rowoff[trow] + (size_t)coloff;
            fl[k] = rflag[trow];
        }
        if (addp) {
#pragma unroll
            for (int k = 0; k < NP; ++k) ta[k] = *(const uint4 *)(addp + o[k]);
        }
        if (maskp) {
#pragma unroll
            for (int k = 0; k < NP; ++k) tk[k] = *(const uint4 *)(maskp + o[k]);
        }
        if (k0 == 0) accw.write(tm, patch, lane);
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const int prow = rrow + RPP * (k0 + k);
            const f32x4 lo = *(const f32x4 *)(patch + prow * EPB_PITCH + 8 * cg);
            const f32x4 hi = *(const f32x4 *)(patch + prow * EPB_PITCH + 8 * cg + 4);
            float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            if (addp) {
                const unsigned tw[4] = {ta[k].x, ta[k].y, ta[k].z, ta[k].w};
#pragma unroll
                for (int c = 0; c < 4; ++c) { v[2 * c] += bf2f((u16)(tw[c] & 0xffff)); v[2 * c + 1] += bf2f((u16)(tw[c] >> 16)); }
            }
            if (p.relu) {
                const bool defer = relu_win && (fl[k] & 1);
#pragma unroll
                for (int c = 0; c < 8; ++c) v[c] = (v[c] > 0.f || defer) ? v[c] : 0.f;
            }
            if (maskp) {
                const unsigned tw[4] = {tk[k].x, tk[k].y, tk[k].z, tk[k].w};
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    // a ReLU output: positive iff its bf16 pattern is neither zero nor negative
                    v[2 * c] = (short)(tw[c] & 0xffff) > 0 ? v[2 * c] : 0.f;
                    v[2 * c + 1] = (short)(tw[c] >> 16) > 0 ? v[2 * c + 1] : 0.f;
                }
            }
            if (n_ok && !(fl[k] & 2)) {
                uint4 w;
                w.x = (unsigned)f2bf(v[0]) | ((unsigned)f2bf(v[1]) << 16);
                w.y = (unsigned)f2bf(v[2]) | ((unsigned)f2bf(v[3]) << 16);
                w.z = (unsigned)f2bf(v[4]) | ((unsigned)f2bf(v[5]) << 16);
                w.w = (unsigned)f2bf(v[6]) | ((unsigned)f2bf(v[7]) << 16);
                *(uint4 *)(dstp + o[k]) = w;
            }
        }
      }
    }
}

template <int BM, int BN, class P, class A>
__device__ __forceinline__ void igemmb_epilogue(const P &p, A &acc, int m0, int n0, int tid, unsigned char *lds)
{
    constexpr int WN = BN / 64;
    unsigned *rowoff = (unsigned *)lds;
    unsigned char *rflag = lds + BM * 4;
    float *patch0 = (float *)(lds + BM * 4 + ((BM + 15) & ~15));
    igemmb_rows_linear<BM>(p, m0, tid, rowoff, rflag);
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const Acc16<2, P> w(p, acc, n0 + wn * 64, lane);
    igemmb_store<2, 0, P, Acc16<2, P>>(p, w, wm * 64, n0 + wn * 64, lane, patch0 + wave * (EPB_WAVE_BYTES / 4), rowoff, rflag);
}

// ---- plain kernel: every tap re-stages its A rows (up-conv GEMMs, and any 3x3 shape the halo kernel does not take) -------
// The products run as v_mfma_f32_16x16x32_bf16 (4 x 4 tiles of 16 x 16 per wave).  Round 3 had 32x32x16 (2 x 2 of 32 x 32): the
// same LDS reads and cycles per flop, but under bf16 MFMA load the part holds a higher clock on the 16 x 16 shape
// (MI355X_MICROARCH.md, DVFS give-back item 7): same-box A/B 9.75 -> 9.49 ms per step, PMC clock of this kernel 2.10-2.16 ->
// 2.31-2.38 GHz at MFMA busy 0.37-0.43 -> 0.36-0.41.
template <int BM, int BN, bool PAD>
__global__ __launch_bounds__(256, 2) void igemmb_kernel(const IgemmP p)
{
    constexpr int WN = BN / 64;
    constexpr int WM = 4 / WN;
    static_assert(WM * 64 == BM, "4 waves of 64x64");
    constexpr int RA = BM / 32, RB = BN / 32;   // staging rows per thread
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const IgEp ep = igb_epilogue_args(p);
    int a_TX = p.TX, a_T = p.T, a_nsrc = p.nsrc, a_oy0 = p.oy0, a_ox0 = p.ox0, a_stride = p.stride;
    IGB_PIN(a_TX); IGB_PIN(a_T); IGB_PIN(a_nsrc); IGB_PIN(a_oy0); IGB_PIN(a_ox0); IGB_PIN(a_stride);

    int logical;
    {
        const int nblk = gridDim.x, q = nblk >> 3, r = nblk & 7, xcd = blockIdx.x & 7;
        logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (blockIdx.x >> 3);
    }
    const int mt = logical / p.ntiles, nt = logical - mt * p.ntiles;
    const int m0 = mt * BM, n0 = nt * BN;

    // staging geometry: thread -> (row within a 32-row pass, 16-B chunk position); the SOURCE chunk is the swizzled one
    const int srow = tid >> 3;
    const int schunk = (tid & 7) ^ ((srow >> 1) & 7);
    const int coff = schunk * 8;                          // bf16 elements

    int a_off[RA], a_iy[RA], a_ix[RA];
    int b_off[RB];
#pragma unroll
    for (int j = 0; j < RB; ++j) {
        int n = n0 + srow + 32 * j;
        n = n < p.Nn ? n : p.Nn - 1;
        b_off[j] = (n * p.ldw + coff) * 2;                // bytes
    }

    int s = 0, ty = 0, tx = 0, kc = 0, kglob = 0, kbase = 0;
    int sH = 0, sW = 0, sC = 0, snch = 0, toff = 0;
    __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void *)p.src[0].p, 0, p.buf_bytes[0], 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc((void *)p.wt, 0, p.buf_bytes[2], 0x00020000);
    auto setup_source = [&](int si) {
        const GSrc &g = p.src[si];
        sH = g.H; sW = g.W; sC = g.C; snch = g.nch;
        rs_a = __builtin_amdgcn_make_buffer_rsrc((void *)g.p, 0, p.buf_bytes[si], 0x00020000);
        const int ohw = ep.OH * ep.OW;
#pragma unroll
        for (int i = 0; i < RA; ++i) {
            int m = m0 + srow + 32 * i;
            m = m < ep.M ? m : ep.M - 1;
            const int img = fdiv(m, ep.d_ohw);
            const int rem = m - img * ohw;
            const int oy = fdiv(rem, ep.d_ow);
            const int ox = rem - oy * ep.OW;
            const int iy = (oy + a_oy0) * a_stride - g.pad;
            const int ix = (ox + a_ox0) * a_stride - g.pad;
            a_iy[i] = iy; a_ix[i] = ix;
            a_off[i] = (((img * g.H + iy) * g.W + ix) * g.C + g.c0 + coff) * 2;      // bytes
        }
        toff = 0;
    };
    auto stage = [&](int buf) {
        unsigned char *abase = smem + buf * STAGE + wave * (8 * 128);
        unsigned char *bbase = abase + A_BYTES;
        // the tap / channel step goes into the VECTOR offset: the range check sees only that part, and a row whose tap-0
        // pixel lies in the virtual padding has a negative base although this tap's pixel is inside the tensor
        const int so = (toff + kc) * 2;
#pragma unroll
        for (int i = 0; i < RA; ++i) {
            int vo = a_off[i] + so;
            if (PAD) {
                const bool inb = (unsigned)(a_iy[i] + ty) < (unsigned)sH && (unsigned)(a_ix[i] + tx) < (unsigned)sW;
                vo = inb ? vo : (int)0x80000000;
            }
            bbuf_lds16(rs_a, abase + i * (32 * 128), vo, 0);
        }
#pragma unroll
        for (int j = 0; j < RB; ++j) bbuf_lds16(rs_b, bbase + j * (32 * 128), b_off[j], kglob * 2);
    };
    // K order of the loop: channel chunk outermost per source, taps innermost - consecutive steps re-read (almost) the same
    // pixels' same 128-byte pieces, shifted by one pixel or one row, while they are still in L2 (measured +0.5-3 % per layer
    // against taps outermost); the filter matrix keeps its [source][tap][channel] K order, so kglob is computed, not counted
    auto advance = [&]() {
        ++tx;
        if (tx == a_TX) { tx = 0; ++ty; }
        if (ty * a_TX + tx == a_T) {
            ty = 0; tx = 0;
            kc += 64;
            if (kc == snch) {
                kc = 0;
                kbase += a_T * snch;
                ++s;
                if (s < a_nsrc) setup_source(s);
            }
        }
        toff = (ty * sW + tx) * sC;
        kglob = kbase + (ty * a_TX + tx) * snch + kc;
    };

    const int nk = p.Kd >> 6;
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
    // operand fragment of a 16 x 16 x 32 MFMA: lane (row l15, k group kq) holds k = 8 kq .. 8 kq + 7 of the k32 step, i.e. the
    // 16-byte chunk 4 g + kq of its row; rows 16 i + l15 all carry the swizzle (l15 >> 1) & 7 (conflict-free: the 16 lanes a
    // ds_read_b128 serves together cover 8 row pairs x 2 chunk parities = 16 distinct slots of the 256-byte bank row)
    const int l15 = lane & 15, kq = lane >> 4;
    const int swz = (l15 >> 1) & 7;
    const int a_rd = (wm * 64 + l15) * 128;
    const int b_rd = A_BYTES + (wn * 64 + l15) * 128;
    setup_source(0);
    stage(0);
    advance();
    __syncthreads();            // drains the LDS-DMA (vmcnt(0)) and publishes buffer 0
    for (int ks = 0; ks < nk; ++ks) {
        const int cur = ks & 1;
        if (ks + 1 < nk) { stage(cur ^ 1); advance(); }
        const unsigned char *sb = smem + cur * STAGE;
        bf16x8 fa[2][4], fb[2][4];
        auto read_frags = [&](int g, int q) {
            const int pos = ((4 * g + kq) ^ swz) * 16;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                fa[q][i] = *(const bf16x8 *)(sb + a_rd + i * (16 * 128) + pos);
                fb[q][i] = *(const bf16x8 *)(sb + b_rd + i * (16 * 128) + pos);
            }
        };
        read_frags(0, 0);
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const int q = g & 1;
            if (g + 1 < 2) read_frags(g + 1, q ^ 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[q][i], fb[q][j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();        // next buffer landed (vmcnt(0)) and this one is free to overwrite
    }
    igemmb_epilogue<BM, BN>(ep, acc, m0, n0, tid, smem);
}

// ---- band kernel: the 3x3 launches with >= 128 input channels ----------------------------------------------------------------
// The plain kernel stages the tile's 128 A rows (pixels x 64 channels) once per tap: nine times 16 KiB per channel chunk, of which
// the three taps of a filter row are the same pixels shifted by one.  Here a filter row's taps share ONE staged band: with the
// output pixels numbered on the pitch of the input window (q = m + 2 (m / OW): two virtual columns per output row, so that
// q + 1 is the right-hand neighbour of q for every real pixel), the A row of pixel m at tap (ty, tx) is band row q(m) - q(m0) + tx of
// the band "tap (ty, 0) of the virtual pixels q(m0) ...".  The M tile stays 128 REAL pixels (nothing is computed for the virtual
// columns, the epilogue is the plain kernel's); a tile that crosses c output rows needs 130 + 2 c band rows, 144 are staged
// (launches with OW >= 19).  A band lives for three K steps and is fetched in two halves during the first two of the three steps
// before them, behind that step's filter tile: the wait at a step's end is a COUNTED vmcnt that retires the filters and leaves the
// band half in flight across the (raw) barrier - two steps of latency for the band rows, which come from HBM / Infinity Cache,
// while the filters are L2-hot.  Two band slots + two filter buffers = 68 KiB (two workgroups per CU), LDS-DMA bytes per step
// 32 -> 22 KiB.  The fragment
// reads of a lane are the plain kernel's with a per-lane row (its pixel's q) instead of the tile row: the swizzle follows
// the band row and is conflict-free for every shift (a row crossing inside a 16-row group costs a two-way conflict).
struct Igb3P {
    IgemmP p;
    FastDiv d_wv, d_oh;      // division by OW + 2 (virtual row pitch) and by OH
    int Wv, nbands;
};
constexpr int IGB3_BAND = 144;                       // staged band rows: 18 LDS-DMA instructions of 8 pixel rows

template <bool PAD>
__global__ __launch_bounds__(256, 2) void igemmb3_kernel(const Igb3P k)
{
    constexpr int BM = 128, BN = 128, WN = 2;
    constexpr int A_SLOT = IGB3_BAND * 128, B_BYTES = BN * 128, OFF_B = 2 * A_SLOT;
    constexpr int OOB = (int)0x80000000;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const IgemmP &p = k.p;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const IgEp ep = igb_epilogue_args(p);
    int a_nsrc = p.nsrc, a_oy0 = p.oy0, a_ox0 = p.ox0, a_Wv = k.Wv, a_nbands = k.nbands;
    IGB_PIN(a_nsrc); IGB_PIN(a_oy0); IGB_PIN(a_ox0); IGB_PIN(a_Wv); IGB_PIN(a_nbands);

    int logical;
    {
        const int nblk = gridDim.x, q = nblk >> 3, r = nblk & 7, xcd = blockIdx.x & 7;
        logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (blockIdx.x >> 3);
    }
    const int mt = logical / p.ntiles, nt = logical - mt * p.ntiles;
    const int m0 = mt * BM, n0 = nt * BN;
    const int grow0 = fdiv(m0, ep.d_ow);
    const int q0 = m0 + 2 * grow0;

    // filter staging: as the plain kernel (thread -> row within a 32-row pass, swizzled source chunk)
    const int srow = tid >> 3;
    int b_off[4];
    {
        const int coff = ((tid & 7) ^ ((srow >> 1) & 7)) * 8;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int n = n0 + srow + 32 * j;
            n = n < p.Nn ? n : p.Nn - 1;
            b_off[j] = (n * p.ldw + coff) * 2;
        }
    }
    const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc((void *)p.wt, 0, p.buf_bytes[2], 0x00020000);

    // band staging: the 18 instructions of a band (G: band rows 8 G .. 8 G + 7) go out in two halves, during the first and the
    // second K step of the band before; entry e = 3 h + kk of a thread is G = 9 h + wave + 4 kk (kk = 2: wave 0 only)
    const int prow = lane >> 3;
    int a_off[6], a_iy[6], a_ix[6];
    // the band being fetched ("next"): source, channel chunk, filter row
    int s = 0, kc = 0, ty = 0, kbase = 0;
    int sH = 0, sW = 0, sC = 0, snch = 0;
    __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void *)p.src[0].p, 0, p.buf_bytes[0], 0x00020000);
    auto setup_source = [&](int si) {
        const GSrc &g = p.src[si];
        sH = g.H; sW = g.W; sC = g.C; snch = g.nch;
        rs_a = __builtin_amdgcn_make_buffer_rsrc((void *)g.p, 0, p.buf_bytes[si], 0x00020000);
        const int nrows = p.NB * ep.OH;
#pragma unroll
        for (int e = 0; e < 6; ++e) {
            const int G = 9 * (e / 3) + wave + 4 * (e % 3);
            const int r = 8 * G + prow;
            const int acoff = ((lane & 7) ^ (((r >> 1) & 3) << 1)) * 8;   // source chunk of this LDS position (swizzle by band row, below)
            const int q = q0 + r;
            const int grow = fdiv(q, k.d_wv);
            const int oxv = q - grow * a_Wv;
            const int img = fdiv(grow, k.d_oh);
            const int oy = grow - img * ep.OH;
            const int iy = oy + a_oy0 - g.pad;
            const int ix = oxv + a_ox0 - g.pad;
            a_iy[e] = iy; a_ix[e] = ix;
            a_off[e] = grow < nrows ? (((img * g.H + iy) * g.W + ix) * g.C + g.c0 + acoff) * 2 : OOB;
            if (PAD && grow >= nrows) a_ix[e] = -1;
        }
    };
    auto stage_a = [&](auto half_tag, int slot) {
        constexpr int HF = decltype(half_tag)::value;
        const int so = (ty * sW * sC + kc) * 2;
        unsigned char *base = smem + slot * A_SLOT + (9 * HF + wave) * 1024;
#pragma unroll
        for (int kk = 0; kk < 3; ++kk) {
            if (kk < 2 || wave == 0) {
                const int e = 3 * HF + kk;
                int vo = a_off[e] + so;
                if (PAD) {
                    const bool inb = (unsigned)(a_iy[e] + ty) < (unsigned)sH && (unsigned)a_ix[e] < (unsigned)sW;
                    vo = inb ? vo : OOB;
                }
                bbuf_lds16(rs_a, base + kk * 4096, vo, 0);
            }
        }
    };
    auto stage_b = [&](int kglob, int buf) {
        unsigned char *bbase = smem + OFF_B + buf * B_BYTES + wave * (8 * 128);
#pragma unroll
        for (int j = 0; j < 4; ++j) bbuf_lds16(rs_b, bbase + j * (32 * 128), b_off[j], kglob * 2);
    };
    auto advance = [&]() {
        ++ty;
        if (ty == 3) {
            ty = 0;
            kc += 64;
            if (kc == snch) {
                kc = 0;
                kbase += 9 * snch;
                ++s;
                if (s < a_nsrc) setup_source(s);
            }
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;

    // fragment reads: lane (l15, kq) of A tile i reads band row R = q(m) - q0 + tx of its pixel m, chunk (4 g + kq) ^ 2 ((R >> 1) & 3).
    // The swizzle must be conflict-free for every shift tx: a ds_read_b128 serves 16 lanes together - 16 consecutive rows, the k
    // groups kq and kq ^ 1 in a fixed pattern (row pairs 0, 1, 6, 7 of the sixteen one group, 2 ... 5 the other).  The plain
    // kernel's (R >> 1) & 7 relies on that pattern staying aligned with the row pairs; shifted by one row it collides two ways
    // (PMC: LDS conflict cycles 0.26 of all).  XORing bits 1-2 only leaves bit 0 = kq: two lanes of one row parity then differ
    // in the row pair mod 4 or, four pairs apart, in kq - whatever the shift.
    const int l15 = lane & 15, kq = lane >> 4;
    int adr_a[4][3][2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int m = m0 + wm * 64 + 16 * i + l15;
        m = m < ep.M ? m : ep.M - 1;
        const int R0 = (m - m0) + 2 * (fdiv(m, ep.d_ow) - grow0);
#pragma unroll
        for (int tx = 0; tx < 3; ++tx) {
            const int R = R0 + tx;
#pragma unroll
            for (int g = 0; g < 2; ++g) adr_a[i][tx][g] = R * 128 + (((4 * g + kq) ^ (((R >> 1) & 3) << 1)) << 4);
        }
    }
    const int swz = (l15 >> 1) & 7;
    const int b_rd = OFF_B + (wn * 64 + l15) * 128;
    const int pos_b0 = (kq ^ swz) * 16, pos_b1 = ((4 + kq) ^ swz) * 16;

    setup_source(0);
    stage_a(std::integral_constant<int, 0>{}, 0);
    stage_a(std::integral_constant<int, 1>{}, 0);
    stage_b(0, 0);
    int cur_kg0 = 0, cur_snch = snch;
    advance();
    __syncthreads();            // drains the LDS-DMA (vmcnt(0)) and publishes band 0 and filter buffer 0

    // one K step: tap (band's filter row, TX) from band slot SLOT and filter buffer BB (all static: immediate LDS offsets)
    auto step = [&](auto slot_tag, auto tx_tag, int b) {
        constexpr int SLOT = decltype(slot_tag)::value, TX = decltype(tx_tag)::value;
        constexpr int BB = (SLOT + TX) & 1;                    // step 3 b + TX, b of SLOT's parity
        const bool more = b + 1 < a_nbands;
        // filters of the next step first, then (steps 0 and 1 of a band) one half of the next band: the wait at the end of the step
        // leaves this step's band half in flight - it is retired by the next step's wait (or by step 2's vmcnt(0))
        if (TX < 2) stage_b(cur_kg0 + (TX + 1) * cur_snch, BB ^ 1);
        else if (more) stage_b(kbase + ty * 3 * snch + kc, BB ^ 1);
        const bool band_dma = TX < 2 && more;
        if constexpr (TX < 2) { if (band_dma) stage_a(tx_tag, SLOT ^ 1); }
        const unsigned char *sa = smem + SLOT * A_SLOT;
        const unsigned char *sb = smem + BB * B_BYTES;
        bf16x8 fa[2][4], fb[2][4];
        auto read_frags = [&](auto g_tag, int q) {
            constexpr int g = decltype(g_tag)::value;
            const int pos = g ? pos_b1 : pos_b0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                fa[q][i] = *(const bf16x8 *)(sa + adr_a[i][TX][g]);
                fb[q][i] = *(const bf16x8 *)(sb + b_rd + i * (16 * 128) + pos);
            }
        };
        read_frags(std::integral_constant<int, 0>{}, 0);
        read_frags(std::integral_constant<int, 1>{}, 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < 2; ++g) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[g][i], fb[g][j], acc[i][j], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        // the next step's filters have landed (this wave's part; the barrier publishes everybody's) and this step's buffers are free.
        // A raw barrier: __syncthreads() would drain the band half with vmcnt(0)
        if (band_dma) {
            if (wave == 0) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
    };
    auto band = [&](auto slot_tag, int b) {
        step(slot_tag, std::integral_constant<int, 0>{}, b);
        step(slot_tag, std::integral_constant<int, 1>{}, b);
        step(slot_tag, std::integral_constant<int, 2>{}, b);
        if (b + 1 < a_nbands) {
            cur_kg0 = kbase + ty * 3 * snch + kc; cur_snch = snch;
            advance();
        }
    };
    for (int b = 0; b < a_nbands; b += 2) {
        band(std::integral_constant<int, 0>{}, b);
        if (b + 1 < a_nbands) band(std::integral_constant<int, 1>{}, b + 1);
    }
    igemmb_epilogue<BM, BN>(ep, acc, m0, n0, tid, smem);
}

// =========================================================================================================================
// convb64: the 64-input-channel 3x3 layers (conv12c, conv21c, conv12e forward; conv12c, conv11e, conv12e dgrad) with bf16
// tensors.  K = 9 x 64 is nine steps of the kernel above - less time than a workgroup costs around them - and every input
// pixel is staged once per tap and per n-tile: those launches sit at 0.25-0.3 of the HBM roof that binds them.  Here
//   * a workgroup is PERSISTENT over output tiles of 8 x 32 pixels x 64 output channels, one workgroup per CU, ONE WAVE PER
//     SIMD with the whole 512-entry register file;
//   * its 64 x 576 filter block (72 KiB) is staged once by LDS-DMA and stays in LDS ([tap][n][64 channels], 16-byte chunks
//     XOR-swizzled);
//   * per tile the 10 x 34 input halo is fetched ONCE for all nine taps - by plain buffer loads into registers at the top of
//     the previous tile's MFMA loop, written to the other LDS buffer after it.  (LDS-DMA costs a wave ~300 cycles of issue per
//     instruction, 11 per tile; with one wave per SIMD nobody else feeds the matrix pipe meanwhile: measured 3300 of a tile's
//     13700 cycles.  A load into registers issues in a few cycles, and a lone wave has the registers to park a whole halo.)
//     Pixel rows of 128 B, chunks swizzled by the halo pixel index: the 32 pixels of an MFMA tile are consecutive halo pixels
//     for every tap shift -> conflict-free ds_read_b128;
//   * a wave owns two tile rows of 32 pixels x 64 channels.  The MFMA runs TRANSPOSED - A = filters, B = pixels - so that in
//     the C/D layout a lane holds, for its pixel, groups of 4 consecutive output channels: bias / ReLU / ReLU' mask / +add
//     and the bf16 rounding happen in registers and the result is stored from there in 8-byte pieces.  No LDS transpose,
//     no second barrier: the LDS-staged epilogue of the kernel above cost this one 4000 of 13700 cycles (a dependent chain
//     of LDS round trips that a lone wave cannot hide).
// Cycle stamps of a tile (conv12c forward, 40 tiles per workgroup): set-up 1590, MFMA loop 5140 (35.7 per MFMA), halo write 700,
// epilogue 1890, barrier 260.  Finishing the previous tile's results INSIDE the next tile's MFMA loop (two accumulator sets, one
// 4-channel group per step) was built and measured: the loop grows by what the epilogue shrinks (6200 + 2550 of set-up: 9920
// against 9570 per tile) - a step's four MFMAs leave a lone wave ~50 free issue cycles, a group needs more.  Not kept.
// One barrier per tile.  The MFMA loop itself (36 steps of 4 ds_read_b128 + 4 MFMAs, fragments requested two steps ahead)
// runs at 33 cycles per MFMA (tools/mfma_bf16_lds.hip: 34.8 for this mix on its own).
// =========================================================================================================================
#ifndef CB64_DEPTH
#define CB64_DEPTH 3
#endif
struct CB64P {
    IgemmP p;
    int tx_n, ty_n, tiles;       // tiles per row / per column / in total (all images)
    int dbg;                     // UNET_CB64_DBG & 8: in-kernel cycle stamps per phase (workgroup 0, wave 0)
    FastDiv d_tpi, d_tx;
};

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef short s16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b)
{
    const f32x2v f = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f, bf16x2));
}
__device__ __forceinline__ unsigned pk_relu_bf16(unsigned v)
{
    const s16x2 z = {0, 0};
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, v), z));
}

template <int TH, int TW>
struct CB64Geom {
    static constexpr int HWp = TW + 2, HHp = TH + 2, NPIX = HHp * HWp;
    static constexpr int NINST = (NPIX + 7) / 8;               // 1-KiB pieces of a halo (8 pixels of 128 B each)
    static constexpr int HALO_BYTES = NINST * 1024;
    static constexpr int W_BYTES = 9 * 64 * 128;               // 73728
    static constexpr int LDS = 2 * HALO_BYTES + W_BYTES;       // [halo 0][halo 1][filters]
    static constexpr int NII = (NINST + 3) / 4;                // pieces per wave
    static_assert(LDS <= 160 * 1024, "one workgroup per CU");
};

__device__ unsigned long long g_cb64_stamps[8];      // UNET_CB64_DBG & 8: cycles per phase (timing experiments)

template <int TH, int TW, bool HAS_ADD, bool HAS_MASK>
__global__ __launch_bounds__(256, 1) void convb64_kernel(const CB64P k)
{
    using G = CB64Geom<TH, TW>;
    static_assert(TH == 8 && TW == 32, "a wave owns two tile rows of 32 pixels: each is one MFMA column block");
    unsigned long long stamp[6] = {0, 0, 0, 0, 0, 0}, tprev = __builtin_readcyclecounter();
    auto mark = [&](int i) { const unsigned long long now = __builtin_readcyclecounter(); stamp[i] += now - tprev; tprev = now; };
    const IgemmP &p = k.p;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nblk = p.Nn >> 6;
    const int nb = blockIdx.x % nblk, wg = blockIdx.x / nblk, nwg = gridDim.x / nblk;
    const int n0 = nb * 64;
    const GSrc &g0 = p.src[0];
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void *)p.wt, 0, p.buf_bytes[2], 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void *)g0.p, 0, p.buf_bytes[0], 0x00020000);
    constexpr int OOB = (int)0x80000000;
    const int l31 = lane & 31, lh = lane >> 5;
    const int tpi = k.tx_n * k.ty_n;

    // ---- halo roles, fixed for the kernel: piece i = wave + 4 ii covers halo pixels 8i .. 8i+7; lane -> (pixel P, 16-byte slot
    // holding channel chunk slot ^ swz(P)).  hrel = byte offset of that piece relative to the halo's first pixel, hpos =
    // (hy << 8 | hx), or -1 past the halo's end.  Per tile only the base offset and - on border tiles - the in-tensor test remain.
    int hrel[G::NII], hpos[G::NII];
    {
        const int sub = lane >> 3, slot = lane & 7;
#pragma unroll
        for (int ii = 0; ii < G::NII; ++ii) {
            const int P = 8 * (wave + 4 * ii) + sub;
            const int hy = P / G::HWp, hx = P - hy * G::HWp;
            const int c = slot ^ ((P >> 1) & 7);
            hrel[ii] = ((hy * g0.W + hx) * g0.C + c * 8) * 2;
            hpos[ii] = (P < G::NPIX && wave + 4 * ii < G::NINST) ? (hy << 8 | hx) : -1;
        }
    }
    struct TileGeo { int img, tyb, txb, iy0, ix0, base; bool interior; };
    auto tile_geo = [&](int t) {
        TileGeo q;
        q.img = fdiv(t, k.d_tpi);
        const int trem = t - q.img * tpi;
        q.tyb = fdiv(trem, k.d_tx);
        q.txb = trem - q.tyb * k.tx_n;
        q.iy0 = q.tyb * TH + p.oy0 - g0.pad; q.ix0 = q.txb * TW + p.ox0 - g0.pad;
        q.base = (((q.img * g0.H + q.iy0) * g0.W + q.ix0) * g0.C + g0.c0) * 2;            // may be "negative" on border tiles
        q.interior = q.iy0 >= 0 && q.ix0 >= 0 && q.iy0 + G::HHp <= g0.H && q.ix0 + G::HWp <= g0.W;
        return q;
    };
    u32x4 hreg[G::NII];                  // the next tile's halo pieces on their way from memory to LDS
    auto halo_load = [&](const TileGeo &q, bool live) {
#pragma unroll
        for (int ii = 0; ii < G::NII; ++ii) {
            bool ok = live && hpos[ii] >= 0;
            if (!q.interior) {
                const int iy = q.iy0 + (hpos[ii] >> 8), ix = q.ix0 + (hpos[ii] & 255);
                ok = ok && (unsigned)iy < (unsigned)g0.H && (unsigned)ix < (unsigned)g0.W;
            }
            hreg[ii] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, ok ? q.base + hrel[ii] : OOB, 0, 0);     // zeros outside the tensor
        }
    };
    auto halo_store = [&](unsigned char *hl) {
#pragma unroll
        for (int ii = 0; ii < G::NII; ++ii)
            if (wave + 4 * ii < G::NINST) *(u32x4 *)(hl + (wave + 4 * ii) * 1024 + lane * 16) = hreg[ii];
    };

    // ---- filters, once, resident in LDS behind the halo buffers: [tap][n][128 B], rows 8i .. 8i+7 per instruction, chunks
    // swizzled by the row's n (the barrier in front of the tile loop publishes them)
    unsigned char *wl = smem + 2 * G::HALO_BYTES;
    {
        const int sub = lane >> 3, slot = lane & 7;
#pragma unroll
        for (int ii = 0; ii < 18; ++ii) {
            const int i = wave + 4 * ii;
            const int R = 8 * i + sub;
            const int tap = R >> 6, n = R & 63;
            const int c = slot ^ ((n >> 1) & 7);
            bbuf_lds16(rs_w, wl + i * 1024, ((n0 + n) * p.ldw + tap * 64 + c * 8) * 2, 0);
        }
    }
    int bw[2];                           // filter fragment base of n-block nbk (without tap / k step): row n = 32 nbk + l31
#pragma unroll
    for (int nbk = 0; nbk < 2; ++nbk) {
        const int n = 32 * nbk + l31;
        bw[nbk] = n * 128 + ((lh ^ ((n >> 1) & 7)) << 4);
    }
    // halo pixel of this lane's pixel in column block j (tile row 2 wave + j, column l31) at tap (0,0)
    int pj[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) pj[j] = (2 * wave + j) * G::HWp + l31;
    // epilogue role = the C/D layout of the transposed product: lane (pixel l31, half lh) holds output channels
    // 32 nbk + 8 g + 4 lh + {0..3} in accumulator registers 4 g .. 4 g + 3 of acc[nbk][j]
    float bv[2][4][4];
#pragma unroll
    for (int nbk = 0; nbk < 2; ++nbk)
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int c = 0; c < 4; ++c) bv[nbk][g][c] = p.bias ? p.bias[n0 + 32 * nbk + 8 * g + 4 * lh + c] : 0.f;
    const u16 *addp = (const u16 *)p.add, *maskp = (const u16 *)p.mask;
    u16 *dstp = (u16 *)p.dst;

    int t = wg, it = 0;
    if (t < k.tiles) {
        halo_load(tile_geo(t), true);
        halo_store(smem);
    }
    __syncthreads();                       // first halo (written above) and the filters (vmcnt(0)) are in LDS
    for (; t < k.tiles; t += nwg, ++it) {
        const int cur = it & 1;
        unsigned char *hl = smem + cur * G::HALO_BYTES;
        const bool more = t + nwg < k.tiles;
        const TileGeo qc = tile_geo(t);
        halo_load(tile_geo(more ? t + nwg : t), more);        // lands in registers while the MFMA loop runs

        // destination (element offsets; the launcher checks the tensor against 2^31 elements) of the lane's pixel in the wave's
        // two tile rows, and the ReLU' mask / +add operands of its 2 x 8 channel groups: requested now, used after the loop
        unsigned eo[2];
        bool ev[2];
        u32x2 tk[2][2][4], ta[2][2][4];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            int oy = qc.tyb * TH + 2 * wave + j, ox = qc.txb * TW + l31;
            ev[j] = oy < p.OH && ox < p.OW;
            oy = oy < p.OH ? oy : p.OH - 1; ox = ox < p.OW ? ox : p.OW - 1;
            eo[j] = (unsigned)((qc.img * p.DH + oy) * p.DW + ox) * (unsigned)p.DC + (unsigned)(p.dn0 + n0);
#pragma unroll
            for (int nbk = 0; nbk < 2; ++nbk)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    if (HAS_MASK) tk[j][nbk][g] = *(const u32x2 *)(maskp + eo[j] + 32 * nbk + 8 * g + 4 * lh);
                    if (HAS_ADD) ta[j][nbk][g] = *(const u32x2 *)(addp + eo[j] + 32 * nbk + 8 * g + 4 * lh);
                }
        }

        f32x16 acc[2][2];                  // [nbk][j]: rows = output channels, columns = pixels; starts at the bias
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = bv[i][r >> 2][r & 3];
        if (k.dbg & 8) mark(0);            // set-up of the tile

        // ---- 36 steps (9 taps x 4 k steps of 16 channels), fully unrolled.  One wave per SIMD: nobody else hides the LDS
        // latency, so the fragments of step s + DEPTH - 1 are requested before the 4 MFMAs of step s issue
        constexpr int DEPTH = CB64_DEPTH;
        bf16x8 fa[DEPTH][2], fb[DEPTH][2];
        auto read_frags = [&](int st, int slot) {
            const int tap = st >> 2, x = (st & 3) << 5;
            const int ty = tap / 3, tx = tap - 3 * ty;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int P = pj[j] + (ty * G::HWp + tx);
                fa[slot][j] = *(const bf16x8 *)(hl + ((P * 128 + ((lh ^ ((P >> 1) & 7)) << 4)) ^ x));
            }
#pragma unroll
            for (int nbk = 0; nbk < 2; ++nbk) fb[slot][nbk] = *(const bf16x8 *)(wl + ((bw[nbk] + tap * 8192) ^ x));
        };
#pragma unroll
        for (int st = 0; st < DEPTH - 1; ++st) read_frags(st, st);
#pragma unroll
        for (int st = 0; st < 36; ++st) {
            if (st + DEPTH - 1 < 36) read_frags(st + DEPTH - 1, (st + DEPTH - 1) % DEPTH);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int nbk = 0; nbk < 2; ++nbk)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[nbk][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[st % DEPTH][nbk], fa[st % DEPTH][j], acc[nbk][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (k.dbg & 8) mark(1);            // MFMA loop

        // ---- the next tile's halo: registers -> the other buffer (every wave finished reading it before the last barrier)
        halo_store(smem + (cur ^ 1) * G::HALO_BYTES);
        if (k.dbg & 8) mark(2);            // wait for the loads + LDS writes

        // ---- epilogue in registers: +add, ReLU, ReLU' mask, one rounding to bf16.  A lane holds 4-channel groups 8 g + 4 lh
        // of its pixel; v_permlane32_swap exchanges groups with the lane of the other half so that each ends up with two runs
        // of 8 consecutive channels: 16-byte stores (lh = 0: channels 0-7 and 16-23 of the n-block, lh = 1: 8-15 and 24-31)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int nbk = 0; nbk < 2; ++nbk) {
                unsigned d[4][2];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float v[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) v[c] = acc[nbk][j][4 * g + c];
                    if (HAS_ADD) {
                        const unsigned w0 = ta[j][nbk][g].x, w1 = ta[j][nbk][g].y;
                        v[0] += bf2f((u16)(w0 & 0xffff)); v[1] += bf2f((u16)(w0 >> 16));
                        v[2] += bf2f((u16)(w1 & 0xffff)); v[3] += bf2f((u16)(w1 >> 16));
                    }
                    if (HAS_MASK) {
                        // a ReLU output: positive iff its bf16 pattern is neither zero nor negative
                        const unsigned w0 = tk[j][nbk][g].x, w1 = tk[j][nbk][g].y;
                        v[0] = (short)(w0 & 0xffff) > 0 ? v[0] : 0.f; v[1] = (short)(w0 >> 16) > 0 ? v[1] : 0.f;
                        v[2] = (short)(w1 & 0xffff) > 0 ? v[2] : 0.f; v[3] = (short)(w1 >> 16) > 0 ? v[3] : 0.f;
                    }
                    // one v_cvt_pk_bf16_f32 per pair (RNE); ReLU on the packed result: a negative bf16 is a negative int16, and
                    // rounding is monotone and keeps the sign, so max(round(x), 0) == round(max(x, 0))
                    d[g][0] = cvt_pk_bf16(v[0], v[1]);
                    d[g][1] = cvt_pk_bf16(v[2], v[3]);
                }
                if (p.relu) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) { d[g][0] = pk_relu_bf16(d[g][0]); d[g][1] = pk_relu_bf16(d[g][1]); }
                }
#pragma unroll
                for (int gp = 0; gp < 2; ++gp)
#pragma unroll
                    for (int w = 0; w < 2; ++w) {
                        // lanes 32-63 of d[2gp] <-> lanes 0-31 of d[2gp+1]
                        const auto r = __builtin_amdgcn_permlane32_swap(d[2 * gp][w], d[2 * gp + 1][w], false, false);
                        d[2 * gp][w] = r[0]; d[2 * gp + 1][w] = r[1];
                    }
                if (ev[j]) {
#pragma unroll
                    for (int gp = 0; gp < 2; ++gp) {
                        u32x4 w = {d[2 * gp][0], d[2 * gp][1], d[2 * gp + 1][0], d[2 * gp + 1][1]};
                        *(u32x4 *)(dstp + eo[j] + 32 * nbk + 16 * gp + 8 * lh) = w;
                    }
                }
            }
        if (k.dbg & 8) mark(3);            // epilogue
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // this wave's halo writes are in LDS
        __builtin_amdgcn_s_barrier();                              // (no vmcnt wait: the stores stay in flight)
        if (k.dbg & 8) mark(4);            // barrier
    }
    if ((k.dbg & 8) && blockIdx.x == 0 && tid == 0) {
        for (int i = 0; i < 5; ++i) atomicAdd(&g_cb64_stamps[i], stamp[i]);
        atomicAdd(&g_cb64_stamps[7], 1ull);
    }
}

static bool convb64_applicable(const IgemmP &p)
{
    return p.T == 9 && p.TX == 3 && p.stride == 1 && p.nsrc == 1 && p.src[0].nch == 64 && p.Kd == 576 && p.Nn % 64 == 0 &&
           !p.scatter && !(p.rw1 > p.rw0) && !p.pool_dst && p.DH == p.OH && p.DW == p.OW;
}

template <int TH, int TW, bool HAS_ADD, bool HAS_MASK>
static int launch_convb64_t(const IgemmP &p, hipStream_t st)
{
    using G = CB64Geom<TH, TW>;
    static bool attr_done[64] = {false};
    auto kern = convb64_kernel<TH, TW, HAS_ADD, HAS_MASK>;
    if (int rc_ = ensure_dynamic_lds((const void *)kern, G::LDS, attr_done)) return rc_;
    CB64P k;
    k.p = p;
    k.tx_n = cdiv(p.OW, TW); k.ty_n = cdiv(p.OH, TH);
    k.tiles = p.NB * k.tx_n * k.ty_n;
    k.d_tpi = make_fastdiv((unsigned)(k.tx_n * k.ty_n));
    k.d_tx = make_fastdiv((unsigned)k.tx_n);
    static const int dbg = [] { const char *e = getenv("UNET_CB64_DBG"); return e ? atoi(e) : 0; }();
    k.dbg = dbg;
    const int nblk = p.Nn / 64;
    int cus = 256;
    {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) cus = n;
    }
    int per = k.tiles < cus ? k.tiles : cus;          // persistent workgroups per n-block: one per CU
    char tag[96];
    snprintf(tag, sizeof(tag), "convb64<%d;%d> M=%d N=%d tiles=%d wgs=%d pad=%d", TH, TW, p.M, p.Nn, k.tiles, per * nblk, p.src[0].pad);
    prof_begin(PK_IGEMM, tag, st, igemm_alg_flops(p), 2.0 * (double)k.tiles * (TH * TW) * p.Nn * 576.0, igemm_alg_bytes(p) / 2.0);
    hipLaunchKernelGGL(kern, dim3(per * nblk), dim3(256), G::LDS, st, k);
    prof_end(st);
    HIP_TRY(hipGetLastError());
    if (dbg & 8) {
        unsigned long long h[8];
        HIP_TRY(hipStreamSynchronize(st));
        HIP_TRY(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_cb64_stamps), sizeof(h)));
        fprintf(stderr, "cb64 stamps %s: setup %llu mfma %llu halo-write %llu epilogue %llu barrier %llu (cycles, cumulative over %llu launches)\n", tag, h[0], h[1], h[2], h[3], h[4], h[7]);
    }
    return 0;
}

template <int TH, int TW>
static int launch_convb64(const IgemmP &p, hipStream_t st)
{
    if (p.add && p.mask) return launch_convb64_t<TH, TW, true, true>(p, st);
    if (p.add) return launch_convb64_t<TH, TW, true, false>(p, st);
    if (p.mask) return launch_convb64_t<TH, TW, false, true>(p, st);
    return launch_convb64_t<TH, TW, false, false>(p, st);
}

template <int BM, int BN, bool PAD>
static int launch_cfgb(const IgemmP &p, hipStream_t st)
{
    constexpr int STAGES = 2 * (BM + BN) * 128;
    constexpr int EPI = BM * 4 + ((BM + 15) & ~15) + 4 * EPB_WAVE_BYTES;
    constexpr int LDS = STAGES > EPI ? STAGES : EPI;
    static bool attr_done[64] = {false};
    auto kern = igemmb_kernel<BM, BN, PAD>;
    if (int rc_ = ensure_dynamic_lds((const void *)kern, LDS, attr_done)) return rc_;
    IgemmP q = p;
    q.mtiles = cdiv(p.M, BM);
    q.ntiles = cdiv(p.Nn, BN);
    char tag[96];
    snprintf(tag, sizeof(tag), "igemmb<%d;%d;%d> M=%d N=%d Kd=%d T=%d s=%d nsrc=%d", BM, BN, (int)PAD, p.M, p.Nn, p.Kd, p.T, p.stride, p.nsrc);
    prof_begin(PK_IGEMM, tag, st, igemm_alg_flops(p), 2.0 * q.mtiles * BM * (double)q.ntiles * BN * p.Kd, igemm_alg_bytes(p) / 2.0);   // every tensor is 2 B/element
    hipLaunchKernelGGL(kern, dim3(q.mtiles * q.ntiles), dim3(256), LDS, st, q);
    prof_end(st);
    HIP_TRY(hipGetLastError());
    return 0;
}

static bool igemmb3_applicable(const IgemmP &p)
{
    if (!(p.T == 9 && p.TX == 3 && p.stride == 1 && p.Nn % 128 == 0 && p.OW >= 19)) return false;
    for (int i = 0; i < p.nsrc; ++i)
        if (p.src[i].nch % 64) return false;
    // virtual pixel indices (two extra columns per output row) and the band rows past the last tile stay below 2^31
    return (size_t)p.NB * p.OH * (p.OW + 2) + 1024 < 0x7FFFFFFFull;
}

template <bool PAD>
static int launch_igemmb3(const IgemmP &p, hipStream_t st)
{
    constexpr int BM = 128, BN = 128;
    constexpr int STAGES = 2 * IGB3_BAND * 128 + 2 * BN * 128;
    constexpr int EPI = BM * 4 + ((BM + 15) & ~15) + 4 * EPB_WAVE_BYTES;
    constexpr int LDS = STAGES > EPI ? STAGES : EPI;
    static bool attr_done[64] = {false};
    auto kern = igemmb3_kernel<PAD>;
    if (int rc_ = ensure_dynamic_lds((const void *)kern, LDS, attr_done)) return rc_;
    Igb3P k;
    k.p = p;
    k.p.mtiles = cdiv(p.M, BM);
    k.p.ntiles = cdiv(p.Nn, BN);
    k.Wv = p.OW + 2;
    k.nbands = p.Kd / 192;
    k.d_wv = make_fastdiv((unsigned)k.Wv);
    k.d_oh = make_fastdiv((unsigned)p.OH);
    char tag[96];
    snprintf(tag, sizeof(tag), "igemmb3<%d> M=%d N=%d Kd=%d OW=%d nsrc=%d", (int)PAD, p.M, p.Nn, p.Kd, p.OW, p.nsrc);
    prof_begin(PK_IGEMM, tag, st, igemm_alg_flops(p), 2.0 * k.p.mtiles * BM * (double)k.p.ntiles * BN * p.Kd, igemm_alg_bytes(p) / 2.0);
    hipLaunchKernelGGL(kern, dim3(k.p.mtiles * k.p.ntiles), dim3(256), LDS, st, k);
    prof_end(st);
    HIP_TRY(hipGetLastError());
    return 0;
}

// p has passed launch_igemm's generic checks; tensors are bf16
int launch_igemmb(IgemmP p, bool pad, hipStream_t st)
{
    for (int i = 0; i < p.nsrc; ++i) {
        ARG_CHECK(p.src[i].nch % 64 == 0, "igemmb: channel count %d is not a multiple of 64", p.src[i].nch);
        ARG_CHECK(p.src[i].C % 8 == 0 && p.src[i].c0 % 8 == 0, "igemmb: channel pitch/offset must be multiples of 8");
    }
    ARG_CHECK(p.ldw % 8 == 0 && p.DC % 8 == 0 && p.dn0 % 8 == 0 && p.Nn % 8 == 0 && (!p.scatter || p.cout % 8 == 0), "igemmb: 16-byte accesses need channel counts that are multiples of 8");
    ARG_CHECK((size_t)p.NB * p.DH * p.DW * p.DC < 0x7FFFFFFFull, "igemmb: destination exceeds 31-bit element offsets");
    for (int i = 0; i < 3; ++i) p.buf_bytes[i] = 0;
    for (int i = 0; i < p.nsrc; ++i) {
        const size_t b = (size_t)p.NB * p.src[i].H * p.src[i].W * p.src[i].C * 2;
        ARG_CHECK(b < 0x7FFFFFFFull, "igemmb: source tensor exceeds 2 GiB");
        p.buf_bytes[i] = (int)b;
    }
    {
        const size_t b = (size_t)p.Nn * p.ldw * 2;
        ARG_CHECK(b < 0x7FFFFFFFull, "igemmb: filter matrix exceeds 2 GiB");
        p.buf_bytes[2] = (int)b;
    }
    static const int use_cb64 = [] { const char *e = getenv("UNET_CONVB64"); return e ? atoi(e) : 1; }();
    if (use_cb64 && convb64_applicable(p)) {
        return launch_convb64<8, 32>(p, st);
    }
    static const int use_band = [] { const char *e = getenv("UNET_IGB_BAND"); return e ? atoi(e) : 1; }();
    if (use_band && igemmb3_applicable(p)) return pad ? launch_igemmb3<true>(p, st) : launch_igemmb3<false>(p, st);
    if (p.Nn % 128 == 0) return pad ? launch_cfgb<128, 128, true>(p, st) : launch_cfgb<128, 128, false>(p, st);
    return pad ? launch_cfgb<256, 64, true>(p, st) : launch_cfgb<256, 64, false>(p, st);
}

}  // namespace unet
