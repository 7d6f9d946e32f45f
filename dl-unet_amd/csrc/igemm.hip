// igemm.hip — fp32 implicit-GEMM on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32).
//
// One kernel family serves every dense contraction on the path:
//   conv3x3 fwd (+bias+ReLU, virtual zero-pad/crop-concat of two sources)   network.py:131-188
//   conv3x3 dgrad (full correlation with the flipped filter, + ReLU' mask / + skip gradient)
//   up-conv 2x2 s2 fwd (GEMM + scatter store) and dgrad (4 taps, stride 2)   network.py:159-183
//
// Why MFMA for all of them: in fp32 every 3x3 layer with Cin >= 64 has 143-730 FLOP/B of
// algorithmic intensity against a ridge of ~20 FLOP/B, i.e. they are bound by the fp32 FMA rate;
// the f32 MFMA has the VALU's peak rate (64 FLOP/clk/SIMD) with exact fmaf-chain numerics, needs
// one VGPR per operand and leaves the VALU free for addressing.
//
// Tile: BM x BN outputs per 256-thread workgroup (4 waves, each a 64x64 tile = 2x2 MFMA tiles of
// 32x32), K step 32.  Operands are staged global -> LDS with global_load_lds_dwordx4 (no VGPR
// round trip), double buffered, one barrier per K step.  LDS rows are 128 B (32 floats of K); the
// 16-byte chunk index is XOR-swizzled with (row>>1)&7 on the SOURCE address and on the fragment
// read, which makes the ds_read_b128 fragment reads bank-conflict free.
//
#include "common.hpp"
#include "igemm_epilogue.hpp"
#include <cstdio>
#include <cstdlib>

namespace unet {

int launch_igemmx(const IgemmP &p, bool pad, hipStream_t st);
int launch_igemmb(IgemmP p, bool pad, hipStream_t st);
int launch_wino(const IgemmP &p, const float *U, hipStream_t st);

// 0 = fp32 MFMA, direct (every product of the correlation: an exact fmaf chain), 1 = bf16x3 split (fp32-class accuracy on
// the bf16 matrix cores), 2 = bf16 compute, 3 (default) = fp32 MFMA with Winograd F(2x2,3x3) for the stride-1 3x3 layers'
// forward and dgrad (2.25x fewer multiplies, same parity tolerances; everything else as mode 0)
static int g_math_mode = [] { const char *e = getenv("UNET_MATH"); return e ? atoi(e) : 3; }();
int get_math_mode() { return g_math_mode; }
void set_math_mode(int m) { g_math_mode = m; }
static int g_lds_dma = [] { const char *e = getenv("UNET_LDS_DMA"); return e ? atoi(e) : 1; }();
int get_lds_dma_mode() { return g_lds_dma; }
void set_lds_dma_mode(int m) { g_lds_dma = m; }

#define GLDS16(gptr, lptr)                                                                    \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gptr),  \
                                     (__attribute__((address_space(3))) void *)(lptr), 16, 0, 0)

// 16-byte LDS-DMA through a buffer descriptor (a __device__ helper: calling the builtin directly from the kernel template
// made the host pass drop the kernel's stub without a diagnostic)
__device__ __forceinline__ void buf_lds16(__amdgpu_buffer_rsrc_t r, unsigned char *lds, int voff)
{
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void *)lds, 16, voff, 0, 0, 0);
}

template <int BM, int BN, bool PAD, bool BUF>
__global__ __launch_bounds__(256, 2) void igemm_f32_kernel(const IgemmP p)
{
    constexpr int WN = BN / 64;                 // waves along N
    constexpr int WM = 4 / WN;                  // waves along M
    static_assert(WM * 64 == BM, "4 waves of 64x64");
    constexpr int RA = BM / 32, RB = BN / 32;   // staging rows per thread
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    // kernel arguments of the K loop and the epilogue, copied up front and pinned in SGPRs (igemm_epilogue.hpp)
    const IgEp ep = igb_epilogue_args(p);
    int a_TX = p.TX, a_T = p.T, a_nsrc = p.nsrc, a_oy0 = p.oy0, a_ox0 = p.ox0, a_stride = p.stride, a_ldw = p.ldw;
    IGB_PIN(a_TX); IGB_PIN(a_T); IGB_PIN(a_nsrc); IGB_PIN(a_oy0); IGB_PIN(a_ox0); IGB_PIN(a_stride); IGB_PIN(a_ldw);

    // XCD-aware tile order: blocks b and b+8 share an XCD (and its L2), so give every XCD a
    // contiguous run of logical tiles; N-tiles of one M-tile are neighbours and reuse the A rows.
    int logical;
    {
        const int nblk = gridDim.x, q = nblk >> 3, r = nblk & 7, xcd = blockIdx.x & 7;
        logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (blockIdx.x >> 3);
    }
    const int mt = logical / p.ntiles, nt = logical - mt * p.ntiles;
    const int m0 = mt * BM, n0 = nt * BN;

    // ---- staging geometry: thread -> (row within a 32-row pass, 16-B chunk position)
    const int srow = tid >> 3;
    const int schunk = (tid & 7) ^ ((srow >> 1) & 7);   // swizzled SOURCE chunk for this LDS slot
    const int coff = schunk * 4;

    int a_off[RA], a_iy[RA], a_ix[RA];
    int b_off[RB];
#pragma unroll
    for (int j = 0; j < RB; ++j) {
        int n = n0 + srow + 32 * j;
        n = n < ep.Nn ? n : ep.Nn - 1;
        b_off[j] = n * a_ldw + coff;
    }

    int s = 0, ty = 0, tx = 0, kc = 0, kglob = 0;
    const float *sp = nullptr;
    int sH = 0, sW = 0, sC = 0, snch = 0, toff = 0;

    // BUF: LDS-DMA through buffer descriptors (buffer_load_dwordx4 ... offen lds): 32-bit byte offsets, the weight column in
    // the scalar offset, taps in the zero padding as offsets beyond num_records (the range check returns zeros).
    __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void *)p.src[0].p, 0, BUF ? p.buf_bytes[0] : 0, 0x00020000);
    __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc((void *)p.wt, 0, BUF ? p.buf_bytes[2] : 0, 0x00020000);
    auto setup_source = [&](int si) {
        const GSrc &g = p.src[si];
        sp = g.p; sH = g.H; sW = g.W; sC = g.C; snch = g.nch;
        if (BUF) rs_a = __builtin_amdgcn_make_buffer_rsrc((void *)g.p, 0, p.buf_bytes[si], 0x00020000);
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
            a_off[i] = ((img * g.H + iy) * g.W + ix) * g.C + g.c0 + coff;
        }
        toff = 0;
    };

    auto stage = [&](int buf) {
        unsigned char *abase = smem + buf * STAGE + wave * (8 * 128);
        unsigned char *bbase = abase + A_BYTES;
        if (BUF) {
#pragma unroll
            for (int i = 0; i < RA; ++i) {
                int vo = (a_off[i] + toff + kc) * 4;
                if (PAD) {
                    const bool inb = (unsigned)(a_iy[i] + ty) < (unsigned)sH && (unsigned)(a_ix[i] + tx) < (unsigned)sW;
                    vo = inb ? vo : (int)0x80000000;
                }
                buf_lds16(rs_a, abase + i * (32 * 128), vo);
            }
#pragma unroll
            for (int j = 0; j < RB; ++j)
                buf_lds16(rs_b, bbase + j * (32 * 128), (b_off[j] + kglob) * 4);
            return;
        }
#pragma unroll
        for (int i = 0; i < RA; ++i) {
            const float *g = sp + (a_off[i] + toff + kc);
            if (PAD) {
                const bool inb = (unsigned)(a_iy[i] + ty) < (unsigned)sH &&
                                 (unsigned)(a_ix[i] + tx) < (unsigned)sW;
                g = inb ? g : p.zeros + coff;
            }
            GLDS16(g, abase + i * (32 * 128));
        }
#pragma unroll
        for (int j = 0; j < RB; ++j) GLDS16(p.wt + (b_off[j] + kglob), bbase + j * (32 * 128));
    };

    auto advance = [&]() {
        kglob += 32;
        kc += 32;
        if (kc == snch) {
            kc = 0;
            ++tx;
            if (tx == a_TX) { tx = 0; ++ty; }
            if (ty * a_TX + tx == a_T) {
                ty = 0; tx = 0;
                ++s;
                if (s < a_nsrc) setup_source(s);
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
    const int swz = (l31 >> 1) & 7;
    const int a_rd = (wm * 64 + l31) * 128;
    const int b_rd = A_BYTES + (wn * 64 + l31) * 128;

    const int nk = p.Kd >> 5;
    setup_source(0);
    stage(0);
    advance();
    __syncthreads();            // drains the LDS-DMA (vmcnt(0)) and publishes buffer 0

    for (int ks = 0; ks < nk; ++ks) {
        const int cur = ks & 1;
        if (ks + 1 < nk) { stage(cur ^ 1); advance(); }
        const unsigned char *sb = smem + cur * STAGE;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int pos = ((2 * g + lh) ^ swz) * 16;
            const f32x4 a0 = *(const f32x4 *)(sb + a_rd + pos);
            const f32x4 a1 = *(const f32x4 *)(sb + a_rd + 32 * 128 + pos);
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
        __syncthreads();        // next buffer landed (vmcnt(0)) and this one is free to overwrite
    }

    igemm_epilogue<BM, BN>(ep, acc, m0, n0, tid, smem);
}

// Algorithmic FLOPs of one launch: 2 * (in-bounds (pixel, tap) pairs) * channels * N.  Taps that fall
// into the virtual zero padding are not counted (they are work the tiling does, not work the op needs).
double igemm_alg_flops(const IgemmP &p)
{
    double total = 0.0;
    const int TYn = p.T / p.TX;
    for (int s = 0; s < p.nsrc; ++s) {
        long cy = 0, cx = 0;
        for (int o = 0; o < p.OH; ++o)
            for (int t = 0; t < TYn; ++t) { const int i = (o + p.oy0) * p.stride - p.src[s].pad + t; cy += (i >= 0 && i < p.src[s].H); }
        for (int o = 0; o < p.OW; ++o)
            for (int t = 0; t < p.TX; ++t) { const int i = (o + p.ox0) * p.stride - p.src[s].pad + t; cx += (i >= 0 && i < p.src[s].W); }
        total += 2.0 * p.NB * (double)cy * (double)cx * p.src[s].nch * p.Nn;
    }
    return total;
}

// Algorithmic HBM bytes of one launch: the source pixels its taps can reach, once; the output, once (plus
// what the fused epilogue reads: add / mask); the filter matrix, once.
double igemm_alg_bytes(const IgemmP &p)
{
    double b = 0.0;
    const int TYn = p.T / p.TX;
    for (int s = 0; s < p.nsrc; ++s) {
        const GSrc &g = p.src[s];
        int y0 = p.oy0 * p.stride - g.pad, y1 = (p.OH - 1 + p.oy0) * p.stride - g.pad + TYn;
        int x0 = p.ox0 * p.stride - g.pad, x1 = (p.OW - 1 + p.ox0) * p.stride - g.pad + p.TX;
        y0 = y0 < 0 ? 0 : y0; x0 = x0 < 0 ? 0 : x0;
        y1 = y1 > g.H ? g.H : y1; x1 = x1 > g.W ? g.W : x1;
        if (y1 > y0 && x1 > x0) b += (double)p.NB * (y1 - y0) * (x1 - x0) * g.nch * 4.0;
    }
    const double out = (double)p.M * p.Nn * 4.0;
    b += out * (1.0 + (p.add ? 1.0 : 0.0) + (p.mask ? 1.0 : 0.0));
    if (p.pool_dst) b += out / 4.0;
    b += (double)p.Nn * p.Kd * 4.0;
    return b;
}

template <int BM, int BN, bool PAD, bool BUF>
static int launch_cfg(const IgemmP &p, hipStream_t st)
{
    constexpr int LDS = 2 * (BM + BN) * 128;
    static bool attr_done[64] = {false};
    auto kern = igemm_f32_kernel<BM, BN, PAD, BUF>;
    if (int rc_ = ensure_dynamic_lds((const void *)kern, LDS, attr_done)) return rc_;
    IgemmP q = p;
    q.mtiles = cdiv(p.M, BM);
    q.ntiles = cdiv(p.Nn, BN);
    char tag[96];
    snprintf(tag, sizeof(tag), "igemm<%d;%d;%d> M=%d N=%d Kd=%d T=%d s=%d nsrc=%d", BM, BN, (int)PAD, p.M, p.Nn, p.Kd, p.T, p.stride, p.nsrc);
    prof_begin(PK_IGEMM, tag, st, igemm_alg_flops(p), 2.0 * q.mtiles * BM * (double)q.ntiles * BN * p.Kd, igemm_alg_bytes(p));
    hipLaunchKernelGGL(kern, dim3(q.mtiles * q.ntiles), dim3(256), LDS, st, q);
    prof_end(st);
    HIP_TRY(hipGetLastError());
    return 0;
}

int launch_igemm(IgemmP p, hipStream_t st)
{
    ARG_CHECK(p.nsrc == 1 || p.nsrc == 2, "igemm: nsrc must be 1 or 2");
    int kd = 0;
    bool pad = false;
    for (int i = 0; i < p.nsrc; ++i) {
        ARG_CHECK(p.src[i].nch > 0 && p.src[i].nch % 32 == 0, "igemm: channel count %d is not a multiple of 32", p.src[i].nch);
        ARG_CHECK(p.src[i].C % 4 == 0 && p.src[i].c0 % 4 == 0, "igemm: channel pitch/offset must be multiples of 4");
        kd += p.src[i].nch * p.T;
        // a source is "padded" if any tap can fall outside it
        const int lo = p.oy0 * p.stride - p.src[i].pad;
        const int hi_y = (p.OH - 1 + p.oy0) * p.stride - p.src[i].pad + (p.T / p.TX - 1);
        const int hi_x = (p.OW - 1 + p.ox0) * p.stride - p.src[i].pad + (p.TX - 1);
        const int lo_x = p.ox0 * p.stride - p.src[i].pad;
        if (lo < 0 || lo_x < 0 || hi_y >= p.src[i].H || hi_x >= p.src[i].W) pad = true;
    }
    ARG_CHECK(kd == p.Kd, "igemm: Kd %d does not match sources (%d)", p.Kd, kd);
    ARG_CHECK(p.M == p.NB * p.OH * p.OW && p.M > 0 && p.Nn > 0, "igemm: bad M/N");
    if (!p.scatter) ARG_CHECK(p.DH == p.OH && p.DW == p.OW, "igemm: linear store needs dst extent == output domain");
    if (p.scatter == 2) ARG_CHECK(p.dwy0 >= 0 && p.dwx0 >= 0 && p.OH + p.dwy0 <= p.DH && p.OW + p.dwx0 <= p.DW, "igemm: window store outside dst");
    if (p.ldw == 0) p.ldw = p.Kd;
    ARG_CHECK(p.ldw >= p.Kd && p.ldw % 4 == 0, "igemm: bad weight pitch");
    ARG_CHECK(p.DC % 4 == 0 && p.dn0 % 4 == 0 && p.Nn % 4 == 0 && (!p.scatter || p.cout % 4 == 0), "igemm: 16-byte stores need channel counts that are multiples of 4");
    ARG_CHECK((size_t)p.NB * p.DH * p.DW * p.DC < 0xFFFFFFFFull, "igemm: destination exceeds 32-bit element offsets");
    for (int i = 0; i < p.nsrc; ++i)
        ARG_CHECK((size_t)p.NB * p.src[i].H * p.src[i].W * p.src[i].C < 0x7FFFFFFFull, "igemm: source exceeds 31-bit element offsets");
    ARG_CHECK(!p.pool_dst || (p.math == 3 && p.wino_u), "igemm: the fused max-pool exists only in the Winograd epilogue");
    p.zeros = zero_page();
    if (!p.zeros) return -2;
    p.d_ohw = make_fastdiv((unsigned)(p.OH * p.OW));
    p.d_ow = make_fastdiv((unsigned)p.OW);
    ARG_CHECK(p.math >= 0 && p.math <= 3, "igemm: bad arithmetic mode %d", p.math);
    if (p.math == 3 && p.wino_u && wino_applicable(p)) {
        ARG_CHECK((size_t)p.NB * p.DH * p.DW * p.DC < 0x7FFFFFFFull, "igemm: destination exceeds 31-bit element offsets");
        return launch_wino(p, p.wino_u, st);
    }
    if (p.math == 1) return launch_igemmx(p, pad, st);
    if (p.math == 2) return launch_igemmb(p, pad, st);          // bf16 tensors (igemmb.hip)
    ARG_CHECK((size_t)p.NB * p.DH * p.DW * p.DC < 0x7FFFFFFFull, "igemm: destination exceeds 31-bit element offsets");
    // buffer-descriptor LDS-DMA needs sources and weights below 2 GiB
    bool buf = g_lds_dma != 0;
    for (int i = 0; i < 3; ++i) p.buf_bytes[i] = 0;
    for (int i = 0; i < p.nsrc; ++i) {
        const size_t b = (size_t)p.NB * p.src[i].H * p.src[i].W * p.src[i].C * sizeof(float);
        if (b >= 0x7FFFFFFFull) buf = false;
        p.buf_bytes[i] = (int)b;
    }
    {
        const size_t b = (size_t)p.Nn * p.ldw * sizeof(float);
        if (b >= 0x7FFFFFFFull) buf = false;
        p.buf_bytes[2] = (int)b;
    }
    if (buf) {
        if (p.Nn % 128 == 0) return pad ? launch_cfg<128, 128, true, true>(p, st) : launch_cfg<128, 128, false, true>(p, st);
        return pad ? launch_cfg<256, 64, true, true>(p, st) : launch_cfg<256, 64, false, true>(p, st);
    }
    if (p.Nn % 128 == 0) return pad ? launch_cfg<128, 128, true, false>(p, st) : launch_cfg<128, 128, false, false>(p, st);
    return pad ? launch_cfg<256, 64, true, false>(p, st) : launch_cfg<256, 64, false, false>(p, st);
}

}  // namespace unet
