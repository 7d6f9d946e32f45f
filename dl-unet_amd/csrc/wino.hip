// wino.hip — fp32 Winograd F(2x2,3x3) for the stride-1 3x3 convolutions of the path (forward and dgrad;
// network.py:131-188 and their autograd), on the gfx950 matrix cores.
//
//   Y = A^T [ (G g G^T) (.) (B^T d B) ] A            per 2x2 output tile, 4x4 input patch d, 3x3 filter g
//   M[xi][tile][n] = sum_c V[xi][tile][c] * U[xi][c][n]       xi = 0..15: 16 small GEMMs, 16 instead of 36
//                                                              multiplies per (tile, c, n) -> 2.25x fewer MFMA flops
// All arithmetic is fp32 (v_mfma_f32_16x16x4_f32 + fp32 adds); only the evaluation order differs from the
// direct correlation, which costs ~1e-6 relative (tests/test_ops_gpu.py pins it against the oracle).
//
// Work decomposition — a 256-thread workgroup owns 64 Winograd tiles x 32 output channels, two workgroups per CU (their
// set-up, first-stage latency, epilogue and barrier stalls hide behind each other's MFMAs); a wave owns 16 tiles x 32
// channels for all 16 xi -> 16*2 accumulators of 16x16 (128 VGPRs):
//   * tiles are numbered linearly over (image, tile row, tile column): no 2-D edge waste, only the last workgroup
//     of a launch is ragged; consecutive workgroups are the n-tiles of one m-tile (they are resident together on an XCD
//     and re-read one patch image from its L2: HBM traffic of the family -34 % against n-major order, step -0.7 %);
//   * per K step (8 channels) the workgroup stages by LDS-DMA (buffer descriptors: fixed per-lane offsets, the channel
//     step in the scalar offset, pixels outside the tensor through the range check)  (a) an 18 KiB patch image — each
//     tile's two own pixel columns; columns 2,3 are the right neighbour's, or a per-tile-row "tail" — and (b) the
//     16 KiB block of its channels' transformed filters U, which wino_transform_ref_kernel wrote in LDS/fragment
//     order; 2-deep rings, one barrier per step;
//   * the INPUT transform is done in registers: lane (tile, kg) reads its 16 pixels x 2 channels (ds_read_b64,
//     conflict-free layout), 32 packed adds give V[xi] for those 2 channels = exactly the A operands of the 16x16x4
//     MFMAs (k index = kg); V never exists in memory;
//   * the OUTPUT transform is done in registers too: the accumulator layout gives a lane (channel, 4 tiles) for
//     every xi; 24 adds per tile yield the 2x2 outputs, which go through LDS to 128-B-contiguous float4 stores with
//     the usual fused epilogue (bias, +add, ReLU / deferred-ReLU window, ReLU' mask) and, for the layers in front of
//     a max-pool, the pooled tensor as well (a tile is one pooling window).
//
// Why this shape: accumulators are 16x the output tile, so the tile is small and the staged bytes per MFMA cycle are
// what a step has to hide: 34 KiB per 64 MFMAs per wave.  DESIGN.md section 4 has the history (a 512-thread, 64-channel
// wide, one-per-CU variant with 3-deep rings was measured slower or equal on every launch of the net and removed).
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
    int xbytes[2], ubytes;   // buffer-descriptor sizes (bytes) of the sources and of U; 0 if any exceeds 2 GiB
    int gn, gn_shift;        // tile order: n-tiles are walked in groups of gn (a power of two dividing ntiles), n fastest inside a group
};


// --------------------------------------------------------------------------------------------------------
// U = G g G^T for every (n, c), straight from the reference's OIHW weights, written in the order the main kernel
// stages and reads it:
//   block (nt = n/64, step = c/8)  [32 KiB]:  piece (n16 = (n/16)%4, xg = xi/4, h = c%2) [1 KiB]:
//   lane (kg = (c%8)/2, nl = n%16) [16 B]: xi%4
//   forward  (dgrad = 0): filter matrix row n = output channel n0 + n, column k = input channel k0 + k, taps as stored
//   dgrad    (dgrad = 1): row n = INPUT channel n0 + n, column k = output channel k0 + k, taps flipped (index 8 - t)
// Thread order keeps the 36-byte reads of neighbouring threads adjacent (k fastest forward, n fastest dgrad).
// --------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void wino_transform_ref_kernel(const float *__restrict__ w, int I, int dgrad, int n0, int Nn, int k0, int Kc,
                                                                 float *__restrict__ U)
{
    const size_t total = (size_t)Nn * Kc;
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    int n, c;
    if (dgrad) { c = (int)(idx / Nn); n = (int)(idx - (size_t)c * Nn); }
    else       { n = (int)(idx / Kc); c = (int)(idx - (size_t)n * Kc); }
    const int o = dgrad ? k0 + c : n0 + n, i = dgrad ? n0 + n : k0 + c;
    const float *src = w + ((size_t)o * I + i) * 9;
    float g[3][3];
#pragma unroll
    for (int t = 0; t < 9; ++t) g[t / 3][t % 3] = src[dgrad ? 8 - t : t];
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
    for (int a = 0; a < 4; ++a) {
        u[4 * a + 0] = r[a][0];
        u[4 * a + 1] = 0.5f * (r[a][0] + r[a][1] + r[a][2]);
        u[4 * a + 2] = 0.5f * (r[a][0] - r[a][1] + r[a][2]);
        u[4 * a + 3] = r[a][2];
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

int wino_transform_ref(const float *w_oihw, int I, int dgrad, int n0, int Nn, int k0, int Kc, float *U, hipStream_t st)
{
    ARG_CHECK(w_oihw && U && Nn % 32 == 0 && Kc % 8 == 0 && Kc > 0, "wino_transform_ref: bad shape");      // (U is allocated in whole 64-row blocks: wino_u_floats)
    const size_t total = (size_t)Nn * Kc;
    hipLaunchKernelGGL(wino_transform_ref_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, w_oihw, I, dgrad, n0, Nn, k0, Kc, U);
    HIP_TRY(hipGetLastError());
    return 0;
}

// --------------------------------------------------------------------------------------------------------
// LDS of a workgroup: [2 x patch stage 18 KiB][2 x U block 16 KiB] = 68 KiB (+ row tables), two workgroups per CU.
//   patch stage  [qy 4][par 2][idx 72][slot 2][16 B]:  idx < 64: pixel (qy, qx = par) of tile idx; idx = 64 + j: pixel
//                (qy, qx = 2 + par) of "tail" j; slot = half ^ ((idx>>3)&1) holds channels 4*half..+3 of the step.
//                The two halves of a pixel are adjacent lanes of one LDS-DMA instruction (one 32-B access instead of
//                two cache-line lookups); the slot swizzle keeps the ds_read_b64 of 16 tiles on 64 distinct banks.
//   U block      [n16 2][xg 4][h 2][lane 64][16 B]
// A tile's columns 2,3 are its right neighbour's columns 0,1 — tiles are numbered linearly over (image, tile row,
// tile column), so the neighbour is tile+1 except at the end of a tile row (or of the workgroup's 64 tiles), where
// the two pixels come from tail j = (tile row) - (tile row of the workgroup's first tile).  That halves the patch
// bytes against private 4x4 patches and keeps the numbering free of 2-D edge waste.  The stage is exactly 18 LDS-DMA
// instructions (1 KiB each, lane-linear); a lane derives (row, idx) from its byte position.
constexpr int WINO_ROW = 72 * 32;                       // 2304 B per (qy, par)
constexpr int WINO_PATCH = 8 * WINO_ROW;                // 18432

template <int N> __device__ __forceinline__ void wino_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// plain float2 arithmetic: the backend selects v_pk_add_f32 (inline asm here hid the VALU->MFMA hazards from the compiler)
__device__ __forceinline__ f32x2 pk_add(f32x2 a, f32x2 b) { return a + b; }
__device__ __forceinline__ f32x2 pk_sub(f32x2 a, f32x2 b) { return a - b; }

// --------------------------------------------------------------------------------------------------------
constexpr int W32_UBASE = 2 * WINO_PATCH;                 // 36864
constexpr int W32_LDS = W32_UBASE + 2 * 16384;            // 69632
constexpr int W32_TOTAL = W32_LDS + 1280 + 5 * 256 * 4;   // + row tables + parked second-source offsets = 76032

template <bool BUF>
__global__ __launch_bounds__(256, 2) void wino32_f32_kernel(const WinoP k)
{
    const IgemmP &p = k.p;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // = wm: tiles 16*wave .. +15, all 32 channels
    const int l15 = lane & 15, kg = lane >> 4;
    const int tpi = k.tiles_x * k.tiles_y;
    const int ns = k.nsteps;
    // the epilogue's kernel arguments, fetched now in one batch and pinned in SGPRs: fetched where they are used, each costs the
    // epilogue a scalar-load round trip behind a branch (fourteen of them in a row)
    int e_scatter = p.scatter, e_DH = p.DH, e_DW = p.DW, e_DC = p.DC, e_dwy0 = p.dwy0, e_dwx0 = p.dwx0, e_OH = p.OH, e_OW = p.OW;
    int e_rw0 = p.rw0, e_rw1 = p.rw1, e_dn0 = p.dn0, e_cout = p.cout, e_relu = p.relu, e_Nn = p.Nn, e_MT = k.MT;
    const float *e_add = p.add, *e_mask = p.mask, *e_bias = p.bias;
    float *e_dst = p.dst, *e_pool = p.pool_dst;
    asm volatile("" : "+s"(e_scatter), "+s"(e_DH), "+s"(e_DW), "+s"(e_DC), "+s"(e_dwy0), "+s"(e_dwx0), "+s"(e_OH), "+s"(e_OW));
    asm volatile("" : "+s"(e_rw0), "+s"(e_rw1), "+s"(e_dn0), "+s"(e_cout), "+s"(e_relu), "+s"(e_Nn), "+s"(e_MT));
    asm volatile("" : "+s"(e_add), "+s"(e_mask), "+s"(e_bias), "+s"(e_dst), "+s"(e_pool));
    // (behind the asm the compiler no longer knows these are global pointers and would use FLAT instructions: say so)
    typedef const float __attribute__((address_space(1))) *gcfp;
    typedef float __attribute__((address_space(1))) *gfp;
    const gcfp g_add = (gcfp)e_add, g_mask = (gcfp)e_mask, g_bias = (gcfp)e_bias;
    const gfp g_dst = (gfp)e_dst, g_pool = (gfp)e_pool;
    // set-up and epilogue are chains of vector instructions that the partner workgroup's MFMA stream starves (~20 cycles per
    // instruction): they run at raised priority, the K loop at the default (-0.3 % per step)
    __builtin_amdgcn_s_setprio(3);

    int slot;
    {
        const int G = gridDim.x, q = G >> 3, r = G & 7, xcd = blockIdx.x & 7;
        slot = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (blockIdx.x >> 3);
    }
    // consecutive slots (= workgroups resident together on an XCD, sharing its L2): gn n-tiles of one m-tile, then the next
    // m-tile; after all m-tiles the next group of n-tiles.  gn = 1 is n-major (the U block shared, every patch image distinct)
    int nt, mt;
    {
        const int per = p.mtiles << k.gn_shift;
        const int ng = slot / per, rem = slot - ng * per;
        mt = rem >> k.gn_shift;
        nt = (ng << k.gn_shift) + (rem & (k.gn - 1));
    }
    const int T0 = mt * 64, n0 = nt * 32;

    // U: the 32 channels are half (nt & 1) of the 64-channel block nt >> 1
    const float *ublk = k.U + (size_t)(nt >> 1) * ns * 8192 + (nt & 1) * 4096 + (4 * wave) * 256 + lane * 4;
    // BUF: LDS-DMA through buffer descriptors (buffer_load_dwordx4 ... offen lds): the per-lane byte offset is fixed for the
    // whole tile, the channel step goes into the scalar offset, and pixels outside the tensor are offsets beyond
    // num_records (the range check returns zeros) - no per-step address VALU at all.  Needs tensors < 2 GiB.
    __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void *)p.src[0].p, 0, BUF ? k.xbytes[0] : 0, 0x00020000);
    __amdgpu_buffer_rsrc_t rs_u = __builtin_amdgcn_make_buffer_rsrc((void *)k.U, 0, BUF ? k.ubytes : 0, 0x00020000);
    const int u_voff = ((4 * wave) * 256 + lane * 4) * 4;
    const int u_soff0 = ((nt >> 1) * ns * 8192 + (nt & 1) * 4096) * 4;
    int uissued = 0;
    auto issue_u = [&]() {
        const int buf = uissued & 1;
        unsigned char *ub = smem + W32_UBASE + buf * 16384 + (4 * wave) * 1024;
        if (BUF) {
            const int us = u_soff0 + uissued * 32768;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_u, (__attribute__((address_space(3))) void *)(ub + i * 1024), 16, u_voff, us + i * 1024, 0, 0);
        } else {
            const float *us = ublk + (size_t)uissued * 8192;
#pragma unroll
            for (int i = 0; i < 4; ++i) GLDS16(us + i * 256, ub + i * 1024);
        }
        ++uissued;
    };
    // ---- DMA role: patch instructions i = wave + 4*ii (i < 18): 5 for waves 0,1, 4 for waves 2,3; U pieces 4*wave .. +3
    // The per-lane source offsets are worked out branch-free from kernel arguments read up front in one batch: written with
    // the conditions as branches, the compiler fetches each argument under the condition that needs it — thirty dependent
    // scalar-load round trips, 4-7k cycles per workgroup (in-kernel stamps).
    constexpr int NPI = 5;
    const int npi = wave < 2 ? 5 : 4;
    int poff[NPI];
    int *poff1 = (int *)(smem + W32_LDS + 1280) + tid;
    // tile T0 (uniform: scalar unit) -> image, tile row, tile column; a lane's tile is at most 63 tiles, 7 tile rows and (at least 7
    // tile rows per image: wino_applicable) one image further.  x / tiles_x by an exact float floor (values below 2^12).
    const int tiles_x = k.tiles_x;
    const int img0 = fdiv(T0, k.d_tpi);
    const int trem0 = T0 - img0 * tpi;
    const int ty0 = fdiv(trem0, k.d_tx);
    const int tx0 = trem0 - ty0 * tiles_x;
    const float rtx = 1.0f / (float)tiles_x;
    {
        const int nsrc = p.nsrc, oy0 = p.oy0, ox0 = p.ox0, MT = k.MT;
        const int img0_ = img0;
        int sH[2], sW[2], sC[2], sc0[2], spad[2];
#pragma unroll
        for (int si = 0; si < 2; ++si) { sH[si] = p.src[si].H; sW[si] = p.src[si].W; sC[si] = p.src[si].C; sc0[si] = p.src[si].c0; spad[si] = p.src[si].pad; }
        // offset = image base (one of two scalars) + row * row pitch + column * channels: 24-bit multiplies (full rate; rows,
        // columns and channels are below 2^12, a row pitch below 2^24)
        int rowp[2], imgb[2][2];
#pragma unroll
        for (int si = 0; si < 2; ++si) { rowp[si] = sW[si] * sC[si]; imgb[si][0] = img0_ * sH[si] * rowp[si] + sc0[si]; imgb[si][1] = imgb[si][0] + sH[si] * rowp[si]; }
        const int lastrel = MT - 1 - T0;
#pragma unroll
        for (int ii = 0; ii < NPI; ++ii) {
            // byte position in the stage, in 16-byte units: u = row * 144 + idx * 2 + slot  (exact float floor: u < 1152)
            const int u = (wave + 4 * ii) * 64 + lane;
            const int row = (int)(((float)u + 0.5f) * (1.0f / 144.0f));
            const int rem = u - row * 144;
            const int idx = rem >> 1;
            const int qy = row >> 1, par = row & 1, half = (rem & 1) ^ ((idx >> 3) & 1);
            // tile index relative to T0: own tiles 0..63; tail j: the last tile of tile row R0 + j, or tile 63
            const bool tail = idx >= 64;
            const int j = idx - 64;
            const int erow = j * tiles_x - tx0;                       // first tile of row R0 + j, relative
            int e = tail ? erow + tiles_x - 1 : idx;
            e = e < 63 ? e : 63;
            e = e < lastrel ? e : lastrel;
            const bool ok = !tail | (erow <= 63);
            const int qx = tail ? 2 + par : par;
            const int t = tx0 + e;
            const int r = (int)(((float)t + 0.5f) * rtx);
            const int tx = t - r * tiles_x;
            int ty = ty0 + r;
            const bool wrap = ty >= k.tiles_y;
            ty = wrap ? ty - k.tiles_y : ty;
#pragma unroll
            for (int si = 0; si < 2; ++si) {
                const int iy = 2 * ty + oy0 - spad[si] + qy, ix = 2 * tx + ox0 - spad[si] + qx;
                const bool inb = ok & ((unsigned)iy < (unsigned)sH[si]) & ((unsigned)ix < (unsigned)sW[si]) & (si < nsrc) & (ii < npi);
                int o = (wrap ? imgb[si][1] : imgb[si][0]) + __mul24(iy, rowp[si]) + __mul24(ix, sC[si]) + 4 * half;
                o = inb ? o : -1;
                if (si == 0) poff[ii] = o; else if (nsrc > 1) poff1[ii * 256] = o;
            }
        }
    }
    const float *sp = p.src[0].p;
    int snch = p.src[0].nch, kc = 0, pissued = 0;
    if (BUF) {
#pragma unroll
        for (int ii = 0; ii < NPI; ++ii) poff[ii] = poff[ii] >= 0 ? poff[ii] * 4 : (int)0x80000000;
    }
    const int tl = wave * 16 + l15;
    const int offA = tl * 32 + (((kg >> 1) ^ ((tl >> 3) & 1)) * 16) + (kg & 1) * 8;
    int offB;
    {
        const int t = tx0 + tl;
        const int r = (int)(((float)t + 0.5f) * rtx);
        const bool rowend = (t - r * tiles_x == tiles_x - 1) || tl == 63;
        int j = r;                                                         // tile row of the lane's tile - tile row of T0
        j = j < 7 ? j : 7;
        const int idxB = rowend ? 64 + j : tl + 1;
        offB = idxB * 32 + (((kg >> 1) ^ ((idxB >> 3) & 1)) * 16) + (kg & 1) * 8;
    }
    const int b_rd = W32_UBASE + lane * 16;

    // one stage = this wave's patch instructions + its 4 U pieces (9 or 8 LDS-DMA instructions)
    auto issue_patch = [&]() {
        const int buf = pissued & 1;
        unsigned char *sb = smem + buf * WINO_PATCH + wave * 1024;
        if (BUF) {
#pragma unroll
            for (int ii = 0; ii < 4; ++ii)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (__attribute__((address_space(3))) void *)(sb + ii * 4096), 16, poff[ii], kc * 4, 0, 0);
            if (wave < 2)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (__attribute__((address_space(3))) void *)(sb + 16384), 16, poff[4], kc * 4, 0, 0);
        } else {
#pragma unroll
            for (int ii = 0; ii < 4; ++ii) {
                const float *g = poff[ii] >= 0 ? sp + (poff[ii] + kc) : p.zeros;
                GLDS16(g, sb + ii * 4096);
            }
            if (wave < 2) {
                const float *g = poff[4] >= 0 ? sp + (poff[4] + kc) : p.zeros;
                GLDS16(g, sb + 16384);
            }
        }
        ++pissued;
        kc += 8;
        if (kc == snch && pissued < ns) {
            kc = 0;
            sp = p.src[1].p; snch = p.src[1].nch;
            if (BUF) rs_x = __builtin_amdgcn_make_buffer_rsrc((void *)p.src[1].p, 0, k.xbytes[1], 0x00020000);
#pragma unroll
            for (int ii = 0; ii < NPI; ++ii) { const int o = poff1[ii * 256]; poff[ii] = BUF ? (o >= 0 ? o * 4 : (int)0x80000000) : o; }
        }
    };
    auto issue_stage = [&]() { issue_patch(); issue_u(); };

    f32x4 acc[16][2];
#pragma unroll
    for (int x = 0; x < 16; ++x)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[x][j][r] = 0.f;
    f32x2 v0[16];
    f32x4 bf[2][2];

    auto load_v = [&](int buf) {
        const unsigned char *sa = smem + buf * WINO_PATCH + offA;
        const unsigned char *sb = smem + buf * WINO_PATCH + offB;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned char *src = (j < 2 ? sa : sb) + (j & 1) * WINO_ROW;
            const f32x2 a0 = *(const f32x2 *)(src);
            const f32x2 a1 = *(const f32x2 *)(src + 2 * WINO_ROW);
            const f32x2 a2 = *(const f32x2 *)(src + 4 * WINO_ROW);
            const f32x2 a3 = *(const f32x2 *)(src + 6 * WINO_ROW);
            v0[j] = pk_sub(a0, a2);
            v0[4 + j] = pk_add(a1, a2);
            v0[8 + j] = pk_sub(a2, a1);
            v0[12 + j] = pk_sub(a1, a3);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const f32x2 t0 = v0[4 * i], t1 = v0[4 * i + 1], t2 = v0[4 * i + 2], t3 = v0[4 * i + 3];
            v0[4 * i + 0] = pk_sub(t0, t2);
            v0[4 * i + 1] = pk_add(t1, t2);
            v0[4 * i + 2] = pk_sub(t2, t1);
            v0[4 * i + 3] = pk_sub(t1, t3);
        }
    };
    auto read_b = [&](int buf, int c) {
        const unsigned char *sb = smem + buf * 16384 + b_rd;
        bf[c & 1][0] = *(const f32x4 *)(sb + c * 1024);
        bf[c & 1][1] = *(const f32x4 *)(sb + 8192 + c * 1024);
    };

    // ---- pipeline (2-deep): stages s, s+1 in flight at most.  Step s: 64 MFMAs from U(s); wait stage s+1, barrier (every
    // wave is done with U(s) and, since the previous step, with patches(s)); issue stage s+2 into the freed buffers;
    // transform patches(s+1).
    issue_stage();
    issue_stage();                                    // ns >= 4
    if (wave < 2) wino_wait_vmcnt<9>(); else wino_wait_vmcnt<8>();
    __builtin_amdgcn_s_barrier();
    read_b(0, 0);
    load_v(0);
    __builtin_amdgcn_s_setprio(0);
    for (int s = 0; s < ns; ++s) {
        const int ub = s & 1;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const int xg = c >> 1, h = c & 1;
            if (c + 1 < 8) read_b(ub, c + 1);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc[4 * xg + e][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(v0[4 * xg + e][h], bf[c & 1][0][e], acc[4 * xg + e][0], 0, 0, 0);
                acc[4 * xg + e][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(v0[4 * xg + e][h], bf[c & 1][1][e], acc[4 * xg + e][1], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        wino_wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        if (s + 2 < ns) issue_stage();
        if (s + 1 < ns) {
            read_b((s + 1) & 1, 0);           // the first B fragments of the next step fly while its V is transformed
            load_v((s + 1) & 1);
        }
    }
    __builtin_amdgcn_s_setprio(3);
    // ---- epilogue: staging[256 rows = tile*4 + 2*py + px][32 n] in the U ring | rowoff[256] | flags[256]
    float *stg = (float *)(smem + W32_UBASE);
    unsigned *rowoff = (unsigned *)(smem + W32_LDS);
    unsigned char *rflag = smem + W32_LDS + 1024;
    {
        const int tl2 = tid >> 2, py = (tid >> 1) & 1, px = tid & 1;
        const int lastrel = e_MT - 1 - T0;
        const bool tok = tl2 <= lastrel;
        // tile T0 + tl2 -> (image, tile row, tile column) as in the set-up: float floor, one image wrap
        const int t = tx0 + (tok ? tl2 : lastrel);
        const int r = (int)(((float)t + 0.5f) * rtx);
        const int tx = t - r * tiles_x;
        int ty = ty0 + r;
        const bool wrap = ty >= k.tiles_y;
        ty = wrap ? ty - k.tiles_y : ty;
        int oy = 2 * ty + py, ox = 2 * tx + px;
        const bool ok = tok & (oy < e_OH) & (ox < e_OW);
        oy = oy < e_OH ? oy : e_OH - 1; ox = ox < e_OW ? ox : e_OW - 1;
        // destination pixel: window scatter (2) or dst pixel == output pixel; rows, columns, channels below 2^12: 24-bit multiplies
        const bool win = e_scatter == 2;
        const int PH = win ? e_DH : e_OH, PW = win ? e_DW : e_OW, py0 = win ? e_dwy0 : 0, px0 = win ? e_dwx0 : 0;
        const int rowp = PW * e_DC;
        const int ib = (img0 + (wrap ? 1 : 0)) * PH * rowp;                  // (two scalar values)
        const unsigned off = (unsigned)(ib + __mul24(oy + py0, rowp) + __mul24(ox + px0, e_DC));
        rowoff[tid] = off;
        const bool inwin = (e_rw1 > e_rw0) & (oy >= e_rw0) & (oy < e_rw1) & (ox >= e_rw0) & (ox < e_rw1);
        rflag[tid] = (ok ? 0 : 1) | (inwin ? 2 : 0);
    }
    __syncthreads();                 // K loop done in every wave (the staging aliases the U ring); row tables visible

    // destination offsets of this thread's 2 x 4 rows, and the +add / ReLU' mask operands: issued now, their latency runs
    // under the output transform
    const bool relu_win = e_rw1 > e_rw0;
    const int c4 = tid & 7;
    const int ncol = n0 + 4 * c4;
    size_t eo[2][4];
    unsigned char efl[2][4];
    f32x4 eadd[2][4], emask[2][4];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int row = (h * 32 + (tid >> 3)) * 4 + u;                 // the four pixels of one tile
            eo[h][u] = (size_t)rowoff[row] + (size_t)(e_dn0 + ncol);
            efl[h][u] = rflag[row];
        }
    if (e_add) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int u = 0; u < 4; ++u) eadd[h][u] = *(const __attribute__((address_space(1))) f32x4 *)(g_add + eo[h][u]);
    }
    if (e_mask) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int u = 0; u < 4; ++u) emask[h][u] = *(const __attribute__((address_space(1))) f32x4 *)(g_mask + eo[h][u]);
    }
#pragma unroll
    for (int nn = 0; nn < 2; ++nn) {
        const int n = nn * 16 + l15;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float s0[4], s1[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                s0[j] = acc[j][nn][r] + acc[4 + j][nn][r] + acc[8 + j][nn][r];
                s1[j] = acc[4 + j][nn][r] - acc[8 + j][nn][r] - acc[12 + j][nn][r];
            }
            // staging image, bank-conflict free on both sides: pixel row u of tile t sits at row u ^ (t & 1) of the tile's
            // four 128-B rows (neighbouring tiles of a ds_read_b128 lane group alternate bank-row halves) and channel n at
            // n ^ 16*((t >> 2) & 1) (the two k groups of a 32-lane store group hit different banks); t & 1 = r & 1,
            // (t >> 2) & 1 = kg & 1 here
            const int tile = wave * 16 + 4 * kg + r;
            float *o = stg + tile * 128 + (n ^ ((kg & 1) << 4));
            o[((0 ^ (r & 1))) * 32] = s0[0] + s0[1] + s0[2];
            o[((1 ^ (r & 1))) * 32] = s0[1] - s0[2] - s0[3];
            o[((2 ^ (r & 1))) * 32] = s1[0] + s1[1] + s1[2];
            o[((3 ^ (r & 1))) * 32] = s1[1] - s1[2] - s1[3];
        }
    }
    __syncthreads();
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (e_bias) {
#pragma unroll
        for (int c = 0; c < 4; ++c) { const int n = ncol + c; bv[c] = g_bias[e_cout ? n % e_cout : n]; }
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        f32x4 v[4];
        const int tloc = h * 32 + (tid >> 3);
#pragma unroll
        for (int u = 0; u < 4; ++u)
            v[u] = *(const f32x4 *)(stg + tloc * 128 + (u ^ (tloc & 1)) * 32 + ((4 * c4) ^ (((tloc >> 2) & 1) << 4))) + bv;
        if (e_add) {
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] += eadd[h][u];
        }
        if (e_relu) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const bool defer = relu_win && (efl[h][u] & 2);
#pragma unroll
                for (int c = 0; c < 4; ++c) v[u][c] = (v[u][c] > 0.f || defer) ? v[u][c] : 0.f;
            }
        }
        if (e_mask) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int c = 0; c < 4; ++c) v[u][c] = emask[h][u][c] > 0.f ? v[u][c] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (!(efl[h][u] & 1)) *(__attribute__((address_space(1))) f32x4 *)(g_dst + eo[h][u]) = v[u];
        if (e_pool) {
            // fused 2x2 max-pool (network.py:132-150): a Winograd tile IS one pooling window and the linear tile index
            // is the pooled pixel index (launch_wino checks even extents)
            const int T = T0 + tloc;
            if (T < e_MT) {
                f32x4 m;
#pragma unroll
                for (int c = 0; c < 4; ++c) m[c] = fmaxf(fmaxf(v[0][c], v[1][c]), fmaxf(v[2][c], v[3][c]));
                *(__attribute__((address_space(1))) f32x4 *)(g_pool + (size_t)T * e_Nn + ncol) = m;
            }
        }
    }
}


bool wino_applicable(const IgemmP &p)
{
    if (p.T != 9 || p.TX != 3 || p.stride != 1 || p.scatter == 1) return false;
    if (p.Nn % 32 != 0) return false;         // a workgroup owns 32 filter rows (half a 64-row block of U; a last half block stays unused)
    if (cdiv(p.OW, 2) < 9) return false;      // a workgroup's 64 linear tiles may then span more than 8 tile rows (8 tails are staged)
    if (cdiv(p.OH, 2) < 7) return false;      // ... or more than two images (the kernel's set-up handles one image wrap)
    for (int i = 0; i < p.nsrc; ++i)
        if (p.src[i].nch % 8 != 0) return false;
    // ranges of the kernel's offset arithmetic: rows x row pitch and columns x channels are 24-bit multiplies, and the tile
    // decomposition x / tiles_x is an exact float floor only for tiles_x < 2^12.  A shape beyond them takes the implicit GEMM
    // (launch_wino keeps the same conditions as internal assertions)
    for (int i = 0; i < p.nsrc; ++i)
        if ((long)p.src[i].W * p.src[i].C >= (1l << 23) || p.src[i].H >= (1 << 22)) return false;
    if ((long)(p.scatter == 2 ? p.DW : p.OW) * p.DC >= (1l << 23)) return false;
    if (cdiv(p.OW, 2) >= (1 << 12)) return false;
    return true;
}

size_t wino_u_floats(int Kc, int Nn) { return (size_t)16 * Kc * ((Nn + 63) / 64 * 64); }     // whole 64-row blocks

// true iff launch_igemm will hand this 3x3 launch to the Winograd kernel with an epilogue that can also write the 2x2
// max-pool of its output (IgemmP::pool_dst)
bool wino_fuses_pool(const IgemmP &p)
{
    return p.math == 3 && wino_applicable(p) && !p.scatter && !(p.OH & 1) && !(p.OW & 1) && !p.dn0 && p.DC == p.Nn;
}

// p must have passed launch_igemm's argument checks (launch_igemm calls this)
int launch_wino(const IgemmP &p, const float *U, hipStream_t st)
{
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
    // the kernel's offset arithmetic multiplies rows by row pitches and columns by channel counts with 24-bit multiplies
    for (int i = 0; i < p.nsrc; ++i)
        ARG_CHECK((long)p.src[i].W * p.src[i].C < (1l << 23) && p.src[i].H < (1 << 22), "wino: source row pitch %ld exceeds the 24-bit multiply range", (long)p.src[i].W * p.src[i].C);
    ARG_CHECK((long)(p.scatter == 2 ? p.DW : p.OW) * p.DC < (1l << 23), "wino: destination row pitch exceeds the 24-bit multiply range");
    ARG_CHECK(q.tiles_x < (1 << 12), "wino: %d tile columns exceed the exact range of the float-floor tile decomposition", q.tiles_x);
    q.p.ntiles = p.Nn / 32;
    // buffer-descriptor LDS-DMA needs every tensor below 2 GiB (32-bit num_records and the out-of-range marker);
    // larger tensors (config #5 at batch 16) and unet_set_lds_dma(0) take the global_load_lds instantiation
    bool buf = get_lds_dma_mode() != 0;
    {
        const size_t ub = wino_u_floats(kc, p.Nn) * sizeof(float);
        if (ub >= 0x7FFFFFFFull) buf = false;
        q.ubytes = (int)ub;
        for (int i = 0; i < 2; ++i) {
            q.xbytes[i] = 0;
            if (i < p.nsrc) {
                const size_t xb = (size_t)p.NB * p.src[i].H * p.src[i].W * p.src[i].C * sizeof(float);
                if (xb >= 0x7FFFFFFFull) buf = false;
                q.xbytes[i] = (int)xb;
            }
        }
    }
    // tile order: consecutive slots walk gn n-tiles of one m-tile, then the same n-tiles of the next m-tile (all m-tiles), then
    // the next group of n-tiles.  The 64 workgroups resident together on an XCD (32 CUs x 2) are then a 16 (m) x 4 (n) block of
    // the tile grid at about the same K step, and its L2 serves each patch stage to 4 and each U stage to 16 of them.  Round 3
    // had all n-tiles of an m-tile as neighbours (gn = ntiles: for the >= 512-channel layers 2 m-tiles x 32 n-tiles, i.e. the
    // whole 37-67 MB of transformed filters re-streamed per pair of m-tiles).  Measured per launch (rocprofv3 FETCH_SIZE x2 +
    // WRITE_SIZE, tools/wino_gn_pmc.sh; gn = all -> 8 -> 4): 8x30^2x1024->1024 1139 -> 520 -> 462 MB, 8x32^2x512->1024
    // 583 -> 263 -> 236, 8x56^2x1024->512 1041 -> 730 -> 714, 8x66^2x512->512 651 -> 461 -> 465; launch time unchanged within
    // 1 % (the counters count L2 misses, most of which the 256 MB Infinity Cache serves: these launches' whole working set fits it)
    static const int gn_env = [] { const char *e = getenv("UNET_WINO_GN"); return e ? atoi(e) : 4; }();
    q.gn = 1; q.gn_shift = 0;
    while (q.gn * 2 <= gn_env && q.p.ntiles % (q.gn * 2) == 0) { q.gn *= 2; ++q.gn_shift; }
    static bool attr32[64] = {false}, attr32b[64] = {false};
    auto k32 = buf ? wino32_f32_kernel<true> : wino32_f32_kernel<false>;
    if (int rc_ = ensure_dynamic_lds((const void *)k32, W32_TOTAL, buf ? attr32b : attr32)) return rc_;
    char tag32[96];
    snprintf(tag32, sizeof(tag32), "wino32<%d> M=%d N=%d Kd=%d nsrc=%d tiles=%d", (int)buf, p.M, p.Nn, p.Kd, p.nsrc, q.MT);
    prof_begin(PK_WINO, tag32, st, igemm_alg_flops(p), igemm_alg_flops(p) * (16.0 / 36.0), igemm_alg_bytes(p));
    hipLaunchKernelGGL(k32, dim3(q.p.mtiles * q.p.ntiles), dim3(256), W32_TOTAL, st, q);
    prof_end(st);
    HIP_TRY(hipGetLastError());
    return 0;
}

}  // namespace unet
