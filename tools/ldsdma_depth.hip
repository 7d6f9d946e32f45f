// micro-benchmark: L2/MALL/HBM -> LDS staging rate per CU of buffer LDS-DMA (buffer_load_dwordx4 ... lds) as a function of the
// number of stages in flight, with and without the bf16 MFMA work of an igemmb K step next to it.
//   256-thread workgroups, 2 per CU; a stage = SKB KiB (each wave issues SKB/4 one-KiB instructions); ring of DEPTH stages,
//   DEPTH-1 in flight while one is consumed; counted vmcnt + one barrier per stage like the real kernels.
//   region: bytes the whole grid cycles through (2 MiB: L2 hits; 128 MiB: MALL; 2 GiB: HBM).
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/ldsdma_depth tools/ldsdma_depth.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// the igemmb K step as it is: 128 x 128 tile, stage = 128 A rows + 128 B rows of 128 B (XOR-swizzled 16-B chunks), per wave
// 4 k-groups x (4 ds_read_b128 + 4 MFMA); PITCH = byte distance of consecutive A rows in memory (128: C = 64; 2048: C = 1024)
template <int PITCH, bool DBUF>
__global__ __launch_bounds__(256, 2) void kreal(const unsigned char *src, unsigned region_mask, int iters, float *out)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int STAGE = 32768;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, 0x7FFFFFFF, 0x00020000);
    const int srow = tid >> 3, schunk = (tid & 7) ^ ((srow >> 1) & 7);
    int a_off[4], b_off[4];
    for (int i = 0; i < 4; ++i) { a_off[i] = (srow + 32 * i) * PITCH + schunk * 16; b_off[i] = (srow + 32 * i) * 128 + schunk * 16; }
    unsigned pos = (unsigned)blockIdx.x * 7919u * (128u * PITCH);
    int issued = 0;
    auto issue = [&]() {
        unsigned char *ab = smem + (issued & 1) * STAGE + wave * 1024;
        const unsigned base = pos & region_mask;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)(ab + i * 4096), 16, a_off[i] + (int)base, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)(ab + 16384 + i * 4096), 16, b_off[i], (int)((issued & 63) * 16384), 0, 0);
        pos += PITCH == 128 ? 16384 : 128;
        ++issued;
    };
    f32x16 acc[2][2];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i >> 1][i & 1][r] = 0.f;
    const int l31 = lane & 31, lh = lane >> 5, swz = (l31 >> 1) & 7, wm = wave >> 1, wn = wave & 1;
    const int a_rd = (wm * 64 + l31) * 128, b_rd = 16384 + (wn * 64 + l31) * 128;
    issue();
    __syncthreads();
    for (int it = 0; it < iters; ++it) {
        issue();
        const unsigned char *sb = smem + (it & 1) * STAGE;
        bf16x8 fa[2][2], fb[2][2];
        auto rd = [&](int g, int q) {
            const int p = ((2 * g + lh) ^ swz) * 16;
            fa[q][0] = *(const bf16x8 *)(sb + a_rd + p); fa[q][1] = *(const bf16x8 *)(sb + a_rd + 4096 + p);
            fb[q][0] = *(const bf16x8 *)(sb + b_rd + p); fb[q][1] = *(const bf16x8 *)(sb + b_rd + 4096 + p);
        };
        rd(0, 0);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int q = DBUF ? (g & 1) : 0;
            if (DBUF) { if (g + 1 < 4) rd(g + 1, q ^ 1); __builtin_amdgcn_sched_barrier(0); }
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[q][0], fb[q][0], acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[q][0], fb[q][1], acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[q][1], fb[q][0], acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[q][1], fb[q][1], acc[1][1], 0, 0, 0);
            if (DBUF) __builtin_amdgcn_sched_barrier(0);
            else if (g + 1 < 4) rd(g + 1, 0);
        }
        __syncthreads();
    }
    float s = 0;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i >> 1][i & 1][r];
    if (s == 12345.f) out[tid] = s;
}

// the same tile with 32-channel K steps: 64-byte LDS rows (chunk swizzle (row >> 2) & 3), 16 KiB stages, DEPTH-deep ring,
// WPC workgroups per CU (LDS DEPTH x 16 KiB each)
template <int PITCH, int DEPTH, int WPC>
__global__ __launch_bounds__(256, WPC) void kreal32(const unsigned char *src, unsigned region_mask, int iters, float *out)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int STAGE = 16384;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, 0x7FFFFFFF, 0x00020000);
    const int srow = tid >> 2, schunk = (tid & 3) ^ ((srow >> 2) & 3);
    int a_off[2], b_off[2];
    for (int i = 0; i < 2; ++i) { a_off[i] = (srow + 64 * i) * PITCH + schunk * 16; b_off[i] = (srow + 64 * i) * 128 + schunk * 16; }
    unsigned pos = (unsigned)blockIdx.x * 7919u * (128u * PITCH);
    int issued = 0;
    auto issue = [&]() {
        unsigned char *ab = smem + (issued % DEPTH) * STAGE + wave * 1024;
        const unsigned base = pos & region_mask;
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)(ab + i * 4096), 16, a_off[i] + (int)base, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)(ab + 8192 + i * 4096), 16, b_off[i], (int)((issued & 63) * 16384), 0, 0);
        pos += PITCH == 128 ? ((issued & 1) ? 16384 - 64 : 64) : 64;
        ++issued;
    };
    f32x16 acc[2][2];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i >> 1][i & 1][r] = 0.f;
    const int l31 = lane & 31, lh = lane >> 5, swz = (l31 >> 2) & 3, wm = wave >> 1, wn = wave & 1;
    const int a_rd = (wm * 64 + l31) * 64, b_rd = 8192 + (wn * 64 + l31) * 64;
    for (int d = 0; d < DEPTH - 1; ++d) issue();
    for (int it = 0; it < iters; ++it) {
        issue();
        wait_vm<4 * (DEPTH - 1)>();
        __builtin_amdgcn_s_barrier();
        const unsigned char *sb = smem + (it % DEPTH) * STAGE;
        bf16x8 fa[2][2], fb[2][2];
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const int p = ((2 * g + lh) ^ swz) * 16;
            fa[g][0] = *(const bf16x8 *)(sb + a_rd + p); fa[g][1] = *(const bf16x8 *)(sb + a_rd + 2048 + p);
            fb[g][0] = *(const bf16x8 *)(sb + b_rd + p); fb[g][1] = *(const bf16x8 *)(sb + b_rd + 2048 + p);
        }
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[g][0], fb[g][0], acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[g][0], fb[g][1], acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[g][1], fb[g][0], acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[g][1], fb[g][1], acc[1][1], 0, 0, 0);
        }
        __builtin_amdgcn_s_barrier();
    }
    wait_vm<0>();
    float s = 0;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i >> 1][i & 1][r];
    if (s == 12345.f) out[tid] = s;
}

template <int PITCH, int DEPTH, int WPC> void run_real32(const unsigned char *src, size_t region, int blocks, int iters)
{
    float *out; hipMalloc(&out, 4096);
    auto kern = kreal32<PITCH, DEPTH, WPC>;
    hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, DEPTH * 16384);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const unsigned mask = (unsigned)(region - 1) & ~(unsigned)(128 * PITCH - 1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), DEPTH * 16384, 0, src, mask, iters, out);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), DEPTH * 16384, 0, src, mask, iters, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double tf = (double)blocks * 4 * iters * 8 * 32768.0 / ms / 1e9;
    printf("igemmb-step32 pitch %4d depth %d wg/CU %d region %5zu MiB blocks %d: %.3f ms  %.0f TFLOP/s  (%.0f clk/stage)\n", PITCH, DEPTH, WPC, region >> 20, blocks, ms,
           tf, ms * 1e-3 * 2.4e9 / iters);
    hipFree(out);
}

template <int PITCH, bool DBUF> void run_real(const unsigned char *src, size_t region, int blocks, int iters)
{
    float *out; hipMalloc(&out, 4096);
    auto kern = kreal<PITCH, DBUF>;
    hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const unsigned mask = (unsigned)(region - 1) & ~(unsigned)(128 * PITCH - 1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 65536, 0, src, mask, iters, out);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 65536, 0, src, mask, iters, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double tf = (double)blocks * 4 * iters * 16 * 32768.0 / ms / 1e9;
    printf("igemmb-step pitch %4d dbuf %d region %5zu MiB blocks %d: %.3f ms  %.0f TFLOP/s  (%.0f clk/stage)\n", PITCH, (int)DBUF, region >> 20, blocks, ms,
           tf, ms * 1e-3 * 2.4e9 / iters);
    hipFree(out);
}

template <int DEPTH, int SKB, int NMFMA>
__global__ __launch_bounds__(256, 2) void k(const unsigned char *src, unsigned region_mask, int iters, float *out)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int STAGE = SKB * 1024, IPW = SKB / 4;                 // instructions per wave per stage
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, 0x7FFFFFFF, 0x00020000);
    unsigned pos = (unsigned)blockIdx.x * 7919u * STAGE;
    int issued = 0;
    auto issue = [&]() {
        unsigned char *sb = smem + (issued % DEPTH) * STAGE + wave * (IPW * 1024);
        const unsigned base = pos & region_mask;
#pragma unroll
        for (int i = 0; i < IPW; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)(sb + i * 1024), 16,
                                                     (int)(wave * (IPW * 1024) + i * 1024 + lane * 16), (int)base, 0, 0);
        pos += STAGE;
        ++issued;
    };
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int d = 0; d < DEPTH - 1; ++d) issue();
    for (int it = 0; it < iters; ++it) {
        issue();
        wait_vm<IPW *(DEPTH - 1)>();
        __builtin_amdgcn_s_barrier();
        if (NMFMA) {
            const unsigned char *sb = smem + (it % DEPTH) * STAGE;
            bf16x8 a = *(const bf16x8 *)(sb + lane * 16), b = *(const bf16x8 *)(sb + 4096 + lane * 16);
#pragma unroll
            for (int m = 0; m < NMFMA; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m & 3], 0, 0, 0);
        }
        __builtin_amdgcn_s_barrier();
    }
    wait_vm<0>();
    float s = 0;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 12345.f) out[tid] = s;
}

template <int DEPTH, int SKB, int NMFMA> void run(const unsigned char *src, size_t region, int blocks, int iters)
{
    float *out; hipMalloc(&out, 4096);
    auto kern = k<DEPTH, SKB, NMFMA>;
    hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, DEPTH * SKB * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const unsigned mask = (unsigned)(region - 1) & ~(unsigned)(SKB * 1024 - 1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), DEPTH * SKB * 1024, 0, src, mask, iters, out);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), DEPTH * SKB * 1024, 0, src, mask, iters, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)blocks * iters * SKB * 1024.0;
    const double bpc = bytes / (ms * 1e-3) / 2.4e9 / 256.0;
    const double tf = (double)blocks * 4 * iters * NMFMA * 32768.0 / ms / 1e9;
    printf("depth %d stage %2d KiB mfma %2d region %5zu MiB blocks %d: %.3f ms  %.2f TB/s  %.1f B/clk/CU  %.0f TFLOP/s  (%.0f clk/stage)\n", DEPTH, SKB,
           NMFMA, region >> 20, blocks, ms, bytes / ms / 1e9, bpc, tf, ms * 1e-3 * 2.4e9 / iters);
    hipFree(out);
}

int main()
{
    const size_t total = (size_t)1 << 31;
    unsigned char *src; hipMalloc(&src, total);
    hipMemset(src, 0, total);
    const size_t regions[3] = {(size_t)2 << 20, (size_t)128 << 20, (size_t)1 << 31};
    for (size_t rg : regions) {
        run_real<128, true>(src, rg, 512, 400);
        run_real<128, false>(src, rg, 512, 400);
        run_real<2048, true>(src, rg, 512, 400);
        run_real<128, true>(src, rg, 5120, 72);
        run_real32<128, 2, 4>(src, rg, 1024, 800);
        run_real32<128, 3, 3>(src, rg, 768, 800);
        run_real32<128, 4, 2>(src, rg, 512, 800);
        run_real32<128, 2, 2>(src, rg, 512, 800);
        run_real32<2048, 2, 4>(src, rg, 1024, 800);
        run_real32<128, 2, 4>(src, rg, 10240, 144);
    }
    if (getenv("ONLY_REAL")) return 0;
    for (size_t rg : regions) {
        run<2, 32, 0>(src, rg, 512, 400);
        run<2, 32, 16>(src, rg, 512, 400);
        run<3, 16, 0>(src, rg, 512, 800);
        run<3, 16, 8>(src, rg, 512, 800);
        run<4, 16, 0>(src, rg, 512, 800);
        run<4, 16, 8>(src, rg, 512, 800);
        run<5, 16, 0>(src, rg, 512, 800);
        run<5, 16, 8>(src, rg, 512, 800);
        run<2, 16, 8>(src, rg, 512, 800);
        run<3, 24, 12>(src, rg, 512, 600);
        run<8, 8, 4>(src, rg, 512, 1600);
    }
    // one workgroup per CU with a deep ring of big stages (a 256 x 128 tile: 48 KiB per 64-channel step, 32 MFMAs per wave of 4)
    for (size_t rg : regions) {
        run<3, 48, 32>(src, rg, 256, 400);
        run<3, 48, 0>(src, rg, 256, 400);
    }
    hipFree(src);
    return 0;
}
