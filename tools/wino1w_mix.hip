// micro-benchmark for a ONE-WAVE-PER-SIMD fp32 Winograd step: 256-thread workgroup per CU, a wave holds 16 xi x 4 n-blocks of
// 16x16 accumulators (256 registers) and runs per step 128 v_mfma_f32_16x16x4_f32 in 8 groups of 16, each group with 4
// ds_read_b128 (filter fragments, requested one group ahead); interleaved per group: 2 ds_read_b64 + 4 packed adds (the next
// step's input transform) and, optionally, the register staging of the next stages (13 buffer loads of 16 B per lane from a 64
// MiB buffer at the top of the step, 13 ds_write_b128 of the previous step's loads) and one workgroup barrier per step.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <bool XF, bool STG, bool BAR>
__global__ __launch_bounds__(256, 1) void k(float *out, int iters, float seed, const float *src)
{
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, 1 << 26, 0x00020000);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 32768; i += 256) ((float *)smem)[i] = seed * (i & 7);
    __syncthreads();
    f32x4 acc[16][4];
    for (int x = 0; x < 16; ++x) for (int j = 0; j < 4; ++j) for (int r = 0; r < 4; ++r) acc[x][j][r] = 0.f;
    f32x2 v[16], vn[16];
    for (int x = 0; x < 16; ++x) { v[x][0] = seed + x + lane; v[x][1] = seed - x; vn[x] = v[x]; }
    f32x4 bf[2][4];
    for (int j = 0; j < 4; ++j) { bf[0][j] = f32x4{seed, 1.f, 2.f, 3.f}; bf[1][j] = bf[0][j]; }
    u32x4 st[13];
    for (int i = 0; i < 13; ++i) st[i] = u32x4{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
        if (STG) {
            // last step's loads -> LDS; this step's loads -> registers
#pragma unroll
            for (int i = 0; i < 13; ++i) *(u32x4 *)(smem + 65536 + ((wave * 13 + i) & 31) * 1024 + lane * 16) = st[i];
#pragma unroll
            for (int i = 0; i < 13; ++i)
                st[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, (int)(((unsigned)(blockIdx.x * 64 + it) * 65536u + (wave * 13 + i) * 1024u) & 0x3FFFFFFu), 0);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) bf[0][j] = *(const f32x4 *)(smem + 32768 + j * 8192 + lane * 16);
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const int xg = c >> 1, h = c & 1;
            if (c + 1 < 8) {
#pragma unroll
                for (int j = 0; j < 4; ++j) bf[(c + 1) & 1][j] = *(const f32x4 *)(smem + 32768 + j * 8192 + (c + 1) * 1024 + lane * 16);
            }
            if (XF) {
                // a quarter column of the next step's transform: 2 ds_read_b64 and 4 packed adds
                const f32x2 a0 = *(const f32x2 *)(smem + (2 * c) * 2048 + lane * 8), a1 = *(const f32x2 *)(smem + (2 * c + 1) * 2048 + lane * 8);
                vn[2 * c] = a0 - a1; vn[2 * c + 1] = a0 + a1;
                vn[(2 * c + 8) & 15] = vn[(2 * c + 8) & 15] - a0; vn[(2 * c + 9) & 15] = vn[(2 * c + 9) & 15] + a1;
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[4 * xg + e][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[4 * xg + e][h], bf[c & 1][j][e], acc[4 * xg + e][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (XF) {
#pragma unroll
            for (int x = 0; x < 16; ++x) v[x] = vn[x];
        }
        if (BAR) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
    }
    float s = 0;
    for (int x = 0; x < 16; ++x) for (int j = 0; j < 4; ++j) for (int r = 0; r < 4; ++r) s += acc[x][j][r];
    for (int i = 0; i < 13; ++i) s += (float)st[i].x;
    out[blockIdx.x * 256 + tid] = s;
}

template <bool XF, bool STG, bool BAR> void run(const char *name)
{
    const int blocks = 256, iters = 3000;
    float *out; hipMalloc(&out, blocks * 256 * 4);
    static float *src = nullptr; if (!src) { hipMalloc(&src, 1 << 26); hipMemset(src, 0, 1 << 26); }
    auto kern = k<XF, STG, BAR>;
    hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    kern<<<blocks, 256, 131072>>>(out, 100, 1e-3f, src);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    kern<<<blocks, 256, 131072>>>(out, iters, 1e-3f, src);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double fl = (double)blocks * 4 * iters * 128 * 2048.0;
    printf("%-60s %.3f ms  %.1f TFLOP/s executed = %.2f of 157.3\n", name, ms, fl / ms / 1e9, fl / ms / 1e9 / 157.3);
    hipFree(out);
}
int main()
{
    run<false, false, false>("128 MFMA + 32 ds_read_b128 per step");
    run<true, false, false>("+ interleaved transform (16 b64 + 32 packed adds)");
    run<true, false, true>("+ barrier per step");
    run<true, true, true>("+ register staging (13 loads, 13 ds_write_b128)");
    run<false, true, true>("MFMA + staging + barrier (no transform)");
    return 0;
}
