// micro-benchmark: how busy can the fp32 matrix pipe get with the Winograd kernels' instruction mix?
// 2 waves per SIMD (512 threads, one workgroup per CU), 16x16x4 fp32 MFMAs on 32 independent accumulators, and per
// 64 MFMAs optionally 32 packed adds (the transform), 16 ds_read_b128 + 16 ds_read_b64 (the operand reads) and a
// workgroup barrier — and optionally N buffer LDS-DMA instructions per wave from a 64 MiB buffer, waited for at the barrier of the same step (distance 1).  Built and run on the GPU box:
//   hipcc -O3 --offload-arch=gfx950 tools/wino_mix_peak.hip -o /tmp/mix && /tmp/mix
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <bool VALU, bool LDSR, bool BAR, int DMA>
__global__ __launch_bounds__(512, 1) void k(float *out, int iters, float seed, const float *src)
{
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, 1 << 26, 0x00020000);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 16384; i += 512) ((float *)smem)[i] = seed * (i & 7);
    __syncthreads();
    f32x4 acc[16][2];
    for (int x = 0; x < 16; ++x) for (int j = 0; j < 2; ++j) for (int r = 0; r < 4; ++r) acc[x][j][r] = 0.f;
    f32x2 v[16];
    for (int x = 0; x < 16; ++x) { v[x][0] = seed + x + lane; v[x][1] = seed - x; }
    f32x4 b0 = {seed, 1.f, 2.f, 3.f}, b1 = {1.f, seed, 3.f, 2.f};
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int it = 0; it < iters; ++it) {
        if (DMA) {
            // DMA instructions per wave per 64 MFMAs into a scratch LDS region, from a 64 MiB buffer walked linearly (L2/MALL)
#pragma unroll
            for (int i = 0; i < DMA; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)(smem + 49152 + ((wave * DMA + i) & 15) * 1024), 16, lane * 16,
                                                         (int)(((unsigned)(blockIdx.x * 64 + it) * 65536u + (wave * DMA + i) * 1024u) & 0x3FFFFFFu), 0, 0);
        }
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const int xg = c >> 1, h = c & 1;
            if (LDSR) {
                b0 = *(const f32x4 *)(smem + 32768 + ((c * 2) & 15) * 1024 + lane * 16);
                b1 = *(const f32x4 *)(smem + 32768 + ((c * 2 + 1) & 15) * 1024 + lane * 16);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc[4 * xg + e][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[4 * xg + e][h], b0[e], acc[4 * xg + e][0], 0, 0, 0);
                acc[4 * xg + e][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[4 * xg + e][h], b1[e], acc[4 * xg + e][1], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (LDSR) {
#pragma unroll
            for (int q = 0; q < 16; ++q) v[q] = *(const f32x2 *)(smem + q * 2048 + lane * 8);
        }
        if (VALU) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x2 a0 = v[j], a1 = v[4 + j], a2 = v[8 + j], a3 = v[12 + j];
                v[j] = a0 - a2; v[4 + j] = a1 + a2; v[8 + j] = a2 - a1; v[12 + j] = a1 - a3;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f32x2 t0 = v[4 * i], t1 = v[4 * i + 1], t2 = v[4 * i + 2], t3 = v[4 * i + 3];
                v[4 * i] = t0 - t2; v[4 * i + 1] = t1 + t2; v[4 * i + 2] = t2 - t1; v[4 * i + 3] = t1 - t3;
            }
        }
        if (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (BAR) __builtin_amdgcn_s_barrier();
    }
    float s = 0;
    for (int x = 0; x < 16; ++x) for (int j = 0; j < 2; ++j) for (int r = 0; r < 4; ++r) s += acc[x][j][r];
    out[blockIdx.x * 512 + tid] = s;
}

template <bool VALU, bool LDSR, bool BAR, int DMA> void run(const char *name)
{
    const int blocks = 256, iters = 4000;
    float *out; hipMalloc(&out, blocks * 512 * 4);
    static float *src = nullptr; if (!src) { hipMalloc(&src, 1 << 26); hipMemset(src, 0, 1 << 26); }
    auto kern = k<VALU, LDSR, BAR, DMA>;
    hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    kern<<<blocks, 512, 65536>>>(out, 100, 1e-3f, src);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    kern<<<blocks, 512, 65536>>>(out, iters, 1e-3f, src);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double fl = (double)blocks * 8 * iters * 64 * 2048.0;       // 8 waves x 64 MFMAs x 2*16*16*4 flops
    printf("%-44s %.3f ms  %.1f TFLOP/s executed = %.2f of 157.3\n", name, ms, fl / ms / 1e9, fl / ms / 1e9 / 157.3);
    hipFree(out);
}
int main()
{
    run<false, false, false, 0>("MFMA only");
    run<true, false, false, 0>("+ 32 packed adds / 64 MFMA");
    run<false, true, false, 0>("+ 16 b128 + 16 b64 LDS reads / 64 MFMA");
    run<true, true, false, 0>("+ both");
    run<true, true, true, 0>("+ both + barrier per 64 MFMA");
    run<false, false, true, 6>("MFMA + 6 LDS-DMA/wave (wait at the barrier)");
    run<true, true, true, 6>("everything + 6 LDS-DMA/wave");
    run<true, true, true, 9>("everything + 9 LDS-DMA/wave");
    return 0;
}
