// micro-benchmark: v_mfma_f32_32x32x16_bf16 fed from LDS the way convb64 / igemmb feed it: per step FRAGS ds_read_b128 (3-deep
// register ring) and MF MFMAs on 4 accumulators, one wave per SIMD (256-thread workgroup, 1 per CU) or two.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int FRAGS, int WPS>
__global__ __launch_bounds__(256, WPS) void k(float *out, unsigned long long *cyc, int iters)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    for (int i = threadIdx.x; i < 16384; i += 256) ((float *)smem)[i] = 0.001f * i;
    __syncthreads();
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    int base[4];
    for (int f = 0; f < 4; ++f) base[f] = (wave * 8192 + f * 2048 + l31 * 128 + ((lh ^ ((l31 >> 1) & 7)) << 4)) & 65535;
    bf16x8 fr[3][4];
    for (int f = 0; f < 4; ++f) for (int s = 0; s < 3; ++s) fr[s][f] = *(const bf16x8 *)(smem + base[f]);
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int st = 0; st < 12; ++st) {
#pragma unroll
            for (int f = 0; f < FRAGS; ++f) fr[(st + 2) % 3][f] = *(const bf16x8 *)(smem + (base[f] ^ ((st & 3) << 5)));
            __builtin_amdgcn_sched_barrier(0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[st % 3][0], fr[st % 3][2], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[st % 3][0], fr[st % 3][3], acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[st % 3][1], fr[st % 3][2], acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[st % 3][1], fr[st % 3][3], acc[3], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) *cyc = t1 - t0;
}
template <int FRAGS, int WPS> void run(int iters, const char *name)
{
    const int blocks = 256 * WPS;
    float *out; unsigned long long *cyc, h = 0;
    hipMalloc(&out, blocks * 256 * 4); hipMalloc(&cyc, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipFuncSetAttribute((const void *)k<FRAGS, WPS>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    k<FRAGS, WPS><<<blocks, 256, 65536>>>(out, cyc, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<FRAGS, WPS><<<blocks, 256, 65536>>>(out, cyc, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    const double nm = (double)iters * 12 * 4;
    const double fl = (double)blocks * 4 * nm * 32768.0;
    printf("%-44s %.3f ms  %.0f TFLOP/s (%.2f of 2500)  %.1f ticks per MFMA  LDS %.0f B/clk/CU at 2.1 GHz\n", name, ms, fl / ms / 1e9, fl / ms / 1e9 / 2500.0,
           (double)h / nm, (double)blocks * 4 * iters * 12.0 * FRAGS * 1024.0 / (ms * 1e-3) / 256.0 / 2.1e9);
    hipFree(out); hipFree(cyc);
}
int main()
{
    run<0, 1>(4000, "no LDS reads, 1 wave/SIMD");
    run<1, 1>(4000, "1 ds_read_b128 per 4 MFMA, 1 wave/SIMD");
    run<2, 1>(4000, "2 ds_read_b128 per 4 MFMA, 1 wave/SIMD");
    run<4, 1>(4000, "4 ds_read_b128 per 4 MFMA, 1 wave/SIMD");
    run<0, 2>(4000, "no LDS reads, 2 waves/SIMD");
    run<2, 2>(4000, "2 ds_read_b128 per 4 MFMA, 2 waves/SIMD");
    run<4, 2>(4000, "4 ds_read_b128 per 4 MFMA, 2 waves/SIMD");
    return 0;
}
