// micro-benchmark: what the bf16 matrix pipe of one MI355X delivers with v_mfma_f32_32x32x16_bf16 on register operands only
// (no LDS, no memory): NACC independent accumulators per wave, 1 or 2 waves per SIMD, in-kernel cycle counter for the clock.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int NACC, int WPS>
__global__ __launch_bounds__(256, WPS) void k(float *out, unsigned long long *cyc, int iters, float a0)
{
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8 a[4], b[4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) { a[i][j] = (__bf16)(a0 + threadIdx.x + i); b[i][j] = (__bf16)(a0 + j); }
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(t + i) & 3], b[t & 3], acc[i], 0, 0, 0);
        asm volatile("" : "+v"(a[0]), "+v"(b[0]));
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) *cyc = t1 - t0;
}
template <int NACC, int WPS> void run(int blocks, int iters, const char *name)
{
    float *out; unsigned long long *cyc, h = 0;
    hipMalloc(&out, blocks * 256 * 4); hipMalloc(&cyc, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<NACC, WPS><<<blocks, 256>>>(out, cyc, iters, 1.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<NACC, WPS><<<blocks, 256>>>(out, cyc, iters, 1.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    const double nm = (double)iters * 8 * NACC;                       // MFMAs per wave
    const double fl = (double)blocks * 4 * nm * 32768.0;
    printf("%-34s blocks=%d: %.3f ms  %.0f TFLOP/s (%.2f of 2500)  %.1f counter ticks per MFMA per wave  counter %.2f GHz\n", name, blocks, ms, fl / ms / 1e9,
           fl / ms / 1e9 / 2500.0, (double)h / nm, (double)h / (ms * 1e6));
    hipFree(out); hipFree(cyc);
}
int main()
{
    run<4, 1>(256, 4000, "4 acc, 1 wave/SIMD");
    run<8, 1>(256, 2000, "8 acc, 1 wave/SIMD");
    run<4, 2>(512, 4000, "4 acc, 2 waves/SIMD");
    run<8, 2>(512, 2000, "8 acc, 2 waves/SIMD");
    run<2, 2>(512, 8000, "2 acc, 2 waves/SIMD");
    run<1, 1>(256, 16000, "1 acc (dependent chain), 1 wave");
    return 0;
}
