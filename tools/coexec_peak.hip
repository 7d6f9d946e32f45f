// micro-benchmark: can fp32 MFMA waves and fp32 VALU (v_pk_fma_f32) waves on the same SIMDs add up?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
// mode: 0 = all waves MFMA, 1 = all waves VALU, 2 = even waves MFMA / odd waves VALU (8 waves per block, 2 per SIMD)
__global__ __launch_bounds__(512, 2) void k(float *out, int iters, int mode, float a0)
{
    const int wave = threadIdx.x >> 6;
    const bool do_mfma = mode == 0 || (mode == 2 && (wave < 4));
    float s = 0;
    if (do_mfma) {
        f32x16 acc[4];
        for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        float a = a0 + threadIdx.x, b = 2.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int t = 0; t < 16; ++t)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
            asm volatile("" : "+v"(a), "+v"(b));
        }
        for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    } else {
        f32x2 acc[32];
        for (int i = 0; i < 32; ++i) acc[i] = f32x2{0.f, 0.f};
        f32x2 a = {a0 + threadIdx.x, a0}, b = {1.0001f, 0.9999f};
        for (int it = 0; it < iters; ++it) {
            // same flops as the MFMA branch per iteration: 64 MFMA x 4096 flop / 64 lanes = 4096 flop per lane = 1024 pk_fma
#pragma unroll 4
            for (int t = 0; t < 32; ++t)
#pragma unroll
                for (int i = 0; i < 32; ++i) acc[i] = __builtin_elementwise_fma(a, b, acc[i]);
            asm volatile("" : "+v"(a), "+v"(b));
        }
        for (int i = 0; i < 32; ++i) s += acc[i][0] + acc[i][1];
    }
    out[blockIdx.x * 512 + threadIdx.x] = s;
}
void run(int mode, const char *name)
{
    const int blocks = 512, iters = 400;
    float *out; hipMalloc(&out, blocks * 512 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<<<blocks, 512>>>(out, iters, mode, 1.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<<<blocks, 512>>>(out, iters, mode, 1.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double fl = (double)blocks * 8 * iters * 64 * 4096.0;
    printf("%s: %.3f ms  %.1f TFLOP/s\n", name, ms, fl / ms / 1e9);
    hipFree(out);
}
int main() { run(0, "all MFMA      "); run(1, "all VALU pk_fma"); run(2, "half MFMA / half VALU"); return 0; }
