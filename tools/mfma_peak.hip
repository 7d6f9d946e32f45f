// micro-benchmark: peak rate of v_mfma_f32_32x32x2_f32 in the shapes the igemm kernel uses
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256, 2) void k(float *out, int iters, float a0, float b0)
{
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = a0 + threadIdx.x, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < 16; ++t)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
        asm volatile("" : "+v"(a), "+v"(b));
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC> void run(int blocks, int iters, const char *name)
{
    float *out; hipMalloc(&out, blocks * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<NACC><<<blocks, 256>>>(out, iters, 1.f, 2.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<NACC><<<blocks, 256>>>(out, iters, 1.f, 2.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double fl = (double)blocks * 4 * iters * 16 * NACC * 4096.0;
    printf("%s blocks=%d iters=%d: %.3f ms  %.1f TFLOP/s\n", name, blocks, iters, ms, fl / ms / 1e9);
    hipFree(out);
}
int main()
{
    run<4>(256, 2000, "acc4 1blk/CU");
    run<4>(512, 2000, "acc4 2blk/CU");
    run<4>(1024, 1000, "acc4 4 rounds");
    run<4>(5120, 200, "acc4 10 rounds short");
    run<4>(10240, 18, "acc4 nk=18-like (18*64 mfma per block)");
    run<4>(10240, 72, "acc4 nk=72-like");
    run<1>(512, 8000, "acc1 2blk/CU");
    return 0;
}
