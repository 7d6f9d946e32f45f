// micro-benchmark: does packed fp32 VALU work (v_pk_add_f32) run in the shadow of v_mfma_f32_16x16x4_f32 the way plain VALU work
// (v_add_f32) does?  Per MFMA: NV VALU instructions of the chosen kind (inline asm, so that the compiler neither packs nor
// unpacks them), on registers no MFMA touches.  WPS waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int KIND, int NV, int WPS>
__global__ __launch_bounds__(256 * WPS, 1) void k(float *out, int iters, float seed)
{
    const int lane = threadIdx.x & 63;
    f32x4 acc[16];
    for (int x = 0; x < 16; ++x) for (int r = 0; r < 4; ++r) acc[x][r] = 0.f;
    f32x2 t[8];
    for (int i = 0; i < 8; ++i) { t[i][0] = seed * i + lane; t[i][1] = seed - i; }
    float a = seed + lane, b = seed;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 64; ++m) {
            acc[m & 15] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[m & 15], 0, 0, 0);
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const int i = (m * NV + v) & 7, j = (i + 3) & 7;
                if (KIND == 1) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(t[i]) : "v"(t[j]));
                if (KIND == 2) asm volatile("v_add_f32 %0, %0, %1" : "+v"(t[i][0]) : "v"(t[j][0]));
                if (KIND == 3) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(t[i]) : "v"(t[j]));
                if (KIND == 4) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(t[i]) : "v"(t[j]));
                if (KIND == 5) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(t[i][0]) : "v"(t[j][0]));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0;
    for (int x = 0; x < 16; ++x) for (int r = 0; r < 4; ++r) s += acc[x][r];
    for (int i = 0; i < 8; ++i) s += t[i][0] + t[i][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND, int NV, int WPS> void run(const char *name)
{
    const int blocks = 256, iters = 4000;
    float *out; hipMalloc(&out, blocks * 256 * WPS * 4);
    auto kern = k<KIND, NV, WPS>;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    kern<<<blocks, 256 * WPS>>>(out, 100, 1e-3f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    kern<<<blocks, 256 * WPS>>>(out, iters, 1e-3f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double fl = (double)blocks * 4 * WPS * iters * 64 * 2048.0;
    printf("%-52s %d w/SIMD  %.3f ms  %.1f TFLOP/s executed = %.2f of 157.3\n", name, WPS, ms, fl / ms / 1e9, fl / ms / 1e9 / 157.3);
    hipFree(out);
}
template <int WPS> void all()
{
    run<0, 0, WPS>("MFMA only");
    run<2, 1, WPS>("+ 1 v_add_f32 per MFMA");
    run<2, 2, WPS>("+ 2 v_add_f32 per MFMA");
    run<2, 4, WPS>("+ 4 v_add_f32 per MFMA");
    run<1, 1, WPS>("+ 1 v_pk_add_f32 per MFMA");
    run<1, 2, WPS>("+ 2 v_pk_add_f32 per MFMA");
    run<1, 4, WPS>("+ 4 v_pk_add_f32 per MFMA");
    run<3, 2, WPS>("+ 2 v_pk_mul_f32 per MFMA");
    run<4, 2, WPS>("+ 2 v_pk_fma_f32 per MFMA");
    run<5, 2, WPS>("+ 2 v_fma_f32 per MFMA");
    run<5, 4, WPS>("+ 4 v_fma_f32 per MFMA");
}
int main() { all<1>(); all<2>(); return 0; }
