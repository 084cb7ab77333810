// microbenchmark: issue rate of v_mfma_f32_32x32x2_f32 with 1, 2, 4 interleaved accumulator chains
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void k(float* out, int iters, float a0, float b0) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = (float)(threadIdx.x + i + r);
    float a = a0 + threadIdx.x, b = b0 + threadIdx.x * 0.5f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 64 / NACC; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC> void run(const char* name) {
    float* out; (void)hipMalloc(&out, 256 * 256 * 4);
    const int iters = 2000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<NACC>, dim3(256), dim3(256), 0, 0, out, 10, 1.0f, 2.0f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<NACC>, dim3(256), dim3(256), 0, 0, out, iters, 1.0f, 2.0f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double mfma = (double)iters * 64;                 // per wave
    double flops = mfma * 32 * 32 * 2 * 2 * 1024;     // 1024 waves
    printf("%s: %.3f ms, %.1f TFLOP/s, %.1f ns per MFMA (=%.1f cycles @2.4GHz)\n", name, ms, flops / ms / 1e9, ms * 1e6 / mfma, ms * 1e6 / mfma * 2.4);
    (void)hipFree(out);
}
int main() { run<1>("1 chain "); run<2>("2 chains"); run<4>("4 chains"); run<1>("1 chain "); return 0; }
