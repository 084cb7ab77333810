// microbenchmark: what read bandwidth does the chip deliver to 256 workgroups of 4 waves (one per CU) streaming 4 GiB once?
// (0) register loads, 16 B per lane, each wave its own contiguous 4 KiB per iteration (8 loads in flight);
// (1) the same through LDS-DMA (global_load_lds_dwordx4, 16 KiB in flight per wave);
// (2) LDS-DMA in the weight-gradient kernel's pattern: a load instruction = two 512-byte runs 8 KiB apart, sixteen per step.
// hipcc --offload-arch=gfx950 -O3 hbm_read_rate.hip -o hbm_read_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define GLB(p) ((const __attribute__((address_space(1))) void*)(p))
#define LDSP(p) ((__attribute__((address_space(3))) void*)(p))

template <int MODE>
__global__ __launch_bounds__(256) void k(const char* src, size_t bytes, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t n_waves = (size_t)gridDim.x * 4, w = (size_t)blockIdx.x * 4 + wave;
    const size_t per_wave = bytes / n_waves;          // contiguous share of each wave
    const char* g = src + w * per_wave;
    char* my = lds + wave * 32768;
    f32x4 acc = {0, 0, 0, 0};
    const size_t iters = per_wave / 16384;
    for (size_t it = 0; it < iters; ++it) {
        if (MODE == 0) {
            f32x4 r[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) r[j] = *(const f32x4*)(g + j * 1024 + lane * 16);
#pragma unroll
            for (int j = 0; j < 16; ++j) acc += r[j];
        } else if (MODE == 1) {
#pragma unroll
            for (int j = 0; j < 16; ++j) __builtin_amdgcn_global_load_lds(GLB(g + j * 1024 + lane * 16), LDSP(my + (it & 1) * 16384 + j * 1024), 16, 0, 0);
            asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        } else {
            // a 32 KiB window = 32 pieces of 1 KiB, consumed in two steps: a step takes one 512-byte half of every piece,
            // lanes 0-31 of load j from piece j, lanes 32-63 from piece j + 16
            const int half = (int)(it & 1);
            const char* base = g - half * 16384;
#pragma unroll
            for (int j = 0; j < 16; ++j)
                __builtin_amdgcn_global_load_lds(GLB(base + (size_t)(j + 16 * (lane >> 5)) * 1024 + half * 512 + (lane & 31) * 16),
                                                 LDSP(my + half * 16384 + j * 1024), 16, 0, 0);
            asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        }
        g += 16384;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    if (MODE != 0) acc = *(f32x4*)(lds + threadIdx.x * 16);
    if (acc[0] == 12345.0f) sink[0] = acc[1];
}

int main() {
    const size_t bytes = (size_t)4 << 30;
    char* src;
    float* sink;
    if (hipMalloc(&src, bytes + (1 << 20)) != hipSuccess) return 1;
    (void)hipMalloc(&sink, 64);
    (void)hipMemset(src, 0, bytes);
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    (void)hipFuncSetAttribute((const void*)k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 32768);
    (void)hipFuncSetAttribute((const void*)k<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 32768);
    const char* names[3] = {"register loads, 16 B per lane, contiguous", "LDS-DMA, contiguous KiB per instruction", "LDS-DMA, two 512-byte runs per instruction (dW pattern)"};
    for (int rep = 0; rep < 3; ++rep)
        for (int m = 0; m < 3; ++m)
            for (int grid = 256; grid <= 256; grid += 256) {
                (void)hipEventRecord(a);
                if (m == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, src, bytes, sink);
                if (m == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 4 * 32768, 0, src, bytes, sink);
                if (m == 2) hipLaunchKernelGGL(k<2>, dim3(grid), dim3(256), 4 * 32768, 0, src, bytes, sink);
                (void)hipEventRecord(b);
                (void)hipEventSynchronize(b);
                float ms;
                (void)hipEventElapsedTime(&ms, a, b);
                if (hipGetLastError() != hipSuccess) printf("launch failed\n");
                if (rep == 2) printf("%-58s grid %3d: %7.3f ms  %5.2f TB/s\n", names[m], grid, ms, bytes / (ms * 1e-3) / 1e12);
            }
    return 0;
}
