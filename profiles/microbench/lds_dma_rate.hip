// microbenchmark: how fast does a CU move L2-resident data into LDS, by LDS-DMA (global_load_lds_dwordx4 / _dword) and by
// register loads + ds_write_b128? One workgroup of 4 waves per CU (as the MLP and weight-gradient kernels run), every wave streams
// the same 2 MiB window again and again (L2 hits after the first pass), 8 instructions in flight per wave.
// hipcc --offload-arch=gfx950 -O3 lds_dma_rate.hip -o lds_dma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define GLB(p) ((const __attribute__((address_space(1))) void*)(p))
#define LDSP(p) ((__attribute__((address_space(3))) void*)(p))

template <int MODE>   // 0: LDS-DMA 16 B per lane; 1: LDS-DMA 4 B per lane; 2: register loads of 16 B + ds_write_b128; 3: register loads only
__global__ __launch_bounds__(256) void k(const char* src, int iters, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    char* my = lds + wave * 8 * 1024;
    const char* g = src + ((blockIdx.x * 4 + wave) & 255) * 8192 + lane * 16;      // a 2 MiB window
    f32x4 acc = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j) __builtin_amdgcn_global_load_lds(GLB(g + j * 1024), LDSP(my + j * 1024), 16, 0, 0);
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        } else if (MODE == 1) {
#pragma unroll
            for (int j = 0; j < 8; ++j) __builtin_amdgcn_global_load_lds(GLB(g - lane * 12 + j * 256), LDSP(my + j * 256), 4, 0, 0);
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        } else {
            f32x4 r[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) r[j] = *(const f32x4*)(g + j * 1024);
            if (MODE == 2) {
#pragma unroll
                for (int j = 0; j < 8; ++j) *(f32x4*)(my + j * 1024 + lane * 16) = r[j];
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) acc += r[j];
            }
        }
        g += (it & 1) ? -65536 : 65536;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    if (MODE != 3) acc = *(f32x4*)(lds + threadIdx.x * 16);
    if (acc[0] == 12345.0f) sink[0] = acc[1];
}

int main() {
    char* src;
    float* sink;
    hipMalloc(&src, 8 << 20);
    hipMalloc(&sink, 64);
    hipMemset(src, 0, 8 << 20);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    const int iters = 20000;
    const char* names[4] = {"LDS-DMA dwordx4 (1 KiB per instruction)", "LDS-DMA dword (256 B per instruction)", "global_load_dwordx4 + ds_write_b128",
                            "global_load_dwordx4 only"};
    for (int rep = 0; rep < 2; ++rep)
        for (int m = 0; m < 4; ++m) {
            hipEventRecord(a);
            if (m == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(256), 32768, 0, src, iters, sink);
            if (m == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), 32768, 0, src, iters, sink);
            if (m == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(256), 32768, 0, src, iters, sink);
            if (m == 3) hipLaunchKernelGGL(k<3>, dim3(256), dim3(256), 32768, 0, src, iters, sink);
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms;
            hipEventElapsedTime(&ms, a, b);
            const double bytes_per_cu = (double)iters * 4 * 8 * (m == 1 ? 256 : 1024);
            if (rep) printf("%-42s %8.3f ms  %6.1f GB/s per CU  %5.1f B/clk at 2.4 GHz  (%.2f TB/s chip)\n", names[m], ms, bytes_per_cu / ms / 1e6,
                            bytes_per_cu / (ms * 1e-3) / 2.4e9, bytes_per_cu * 256 / (ms * 1e-3) / 1e12);
        }
    return 0;
}
