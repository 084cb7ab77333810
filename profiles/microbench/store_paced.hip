// microbenchmark: do 1 KiB nt stores that arrive SPACED OUT (as the training kernels issue them: a piece every two MFMA steps)
// reach the bandwidth the same pattern reaches back to back (store_pattern.hip: 6.3 TB/s)? 256 workgroups x 4 waves, each wave
// writes its 32-point groups' pieces [tile T][quad Q] of ten layer buffers; between two stores it sleeps (s_sleep 64-cycle units).
// layout 0: blocked by 32 points as shipped ([group][piece 32][1 KiB]: a wave's pieces are contiguous, the four waves of a
//           workgroup 32 KiB apart); layout 1: the four waves' pieces interleaved ([tile of 128 points][piece 32][wave 4][1 KiB]).
// hipcc --offload-arch=gfx950 -O3 store_paced.hip -o store_paced
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int LAYOUT, int SLEEP>
__global__ __launch_bounds__(256) void k(float* buf, int n_points, int layers) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n_tiles = n_points / 128;
    const f32x4 v = {1.0f * lane, 2.0f, 3.0f, 4.0f};
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        for (int l = 0; l < layers; ++l) {
            char* base = (char*)buf + (size_t)l * n_points * 1024;
            for (int piece = 0; piece < 32; ++piece) {
                char* dst = LAYOUT == 0 ? base + ((size_t)tile * 4 + wave) * 32768 + piece * 1024 + lane * 16
                                        : base + (size_t)tile * 131072 + piece * 4096 + wave * 1024 + lane * 16;
                asm volatile("global_store_dwordx4 %0, %1, off nt" : : "v"(dst), "v"(v) : "memory");
                if (SLEEP > 0) __builtin_amdgcn_s_sleep(SLEEP);
            }
        }
    }
}

template <int LAYOUT, int SLEEP>
void run(float* buf, int P, int layers, size_t bytes, const char* name) {
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    for (int rep = 0; rep < 2; ++rep) {
        (void)hipEventRecord(a);
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k<LAYOUT, SLEEP>), dim3(256), dim3(256), 0, 0, buf, P, layers);
        (void)hipEventRecord(b);
        (void)hipEventSynchronize(b);
        float ms;
        (void)hipEventElapsedTime(&ms, a, b);
        if (rep) printf("%-34s sleep %2d: %8.1f us per pass of %.2f GB = %.2f TB/s\n", name, SLEEP, ms * 1000 / 3, bytes / 1e9, bytes / (ms / 3 * 1e-3) / 1e12);
    }
}

int main() {
    const int P = 196608, layers = 10;
    float* buf;
    const size_t bytes = (size_t)P * 1024 * layers;
    if (hipMalloc(&buf, bytes) != hipSuccess) return 1;
    run<0, 0>(buf, P, layers, bytes, "blocked by 32 points (shipped)");
    run<0, 2>(buf, P, layers, bytes, "blocked by 32 points (shipped)");
    run<0, 4>(buf, P, layers, bytes, "blocked by 32 points (shipped)");
    run<0, 6>(buf, P, layers, bytes, "blocked by 32 points (shipped)");
    run<0, 8>(buf, P, layers, bytes, "blocked by 32 points (shipped)");
    run<0, 12>(buf, P, layers, bytes, "blocked by 32 points (shipped)");
    run<1, 0>(buf, P, layers, bytes, "four waves' pieces interleaved");
    run<1, 2>(buf, P, layers, bytes, "four waves' pieces interleaved");
    run<1, 4>(buf, P, layers, bytes, "four waves' pieces interleaved");
    run<1, 6>(buf, P, layers, bytes, "four waves' pieces interleaved");
    run<1, 8>(buf, P, layers, bytes, "four waves' pieces interleaved");
    run<1, 12>(buf, P, layers, bytes, "four waves' pieces interleaved");
    return 0;
}
