// microbenchmark: what does the ADDRESS PATTERN of the training kernels' activation stores cost?
// The fused forward kernels keep every layer's activations for the backward pass: 196 608 points x 2 432 floats = 1.9 GB
// per fine pass, written 16 bytes per lane in the MFMA accumulator layout (lane = point p + 32 h, a register quad =
// features 8 q + 4 h .. + 3). Row-major [point][feature] (what autograd's buffers and the dW kernels use today) makes one
// store instruction touch 32 rows, 32 bytes each; a layout blocked by 32 points ([p/32][feature quad][p%32][4]) makes the
// same instruction write 1 KiB contiguous. No arithmetic here: 256 workgroups of 4 waves write the same volume either way.
// hipcc --offload-arch=gfx950 -O3 store_pattern.hip -o store_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int LAYOUT>   // 0: row-major, 16 B per lane; 1: blocked, 16 B per lane; 2: row-major, 8 B per lane (two instructions)
__global__ __launch_bounds__(256) void store_kernel(float* buf, int n_points, int ld, int layers) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5;
    const int n_tiles = n_points / 128;
    const f32x4 v = {1.0f * lane, 2.0f, 3.0f, 4.0f};
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const unsigned pt = tile * 128 + wave * 32 + (lane & 31);
        for (int l = 0; l < layers; ++l) {
            float* base = buf + (size_t)l * n_points * ld;
#pragma unroll
            for (int t = 0; t < 8; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (LAYOUT == 0) {
                        *(f32x4*)(base + pt * ld + 32 * t + 8 * q + 4 * h) = v;
                    } else if (LAYOUT == 1) {
                        *(f32x4*)(base + (pt >> 5) * (32 * ld) + (8 * t + 2 * q + h) * 128 + (pt & 31) * 4) = v;
                    } else {
                        float* p = base + pt * ld + 32 * t + 8 * q + 4 * h;
                        *(float2*)p = float2{v[0], v[1]};
                        *(float2*)(p + 2) = float2{v[2], v[3]};
                    }
                }
        }
    }
}

int main() {
    const int P = 196608, ld = 256, layers = 9;
    float* buf;
    const size_t bytes = (size_t)P * ld * layers * 4;
    if (hipMalloc(&buf, bytes) != hipSuccess) return 1;
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    const char* names[3] = {"row-major 16 B/lane", "blocked   16 B/lane", "row-major  8 B/lane"};
    for (int rep = 0; rep < 2; ++rep)
        for (int v = 0; v < 3; ++v) {
            hipEventRecord(a);
            for (int i = 0; i < 5; ++i) {
                if (v == 0) hipLaunchKernelGGL(store_kernel<0>, dim3(256), dim3(256), 0, 0, buf, P, ld, layers);
                if (v == 1) hipLaunchKernelGGL(store_kernel<1>, dim3(256), dim3(256), 0, 0, buf, P, ld, layers);
                if (v == 2) hipLaunchKernelGGL(store_kernel<2>, dim3(256), dim3(256), 0, 0, buf, P, ld, layers);
            }
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms;
            hipEventElapsedTime(&ms, a, b);
            if (rep) printf("%s: %.1f us per pass of %.2f GB = %.2f TB/s\n", names[v], ms * 200, bytes / 1e9, bytes / (ms / 5 * 1e-3) / 1e12);
        }
    hipMemset(buf, 0, bytes);
    hipEventRecord(a);
    for (int i = 0; i < 5; ++i) hipMemsetAsync(buf, 0, bytes, 0);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    printf("hipMemset          : %.1f us = %.2f TB/s\n", ms * 200, bytes / (ms / 5 * 1e-3) / 1e12);
    return 0;
}
