// microbenchmark: fp32 GEMM emulated with three bf16 pieces per operand on v_mfma_f32_32x32x16_bf16.
//   (1) issue rate of the bare bf16 MFMA loop (clock under matrix load),
//   (2) accuracy of C[32x32] = A[32xK] * B[Kx32], K = 256, against an fp64 host result for
//       a) the v_mfma_f32_32x32x2_f32 chain the renderer uses today,
//       b) 6 bf16 products (a0b0 a0b1 a1b0 a0b2 a1b1 a2b0) into one accumulator,
//       c) the same with the four small products kept in a second accumulator,
//       d) 3 products (a0b0 a0b1 a1b0) for scale.
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off mfma_bf16_split.hip -o mfma_bf16_split
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __fp16 h16x2 __attribute__((ext_vector_type(2)));

// two fp16 pieces, round-toward-zero head, exact remainder rounded toward zero again
__device__ inline void split2h(float a, _Float16& p0, _Float16& p1) {
    h16x2 h = __builtin_amdgcn_cvt_pkrtz(a, 0.f);
    float r = a - (float)h[0];
    h16x2 l = __builtin_amdgcn_cvt_pkrtz(r, 0.f);
    p0 = (_Float16)h[0];
    p1 = (_Float16)l[0];
}

__device__ inline void split3(float a, unsigned short& p0, unsigned short& p1, unsigned short& p2) {
    unsigned u = __float_as_uint(a);
    float a0 = __uint_as_float(u & 0xffff0000u);
    float r1 = a - a0;
    unsigned u1 = __float_as_uint(r1);
    float a1 = __uint_as_float(u1 & 0xffff0000u);
    float r2 = r1 - a1;
    p0 = u >> 16;
    p1 = u1 >> 16;
    p2 = __float_as_uint(r2) >> 16;
}

__device__ inline f32x16 mma(s16x8 a, s16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void rate(float* out, int iters) {
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = (float)(threadIdx.x + r);
    s16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (short)(0x3f80 + threadIdx.x + j); b[j] = (short)(0x3f00 + threadIdx.x * 3 + j); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 64; ++u) acc = mma(a, b, acc);
    }
    float s = 0;
    for (int r = 0; r < 16; ++r) s += acc[r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// one wave; A row-major [32][K], B row-major [K][32], C row-major [32][32]; mode selects the scheme
__global__ __launch_bounds__(64) void gemm(const float* A, const float* B, float* C, int K, int mode, float wscale) {
    const int l = threadIdx.x, r = l & 31, h = l >> 5;
    f32x16 acc, lo;
    for (int i = 0; i < 16; ++i) { acc[i] = 0.f; lo[i] = 0.f; }
    if (mode == 0) {
        for (int k = 0; k < K; k += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[r * K + k + h], B[(k + h) * 32 + r], acc, 0, 0, 0);
    } else if (mode >= 5) {
        // fp16 pairs: weights pre-scaled by a power of two (host side in the real thing), 3 products
        for (int s = 0; s < K; s += 16) {
            f16x8 a0, a1, b0, b1;
            for (int j = 0; j < 8; ++j) {
                _Float16 p0, p1;
                split2h(A[r * K + s + 8 * h + j] * wscale, p0, p1);
                a0[j] = p0; a1[j] = p1;
                split2h(B[(s + 8 * h + j) * 32 + r], p0, p1);
                b0[j] = p0; b1[j] = p1;
            }
            if (mode == 5) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, acc, 0, 0, 0);
            } else {   // mode 6: all four products
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, acc, 0, 0, 0);
            }
        }
        for (int i = 0; i < 16; ++i) acc[i] *= 1.0f / wscale;
    } else {
        for (int s = 0; s < K; s += 16) {
            s16x8 a0, a1, a2, b0, b1, b2;
            for (int j = 0; j < 8; ++j) {
                unsigned short p0, p1, p2;
                split3(A[r * K + s + 8 * h + j], p0, p1, p2);
                a0[j] = p0; a1[j] = p1; a2[j] = p2;
                split3(B[(s + 8 * h + j) * 32 + r], p0, p1, p2);
                b0[j] = p0; b1[j] = p1; b2[j] = p2;
            }
            if (mode == 1) {
                acc = mma(a2, b0, acc); acc = mma(a1, b1, acc); acc = mma(a0, b2, acc);
                acc = mma(a1, b0, acc); acc = mma(a0, b1, acc); acc = mma(a0, b0, acc);
            } else if (mode == 2) {
                lo = mma(a2, b0, lo); lo = mma(a1, b1, lo); lo = mma(a0, b2, lo);
                lo = mma(a1, b0, lo); lo = mma(a0, b1, lo); acc = mma(a0, b0, acc);
            } else if (mode == 3) {
                acc = mma(a1, b0, acc); acc = mma(a0, b1, acc); acc = mma(a0, b0, acc);
            } else {   // mode 4: 6 products, large first (order sensitivity)
                acc = mma(a0, b0, acc); acc = mma(a0, b1, acc); acc = mma(a1, b0, acc);
                acc = mma(a0, b2, acc); acc = mma(a1, b1, acc); acc = mma(a2, b0, acc);
            }
        }
        if (mode == 2) for (int i = 0; i < 16; ++i) acc[i] += lo[i];
    }
    for (int i = 0; i < 16; ++i) C[((i & 3) + 8 * (i >> 2) + 4 * h) * 32 + r] = acc[i];
}

int main() {
    {
        float* out; (void)hipMalloc(&out, 256 * 256 * 4);
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        for (int rep = 0; rep < 3; ++rep) {
            const int iters = 4000;
            hipLaunchKernelGGL(rate, dim3(256), dim3(256), 0, 0, out, 10);
            (void)hipDeviceSynchronize();
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(rate, dim3(256), dim3(256), 0, 0, out, iters);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            double mfma = (double)iters * 64;
            double flops = mfma * 32 * 32 * 16 * 2 * 1024;
            printf("bf16 32x32x16 bare loop: %.3f ms, %.1f TFLOP/s, %.2f ns per MFMA (32 cycles -> %.3f GHz)\n", ms, flops / ms / 1e9,
                   ms * 1e6 / mfma, 32.0 / (ms * 1e6 / mfma));
        }
        (void)hipFree(out);
    }
    const int K = 256;
    for (int dist = 0; dist < 5; ++dist) {
        std::vector<float> A(32 * K), B(K * 32), C(32 * 32);
        srand(1234 + dist);
        auto rnd = [&]() { return (float)rand() / RAND_MAX * 2.f - 1.f; };
        // dist 2: dist 1 with activations 2^-8 smaller; dist 3: 2^-14 smaller; dist 4: sin/cos-like inputs in [-1,1]
        for (auto& v : A) v = dist == 0 ? rnd() * 0.1f : rnd() * rnd() * rnd() * 3.f;           // weights
        for (auto& v : B) v = dist == 0 ? fabsf(rnd()) * 2.f : (rnd() > 0 ? fabsf(rnd()) * 5.f : 0.f);   // post-ReLU activations
        if (dist == 2) for (auto& v : B) v *= 1.f / 256.f;
        if (dist == 3) for (auto& v : B) v *= 1.f / 16384.f;
        if (dist == 4) for (auto& v : B) v = sinf(1000.f * rnd());
        float wmax = 0;
        for (auto& v : A) wmax = fmaxf(wmax, fabsf(v));
        int e; frexpf(wmax, &e);
        const float wscale = ldexpf(1.f, 13 - e);
        std::vector<double> ref(32 * 32), scale(32 * 32);
        for (int i = 0; i < 32; ++i)
            for (int j = 0; j < 32; ++j) {
                double s = 0, t = 0;
                for (int k = 0; k < K; ++k) { s += (double)A[i * K + k] * B[k * 32 + j]; t += fabs((double)A[i * K + k] * B[k * 32 + j]); }
                ref[i * 32 + j] = s; scale[i * 32 + j] = t;
            }
        float *dA, *dB, *dC;
        (void)hipMalloc(&dA, A.size() * 4); (void)hipMalloc(&dB, B.size() * 4); (void)hipMalloc(&dC, C.size() * 4);
        (void)hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
        (void)hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
        const char* names[] = {"fp32 mfma 32x32x2 chain", "bf16x3, 6 products, small first", "bf16x3, 6 products, low accumulator",
                               "bf16x2-ish, 3 products", "bf16x3, 6 products, large first", "fp16x2 (scaled weights), 3 products",
                               "fp16x2 (scaled weights), 4 products"};
        for (int mode = 0; mode < 7; ++mode) {
            hipLaunchKernelGGL(gemm, dim3(1), dim3(64), 0, 0, dA, dB, dC, K, mode, wscale);
            (void)hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
            double mx = 0, rms = 0;
            for (int i = 0; i < 1024; ++i) { double e = fabs(C[i] - ref[i]) / scale[i]; mx = fmax(mx, e); rms += e * e; }
            printf("dist %d  %-38s max |err|/sum|ab| = %.3e (%.2f eps)  rms = %.3e (%.2f eps)\n", dist, names[mode], mx, mx / 5.96e-8,
                   sqrt(rms / 1024), sqrt(rms / 1024) / 5.96e-8);
        }
        (void)hipFree(dA); (void)hipFree(dB); (void)hipFree(dC);
    }
    {   // are fp16 subnormal operands honoured by the matrix pipe?  A = 2^-20 everywhere, B = 1: C = K * 2^-20 if so
        std::vector<float> A(32 * K, ldexpf(1.f, -20)), B(K * 32, 1.f), C(32 * 32);
        float *dA, *dB, *dC;
        (void)hipMalloc(&dA, A.size() * 4); (void)hipMalloc(&dB, B.size() * 4); (void)hipMalloc(&dC, C.size() * 4);
        (void)hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
        (void)hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(gemm, dim3(1), dim3(64), 0, 0, dA, dB, dC, K, 5, 1.0f);
        (void)hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
        printf("fp16 subnormal operand test: C[0] = %.9g, expected %.9g\n", C[0], K * ldexp(1.0, -20));
    }
    return 0;
}
