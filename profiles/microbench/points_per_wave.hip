// microbenchmark (round 3): the one lever left in the fp16-pair MLP kernel - MORE THAN 32 POINTS PER WEIGHT PASS.
//
// A step of nerf_mlp_h2_kernel is one output tile x one k-tile for the wave's 32 points: 4 ds_read_b128 of A fragments,
// one LDS-DMA piece of the weight stream, 6 v_mfma_f32_32x32x16_f16 and the conversion of one register pair of the
// pending layer (11 vector instructions); at 256 units a wave holds 128 accumulators + 128 operand registers per 32
// points, so a second column set does not fit. At 128 units it would (64 + 64 per set): the same fragments and the same
// LDS-DMA piece would feed 12 MFMAs, at twice the conversion work. This program measures exactly that step mix - NSET
// column sets per wave, everything else as in the kernel: one wave per SIMD, fragment reads one per MFMA gap from inline
// asm with counted waits, a barrier + vmcnt(8) every 8 steps, random fp16 operands (the chip's clock depends on the
// data) - and prints the time per step and per (step x column set).
//
//   hipcc --offload-arch=gfx950 -O3 points_per_wave.hip -o points_per_wave && ./points_per_wave
//   rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d out -- ./points_per_wave
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

__device__ __forceinline__ f32x16 mma(const f32x4& a, const f32x4& b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, a), __builtin_bit_cast(h16x8, b), c, 0, 0, 0);
}
template <int OFF>
__device__ __forceinline__ void frag(f32x4& q, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(q) : "v"(addr), "n"(OFF) : "memory");
}
// the conversion of one register pair: 2 fma, 2 max, 1 max3, 2 mul, cvt_pk, 2 fma_mix (what conv_slice0..2 issue)
__device__ __forceinline__ void convert(float& v0, float& v1, float& m, unsigned& hi, unsigned& lo) {
    const float y0 = fmaxf(__builtin_fmaf(v0, 1.0001f, v1), 0.f), y1 = fmaxf(__builtin_fmaf(v1, 0.9999f, v0), 0.f);
    m = fmaxf(fmaxf(m, y0), y1);
    const float a0 = y0 * 1.5f, a1 = y1 * 1.5f;
    asm volatile("v_cvt_pk_f16_f32 %0, %2, %3\n\tv_fma_mixlo_f16 %1, %0, -1.0, %2 op_sel_hi:[1,0,0]\n\ts_nop 0\n\t"
                 "v_fma_mixhi_f16 %1, %0, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
                 : "=&v"(hi), "=&v"(lo)
                 : "v"(a0), "v"(a1));
    v0 = y0 * 0.5f + 1e-3f;
    v1 = __uint_as_float((lo & 0xffffu) | 0x3f000000u) * 0.5f;
}
#define FENCE() __builtin_amdgcn_sched_barrier(0)

template <int NSET>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void step_kernel(const char* stream, const f32x4* operands, float* out, int steps) {
    extern __shared__ __attribute__((aligned(16))) char lds[];      // [0, 32 KiB): fragments; then four 8 KiB landing zones
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 32768 / 16; i += 256) ((f32x4*)lds)[i] = ((const f32x4*)stream)[i];
    __syncthreads();
    f32x16 acc[NSET];
    f32x4 bh[NSET], bl[NSET];      // a column set's B operand (hi, lo): random fp16 pairs
    float v0[NSET], v1[NSET], m[NSET];
    unsigned hi[NSET], lo[NSET];
#pragma unroll
    for (int k = 0; k < NSET; ++k) {
        for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
        bh[k] = operands[(2 * k) * 64 + lane];
        bl[k] = operands[(2 * k + 1) * 64 + lane];
        v0[k] = lane * 1e-3f + k;
        v1[k] = 0.5f;
        m[k] = 0.f;
        hi[k] = lo[k] = 0;
    }
    const unsigned fr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds + lane * 16;
    const char* g = stream + 32768 + wave * 8192 + lane * 16;
    char* l = lds + 32768 + wave * 8192;
    f32x4 q0, q1, q2, q3;
    frag<0>(q0, fr);
    frag<1024>(q1, fr);
    frag<2048>(q2, fr);
    frag<3072>(q3, fr);
    for (int s8 = 0; s8 < steps; s8 += 8) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            f32x4 n0, n1, n2, n3;
            const unsigned ad = fr + ((j + 1) & 7) * 4096;
            FENCE();
            asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
            FENCE();
#pragma unroll
            for (int k = 0; k < NSET; ++k) acc[k] = mma(q1, bh[k], acc[k]);       // W_lo x_hi
            FENCE();
            frag<0>(n0, ad);
            __builtin_amdgcn_global_load_lds(GLB_PTR(g + (j & 7) * 1024), LDS_PTR(l + (j & 7) * 1024), 16, 0, 0);
            FENCE();
#pragma unroll
            for (int k = 0; k < NSET; ++k) acc[k] = mma(q0, bl[k], acc[k]);       // W_hi x_lo
            FENCE();
#pragma unroll
            for (int k = 0; k < NSET; ++k) convert(v0[k], v1[k], m[k], hi[k], lo[k]);
            frag<1024>(n1, ad);
            FENCE();
#pragma unroll
            for (int k = 0; k < NSET; ++k) acc[k] = mma(q0, bh[k], acc[k]);       // W_hi x_hi
            FENCE();
            frag<2048>(n2, ad);
            FENCE();
            asm volatile("s_waitcnt lgkmcnt(3)" ::: "memory");
            FENCE();
#pragma unroll
            for (int k = 0; k < NSET; ++k) acc[k] = mma(q3, bh[k], acc[k]);
            FENCE();
            frag<3072>(n3, ad);
            FENCE();
#pragma unroll
            for (int k = 0; k < NSET; ++k) acc[k] = mma(q2, bl[k], acc[k]);
            FENCE();
#pragma unroll
            for (int k = 0; k < NSET; ++k) acc[k] = mma(q2, bh[k], acc[k]);
            FENCE();
            q0 = n0;
            q1 = n1;
            q2 = n2;
            q3 = n3;
            if (j == 3) {
                asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
                FENCE();
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < NSET; ++k) {
        sum += m[k] + __uint_as_float(hi[k]) * 0.f;
        for (int r = 0; r < 16; ++r) sum += acc[k][r];
    }
    out[blockIdx.x * 256 + threadIdx.x] = sum + q0[0] * 0.f;
}

template <int NSET>
double run(const char* stream, const f32x4* operands, float* out, int n_cu) {
    const int steps = 40000;
    auto fn = step_kernel<NSET>;
    const size_t lds = 32768 + 4 * 8192;
    (void)hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(fn, dim3(n_cu), dim3(256), lds, 0, stream, operands, out, 800);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(fn, dim3(n_cu), dim3(256), lds, 0, stream, operands, out, steps);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double ns = ms * 1e6 / steps;
    const double tflops = 6.0 * NSET * 2.0 * 32 * 32 * 16 * 4 * n_cu / (ns * 1e-9) / 1e12;
    printf("%d column set(s) of 32 points per wave: %8.3f ms  %6.1f ns per step  %6.1f ns per (step x set)  %7.1f TFLOP/s executed "
           "(%.1f %% of 2516.6)\n", NSET, ms, ns, ns / NSET, tflops, 100.0 * tflops / 2516.6);
    return ns / NSET;
}

int main() {
    int n_cu = 256;
    (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, 0);
    std::vector<_Float16> h((32768 + 4 * 8192 * 4) / 2 + 4 * 64 * 8);
    srand(7);
    for (auto& v : h) v = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 2.0f);      // random fp16 operands: realistic power
    char* stream = nullptr;
    f32x4* operands = nullptr;
    float* out = nullptr;
    (void)hipMalloc((void**)&stream, 32768 + 4 * 8192 * 4);
    (void)hipMalloc((void**)&operands, 4 * 64 * sizeof(f32x4));
    (void)hipMalloc((void**)&out, (size_t)n_cu * 256 * sizeof(float));
    (void)hipMemcpy(stream, h.data(), 32768 + 4 * 8192 * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(operands, h.data() + (32768 + 4 * 8192 * 4) / 2, 4 * 64 * sizeof(f32x4), hipMemcpyHostToDevice);
    for (int rep = 0; rep < 4; ++rep) {      // (the first repetition also carries the clock's ramp)
        const double one = run<1>(stream, operands, out, n_cu);
        const double two = run<2>(stream, operands, out, n_cu);
        printf("  -> a second column set per wave takes %.1f %% off the time per point\n", 100.0 * (1.0 - two / one));
    }
    return 0;
}
