// microbenchmark: what does ONE STEP of the fp16-pair MLP kernel cost at best?
// A step = 6 dependent v_mfma_f32_32x32x16_f16 on one accumulator (192 matrix-pipe cycles) plus, on the same
// wave (one wave per SIMD, four per CU): 4 ds_read_b128 of A-fragments, 1 ds_read_b64, one LDS-DMA piece
// (global_load_lds_dwordx4, 1 KiB per wave) and 12 vector instructions of the conversion chain.
// Each variant adds one ingredient to a bare MFMA loop; cycles per step = kernel time * clock / steps.
// hipcc --offload-arch=gfx950 -O3 step_mix.hip -o step_mix
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

template <bool READS, int DMA, bool VALU, int AHEAD = 1, int NREADS = 4, bool BARRIER = true, int SPLIT = 0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void step_kernel(const char* stream, float* out, int steps) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 32768 / 4; i += 256) ((float*)lds)[i] = 1e-3f * (i & 255);
    __syncthreads();
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    f32x4 q0 = {1e-3f, 2e-3f, 3e-3f, 4e-3f}, q1 = q0, q2 = q0, q3 = q0;
    f32x4 b = {0.5f, 0.25f, 0.125f, 1.f};
    f32x4 p0 = q0, p1 = q0, p2 = q0, p3 = q0;
    float v0 = lane * 1e-3f, v1 = 0.5f, m = 0.f;
    unsigned hi = 0, lo = 0;
    const f32x4* fr = (const f32x4*)lds + lane;
    const char* g = stream + wave * 8192 + lane * 16;
    char* l = lds + 32768 + wave * 8192;
    f32x4 stage[4] = {q0, q0, q0, q0};   // DMA == 2: pieces on their way through (accumulator-file) registers
    for (int s4 = 0; s4 < steps; s4 += 4) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int s = s4 + j;
        const int grp = (s & 7) * 4;
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, q1), __builtin_bit_cast(h16x8, b), acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        f32x4 n0 = q0, n1 = q1, n2 = q2, n3 = q3;
        if (READS) {
            n0 = fr[(grp + 0) * 64];
            if (NREADS > 1 && SPLIT < 2) n1 = fr[(grp + 1) * 64];
            if (NREADS > 2 && SPLIT == 0) n2 = fr[(grp + 2) * 64];
            if (NREADS > 3 && SPLIT == 0) n3 = fr[(grp + 3) * 64];
        }
        if (DMA == 1) __builtin_amdgcn_global_load_lds(GLB_PTR(g + (s & 3) * 1024), LDS_PTR(l + (s & 3) * 1024), 16, 0, 0);
        if (DMA == 2) {
            // the piece requested three steps ago goes to LDS, then this step's request takes its registers
            const unsigned la = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)(l + j * 1024) + lane * 16;
            asm volatile("s_waitcnt vmcnt(3)\n\tds_write_b128 %1, %0" : "+a"(stage[j]) : "v"(la) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, off" : "=a"(stage[j]) : "v"(g + j * 1024), "0"(stage[j]) : "memory");
        }
        __builtin_amdgcn_sched_barrier(0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, q0), __builtin_bit_cast(h16x8, b), acc, 0, 0, 0);
        if (READS && SPLIT == 2) { __builtin_amdgcn_sched_barrier(0); n1 = fr[(grp + 1) * 64]; __builtin_amdgcn_sched_barrier(0); }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, q0), __builtin_bit_cast(h16x8, b), acc, 0, 0, 0);
        if (READS && SPLIT == 2) { __builtin_amdgcn_sched_barrier(0); n2 = fr[(grp + 2) * 64]; __builtin_amdgcn_sched_barrier(0); }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, q3), __builtin_bit_cast(h16x8, b), acc, 0, 0, 0);
        if (READS && SPLIT == 1) { __builtin_amdgcn_sched_barrier(0); n2 = fr[(grp + 2) * 64]; n3 = fr[(grp + 3) * 64]; __builtin_amdgcn_sched_barrier(0); }
        if (READS && SPLIT == 2) { __builtin_amdgcn_sched_barrier(0); n3 = fr[(grp + 3) * 64]; __builtin_amdgcn_sched_barrier(0); }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, q2), __builtin_bit_cast(h16x8, b), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, q2), __builtin_bit_cast(h16x8, b), acc, 0, 0, 0);
        if (VALU) {
            // the conversion chain of one register pair: 2 fma, 2 max, 1 max3, 2 mul, cvt_pk, 2 fma_mix (+ 2 moves)
            float y0 = fmaxf(__builtin_fmaf(v0, 1.0001f, v1), 0.f), y1 = fmaxf(__builtin_fmaf(v1, 0.9999f, v0), 0.f);
            m = fmaxf(fmaxf(m, y0), y1);
            const float a0 = y0 * 1.5f, a1 = y1 * 1.5f;
            asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hi) : "v"(a0), "v"(a1));
            asm volatile("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(lo) : "v"(hi), "v"(a0));
            asm volatile("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lo) : "v"(hi), "v"(a1));
            v0 = y0 * 0.5f + 1e-3f;
            v1 = __uint_as_float((lo & 0xffffu) | 0x3f000000u) * 0.5f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (AHEAD == 2) {   // one more step of distance between a read and its use
            q0 = p0; q1 = p1; q2 = p2; q3 = p3;
            p0 = n0; p1 = n1; p2 = n2; p3 = n3;
        } else {
            q0 = n0; q1 = n1; q2 = n2; q3 = n3;
        }
        if (BARRIER && (s & 7) == 3) {
            if (DMA == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            asm volatile("s_barrier" ::: "memory");
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float sum = m + __uint_as_float(hi) * 0.f + stage[0][0] * 0.f;
    for (int r = 0; r < 16; ++r) sum += acc[r];
    out[blockIdx.x * 256 + threadIdx.x] = sum;
}


// ---- alternative structure: TWO waves per SIMD (8 per CU, 256 registers each), 16 points per wave, 16x16x32 MFMAs ----
// Same work per SIMD and per "step" as step_kernel (6 x 16x16x32 per wave, two waves = 6 x 32x32x16), same bytes of
// A-fragments per MAC x2 (a wave's fragments feed 16 points instead of 32): per 6-MFMA step 4 ds_read_b128, the
// conversion chain of one register pair every second step, one LDS-DMA piece every second step, barrier every 8 steps.
typedef float f32x4v __attribute__((ext_vector_type(4)));
template <bool READS, bool DMA, bool VALU, bool BARRIER = true, int CHAINS = 2>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
void step16_kernel(const char* stream, float* out, int steps) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 32768 / 4; i += 512) ((float*)lds)[i] = 1e-3f * (i & 255);
    __syncthreads();
    f32x4v acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
    f32x4 q0 = {1e-3f, 2e-3f, 3e-3f, 4e-3f}, q1 = q0, q2 = q0, q3 = q0;
    f32x4 b = {0.5f, 0.25f, 0.125f, 1.f};
    float v0 = lane * 1e-3f, v1 = 0.5f, m = 0.f;
    unsigned hi = 0, lo = 0;
    const f32x4* fr = (const f32x4*)lds + lane;
    const char* g = stream + wave * 4096 + lane * 16;
    char* l = lds + 32768 + wave * 4096;
#define MMA16(A, ACC) ACC = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h16x8, A), __builtin_bit_cast(h16x8, b), ACC, 0, 0, 0)
    for (int s2 = 0; s2 < steps; s2 += 2) {
#pragma unroll
      for (int odd = 0; odd < 2; ++odd) {
        const int s = s2 + odd;
        const int grp = (s & 7) * 4;
        // out tile a: lo*x_hi first
        MMA16(q1, acc0);
        __builtin_amdgcn_sched_barrier(0);
        f32x4 n0 = q0, n1 = q1, n2 = q2, n3 = q3;
        if (READS) {   // from inline asm: hipcc otherwise guards its own LDS reads against the LDS-DMA in flight
            const unsigned ad = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) f32x4*)(fr + grp * 64);
            asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:1024\n\tds_read_b128 %2, %4 offset:2048\n\t"
                         "ds_read_b128 %3, %4 offset:3072" : "=&v"(n0), "=&v"(n1), "=&v"(n2), "=&v"(n3) : "v"(ad) : "memory");
        }
        if (DMA && odd) __builtin_amdgcn_global_load_lds(GLB_PTR(g + ((s >> 1) & 3) * 1024), LDS_PTR(l + ((s >> 1) & 3) * 1024), 16, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (CHAINS == 2) {
            MMA16(q3, acc1);
            MMA16(q0, acc0);
            MMA16(q2, acc1);
            MMA16(q0, acc0);
            MMA16(q2, acc1);
        } else {
            MMA16(q0, acc0);
            MMA16(q0, acc0);
            MMA16(q3, acc1);
            MMA16(q2, acc1);
            MMA16(q2, acc1);
        }
        if (VALU && !odd) {
            float y0 = fmaxf(__builtin_fmaf(v0, 1.0001f, v1), 0.f), y1 = fmaxf(__builtin_fmaf(v1, 0.9999f, v0), 0.f);
            m = fmaxf(fmaxf(m, y0), y1);
            const float a0 = y0 * 1.5f, a1 = y1 * 1.5f;
            asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hi) : "v"(a0), "v"(a1));
            asm volatile("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(lo) : "v"(hi), "v"(a0));
            asm volatile("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lo) : "v"(hi), "v"(a1));
            v0 = y0 * 0.5f + 1e-3f;
            v1 = __uint_as_float((lo & 0xffffu) | 0x3f000000u) * 0.5f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (READS) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(n0), "+v"(n1), "+v"(n2), "+v"(n3)::"memory");
        q0 = n0; q1 = n1; q2 = n2; q3 = n3;
        if (BARRIER && odd && (s & 7) == 3) {
            if (DMA) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            asm volatile("s_barrier" ::: "memory");
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float sum = m + __uint_as_float(hi) * 0.f;
    for (int r = 0; r < 4; ++r) sum += acc0[r] + acc1[r];
    out[blockIdx.x * 512 + threadIdx.x] = sum;
}

template <bool READS, bool DMA, bool VALU, bool BARRIER = true, int CHAINS = 2>
void run16(const char* name, const char* stream, float* out) {
    const int steps = 40000;
    auto fn = step16_kernel<READS, DMA, VALU, BARRIER, CHAINS>;
    (void)hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, 98304 + 4096);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(fn, dim3(256), dim3(512), 98304 + 4096, 0, stream, out, 100);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(fn, dim3(256), dim3(512), 98304 + 4096, 0, stream, out, steps);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double ns = ms * 1e6 / steps;
    printf("%-60s %7.3f ms  %6.1f ns per (2-wave) step, same MACs per SIMD as a 32x32x16 step\n", name, ms, ns);
}

// ---- MFMA shape at equal work: one wave per SIMD, 32 points per wave, the step's four A-fragments (two 16-row output
// tiles x [hi|lo] of a 32-deep k-tile) against two 16-point column groups: 12 v_mfma_f32_16x16x32_f16 per step instead
// of 6 v_mfma_f32_32x32x16_f16; same LDS bytes, same MACs. The clock the chip holds can depend on the shape
// (MI355X_MICROARCH.md, DVFS give-back item 7), so operands are random.
template <bool READS, int DMA, bool VALU>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void step_shape16_kernel(const char* stream, float* out, int steps, const f32x4* rnd) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 32768 / 16; i += 256) ((f32x4*)lds)[i] = rnd[i];
    __syncthreads();
    f32x4v a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, c0 = a0, c1 = a0;
    f32x4 q0 = rnd[lane], q1 = rnd[64 + lane], q2 = rnd[128 + lane], q3 = rnd[192 + lane];
    const f32x4 bh0 = rnd[256 + threadIdx.x], bl0 = rnd[512 + threadIdx.x], bh1 = rnd[768 + threadIdx.x], bl1 = rnd[1024 + threadIdx.x];
    float v0 = lane * 1e-3f, v1 = 0.5f, m = 0.f;
    unsigned hi = 0, lo = 0;
    f32x4 stage[4] = {q0, q0, q0, q0};
    const bool odd_wave = __builtin_amdgcn_readfirstlane(wave & 1) != 0;
    const f32x4* fr = (const f32x4*)lds + lane;
    const char* g = stream + wave * 8192 + lane * 16;
    char* l = lds + 32768 + wave * 8192;
#define MMA16B(A, B, ACC) ACC = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h16x8, A), __builtin_bit_cast(h16x8, B), ACC, 0, 0, 0)
    for (int s4 = 0; s4 < steps; s4 += 4) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int s = s4 + j;
        const int grp = (s & 7) * 4;
        MMA16B(q1, bh0, a0);
        __builtin_amdgcn_sched_barrier(0);
        f32x4 n0 = q0, n1 = q1, n2 = q2, n3 = q3;
        if (READS) {
            const unsigned ad = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) f32x4*)(fr + grp * 64);
            asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:1024\n\tds_read_b128 %2, %4 offset:2048\n\t"
                         "ds_read_b128 %3, %4 offset:3072" : "=&v"(n0), "=&v"(n1), "=&v"(n2), "=&v"(n3) : "v"(ad) : "memory");
        }
        if (DMA == 1 || (DMA == 5 && !odd_wave)) __builtin_amdgcn_global_load_lds(GLB_PTR(g + (s & 3) * 1024), LDS_PTR(l + (s & 3) * 1024), 16, 0, 0);
        if (DMA == 2) {
            const unsigned la = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)(l + j * 1024) + lane * 16;
            asm volatile("s_waitcnt vmcnt(3)\n\tds_write_b128 %1, %0" : "+a"(stage[j]) : "v"(la) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, off" : "=a"(stage[j]) : "v"(g + j * 1024), "0"(stage[j]) : "memory");
        }
        __builtin_amdgcn_sched_barrier(0);
        MMA16B(q1, bh1, a1);
        MMA16B(q0, bl0, a0);
        MMA16B(q0, bl1, a1);
        MMA16B(q0, bh0, a0);
        MMA16B(q0, bh1, a1);
        MMA16B(q3, bh0, c0);
        if (DMA == 5 && odd_wave) { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_global_load_lds(GLB_PTR(g + (s & 3) * 1024), LDS_PTR(l + (s & 3) * 1024), 16, 0, 0); __builtin_amdgcn_sched_barrier(0); }
        MMA16B(q3, bh1, c1);
        MMA16B(q2, bl0, c0);
        MMA16B(q2, bl1, c1);
        MMA16B(q2, bh0, c0);
        MMA16B(q2, bh1, c1);
        if (VALU) {
            float y0 = fmaxf(__builtin_fmaf(v0, 1.0001f, v1), 0.f), y1 = fmaxf(__builtin_fmaf(v1, 0.9999f, v0), 0.f);
            m = fmaxf(fmaxf(m, y0), y1);
            const float e0 = y0 * 1.5f, e1 = y1 * 1.5f;
            asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hi) : "v"(e0), "v"(e1));
            asm volatile("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(lo) : "v"(hi), "v"(e0));
            asm volatile("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lo) : "v"(hi), "v"(e1));
            v0 = y0 * 0.5f + 1e-3f;
            v1 = __uint_as_float((lo & 0xffffu) | 0x3f000000u) * 0.5f;
#pragma unroll
            for (int i = 0; i < 10; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (READS) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(n0), "+v"(n1), "+v"(n2), "+v"(n3)::"memory");
        q0 = n0; q1 = n1; q2 = n2; q3 = n3;
        if ((s & 7) == 3) {
            if (DMA == 1 || DMA == 5) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            asm volatile("s_barrier" ::: "memory");
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float sum = m + __uint_as_float(hi) * 0.f;
    for (int r = 0; r < 4; ++r) sum += a0[r] + a1[r] + c0[r] + c1[r] + stage[r][0] * 0.f;
    out[blockIdx.x * 256 + threadIdx.x] = sum;
}

// the 32x32x16 step with the same random operands and the same inline-asm reads, for a like-for-like comparison
template <bool READS, int DMA, bool VALU>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void step_shape32_kernel(const char* stream, float* out, int steps, const f32x4* rnd) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 32768 / 16; i += 256) ((f32x4*)lds)[i] = rnd[i];
    __syncthreads();
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    f32x4 q0 = rnd[lane], q1 = rnd[64 + lane], q2 = rnd[128 + lane], q3 = rnd[192 + lane];
    const f32x4 bh0 = rnd[256 + threadIdx.x], bl0 = rnd[512 + threadIdx.x], bh1 = rnd[768 + threadIdx.x], bl1 = rnd[1024 + threadIdx.x];
    float v0 = lane * 1e-3f, v1 = 0.5f, m = 0.f;
    unsigned hi = 0, lo = 0;
    f32x4 stage[4] = {q0, q0, q0, q0};
    const bool odd_wave = __builtin_amdgcn_readfirstlane(wave & 1) != 0;
    const f32x4* fr = (const f32x4*)lds + lane;
    const char* g = stream + wave * 8192 + lane * 16;
    char* l = lds + 32768 + wave * 8192;
#define MMA32B(A, B) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, A), __builtin_bit_cast(h16x8, B), acc, 0, 0, 0)
    for (int s4 = 0; s4 < steps; s4 += 4) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int s = s4 + j;
        const int grp = (s & 7) * 4;
        MMA32B(q1, bh0);
        __builtin_amdgcn_sched_barrier(0);
        f32x4 n0 = q0, n1 = q1, n2 = q2, n3 = q3;
        if (READS) {
            const unsigned ad = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) f32x4*)(fr + grp * 64);
            asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:1024\n\tds_read_b128 %2, %4 offset:2048\n\t"
                         "ds_read_b128 %3, %4 offset:3072" : "=&v"(n0), "=&v"(n1), "=&v"(n2), "=&v"(n3) : "v"(ad) : "memory");
        }
        if (DMA == 1 || (DMA == 5 && !odd_wave)) __builtin_amdgcn_global_load_lds(GLB_PTR(g + (s & 3) * 1024), LDS_PTR(l + (s & 3) * 1024), 16, 0, 0);
        if (DMA == 2) {
            const unsigned la = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)(l + j * 1024) + lane * 16;
            asm volatile("s_waitcnt vmcnt(3)\n\tds_write_b128 %1, %0" : "+a"(stage[j]) : "v"(la) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, off" : "=a"(stage[j]) : "v"(g + j * 1024), "0"(stage[j]) : "memory");
        }
        if (DMA == 3) {   // scalar base + 32-bit lane offset instead of a 64-bit address per lane
            const unsigned voff = wave * 8192 + lane * 16 + j * 1024;
            const unsigned la = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)(l + j * 1024));
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(stream), "s"(la) : "memory");
        }
        if (DMA == 4) {   // buffer form: descriptor + 32-bit lane offset
            const unsigned voff = wave * 8192 + lane * 16 + j * 1024;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(__builtin_amdgcn_make_buffer_rsrc((void*)stream, 0, 1 << 20, 0x00020000),
                                                     LDS_PTR(l + j * 1024), 16, voff, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        MMA32B(q0, bl0);
        MMA32B(q0, bh0);
        MMA32B(q3, bh1);
        if (DMA == 5 && odd_wave) { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_global_load_lds(GLB_PTR(g + (s & 3) * 1024), LDS_PTR(l + (s & 3) * 1024), 16, 0, 0); __builtin_amdgcn_sched_barrier(0); }
        MMA32B(q2, bl1);
        MMA32B(q2, bh1);
        if (VALU) {
            float y0 = fmaxf(__builtin_fmaf(v0, 1.0001f, v1), 0.f), y1 = fmaxf(__builtin_fmaf(v1, 0.9999f, v0), 0.f);
            m = fmaxf(fmaxf(m, y0), y1);
            const float e0 = y0 * 1.5f, e1 = y1 * 1.5f;
            asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hi) : "v"(e0), "v"(e1));
            asm volatile("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(lo) : "v"(hi), "v"(e0));
            asm volatile("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lo) : "v"(hi), "v"(e1));
            v0 = y0 * 0.5f + 1e-3f;
            v1 = __uint_as_float((lo & 0xffffu) | 0x3f000000u) * 0.5f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (READS) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(n0), "+v"(n1), "+v"(n2), "+v"(n3)::"memory");
        q0 = n0; q1 = n1; q2 = n2; q3 = n3;
        if ((s & 7) == 3) {
            if (DMA == 1 || DMA >= 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            asm volatile("s_barrier" ::: "memory");
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float sum = m + __uint_as_float(hi) * 0.f + stage[0][0] * 0.f + stage[1][0] * 0.f + stage[2][0] * 0.f + stage[3][0] * 0.f;
    for (int r = 0; r < 16; ++r) sum += acc[r];
    out[blockIdx.x * 256 + threadIdx.x] = sum;
}

template <class F>
void run_shape(const char* name, F fn, const char* stream, float* out, const f32x4* rnd) {
    const int steps = 400000;   // ~40 ms: long enough for the clock to settle
    (void)hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, 98304 + 4096);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(fn, dim3(256), dim3(256), 98304 + 4096, 0, stream, out, steps, rnd);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(fn, dim3(256), dim3(256), 98304 + 4096, 0, stream, out, steps, rnd);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-60s %8.3f ms  %6.1f ns per step\n", name, ms, ms * 1e6 / steps);
}

template <bool READS, int DMA, bool VALU, int AHEAD = 1, int NREADS = 4, bool BARRIER = true, int SPLIT = 0>
void run(const char* name, const char* stream, float* out) {
    const int steps = 40000;
    auto fn = step_kernel<READS, DMA, VALU, AHEAD, NREADS, BARRIER, SPLIT>;
    (void)hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, 98304 + 4096);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(fn, dim3(256), dim3(256), 98304 + 4096, 0, stream, out, 100);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(fn, dim3(256), dim3(256), 98304 + 4096, 0, stream, out, steps);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double ns = ms * 1e6 / steps;
    printf("%-44s %7.3f ms  %6.1f ns per step = %5.0f cycles @1.9 GHz  (MFMA-bound: 192)\n", name, ms, ns, ns * 1.9);
}

int main() {
    char* stream; float* out;
    (void)hipMalloc(&stream, 1 << 20); (void)hipMemset(stream, 0, 1 << 20);
    (void)hipMalloc(&out, 256 * 512 * 4);
    run<false, false, false>("6 MFMA", stream, out);
    run<true, false, false>("6 MFMA + 4 ds_read_b128", stream, out);
    run<true, true, false>("6 MFMA + reads + 1 LDS-DMA piece", stream, out);
    run<true, false, true>("6 MFMA + reads + 12 VALU", stream, out);
    run<true, true, true>("6 MFMA + reads + LDS-DMA piece + 12 VALU", stream, out);
    run<false, true, false>("6 MFMA + 1 LDS-DMA piece", stream, out);
    run<false, false, true>("6 MFMA + 12 VALU", stream, out);
    run<true, false, false, 1, 4, true, 1>("6 MFMA + reads split 2 + 2", stream, out);
    run<true, false, false, 1, 4, true, 2>("6 MFMA + reads split 1 + 1 + 1 + 1", stream, out);
    run<true, true, true, 1, 4, true, 1>("all, reads split 2 + 2", stream, out);
    run<true, true, true, 1, 4, true, 2>("all, reads split 1 + 1 + 1 + 1", stream, out);
    run<true, false, false, 2>("6 MFMA + 4 reads, two steps ahead", stream, out);
    run<true, true, true, 2>("all, reads two steps ahead", stream, out);
    run<true, false, false, 1, 2>("6 MFMA + 2 reads", stream, out);
    run<true, false, false, 1, 1>("6 MFMA + 1 read", stream, out);
    run<true, false, false, 1, 4, false>("6 MFMA + 4 reads, no barrier", stream, out);
    run<true, false, false, 2, 4, false>("6 MFMA + 4 reads, two ahead, no barrier", stream, out);
    printf("--- two waves per SIMD, 16 points per wave, v_mfma_f32_16x16x32_f16 ---\n");
    run16<false, false, false>("2w: 6 MFMA16 (two chains)", stream, out);
    run16<false, false, false, true, 1>("2w: 6 MFMA16 (3 + 3 dependent)", stream, out);
    run16<true, false, false>("2w: + 4 ds_read_b128", stream, out);
    run16<true, true, false>("2w: + reads + LDS-DMA piece every 2nd step", stream, out);
    run16<true, false, true>("2w: + reads + 10 VALU every 2nd step", stream, out);
    run16<true, true, true>("2w: everything", stream, out);
    run16<true, true, true, true, 1>("2w: everything (3 + 3 dependent)", stream, out);
    run16<true, true, true, false>("2w: everything, no barrier", stream, out);
    run<false, 2, false>("6 MFMA + 1 register-staged piece (load -> AGPR -> ds_write)", stream, out);
    run<true, 2, true>("all, piece register-staged", stream, out);
    printf("--- MFMA shape at equal work, random operands, 400k steps (A B A B) ---\n");
    {
        f32x4* rnd; const int n = 4096;
        (void)hipMalloc(&rnd, n * 16);
        unsigned short* h = new unsigned short[n * 8];
        unsigned x = 12345u;
        for (int i = 0; i < n * 8; ++i) { x = x * 1664525u + 1013904223u; h[i] = (unsigned short)(((x >> 16) & 0x83ffu) | 0x3000u | ((x >> 3) & 0x0c00u)); }   // +-[0.125, 2)
        (void)hipMemcpy(rnd, h, n * 16, hipMemcpyHostToDevice);
        for (int rep = 0; rep < 2; ++rep) {
            run_shape("32x32x16 x6, bare", step_shape32_kernel<false, 0, false>, stream, out, rnd);
            run_shape("16x16x32 x12, bare", step_shape16_kernel<false, 0, false>, stream, out, rnd);
            run_shape("32x32x16 x6, reads + LDS-DMA + VALU", step_shape32_kernel<true, 1, true>, stream, out, rnd);
            run_shape("16x16x32 x12, reads + LDS-DMA + VALU", step_shape16_kernel<true, 1, true>, stream, out, rnd);
            run_shape("32x32x16 x6, reads + register-staged piece + VALU", step_shape32_kernel<true, 2, true>, stream, out, rnd);
            run_shape("16x16x32 x12, reads + register-staged piece + VALU", step_shape16_kernel<true, 2, true>, stream, out, rnd);
            run_shape("32x32x16 x6, reads + LDS-DMA staggered (odd waves 3 MFMAs later) + VALU", step_shape32_kernel<true, 5, true>, stream, out, rnd);
            run_shape("16x16x32 x12, reads + LDS-DMA staggered (odd waves 6 MFMAs later) + VALU", step_shape16_kernel<true, 5, true>, stream, out, rnd);
            run_shape("32x32x16 x6, reads + LDS-DMA (scalar base + lane offset) + VALU", step_shape32_kernel<true, 3, true>, stream, out, rnd);
            run_shape("32x32x16 x6, reads + LDS-DMA (buffer_load ... lds) + VALU", step_shape32_kernel<true, 4, true>, stream, out, rnd);
            run_shape("32x32x16 x6, reads + VALU (no weight traffic)", step_shape32_kernel<true, 0, true>, stream, out, rnd);
            run_shape("16x16x32 x12, reads + VALU (no weight traffic)", step_shape16_kernel<true, 0, true>, stream, out, rnd);
        }
    }
    run<false, false, false>("again: 6 MFMA 32x32x16", stream, out);
    run<true, true, true>("again: 32x32x16 everything", stream, out);
    return 0;
}
