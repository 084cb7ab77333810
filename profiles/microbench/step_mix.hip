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

template <bool READS, bool DMA, bool VALU, int AHEAD = 1, int NREADS = 4, bool BARRIER = true, int SPLIT = 0>
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
    for (int s = 0; s < steps; ++s) {
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
        if (DMA) __builtin_amdgcn_global_load_lds(GLB_PTR(g + (s & 3) * 1024), LDS_PTR(l + (s & 3) * 1024), 16, 0, 0);
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
            if (DMA) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            asm volatile("s_barrier" ::: "memory");
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float sum = m + __uint_as_float(hi) * 0.f;
    for (int r = 0; r < 16; ++r) sum += acc[r];
    out[blockIdx.x * 256 + threadIdx.x] = sum;
}

template <bool READS, bool DMA, bool VALU, int AHEAD = 1, int NREADS = 4, bool BARRIER = true, int SPLIT = 0>
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
    (void)hipMalloc(&out, 256 * 256 * 4);
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
    return 0;
}
