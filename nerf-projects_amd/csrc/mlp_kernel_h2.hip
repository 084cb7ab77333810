// Fused positional-encoding + NeRF MLP kernel, "fp16-pair" arithmetic (NERF_PRECISION_F16X2).
//
// Same job and same structure as mlp_kernel.hip (Embedder.embed nerf/embedder.py:72-80, the viewdir
// broadcast + concat of run_network nerf.ipynb:827-843, NeRF.forward nerf/nerf.py:57-111; a wavefront
// owns 32 points, activations stay feature-major in registers, weights stream L2 -> LDS by LDS-DMA),
// but the contraction runs on the half-precision matrix pipe, which on gfx950 has 16x the rate of
// v_mfma_f32_32x32x2_f32:
//
//   * every fp32 operand v is carried as two fp16 numbers, v ~= hi + lo with hi = rn16(v) and
//     lo = rn16(v - hi): |v - hi - lo| <= 2^-24 |v|, what fp32 itself keeps. A product W*x is the three
//     v_mfma_f32_32x32x16_f16 terms W_lo*x_hi + W_hi*x_lo + W_hi*x_hi (fp16 x fp16 is exact in the fp32
//     accumulator; the dropped W_lo*x_lo is below 2^-24 relative). Measured on random 256-long dot products the result
//     is within 0.5-1.0 eps(fp32) rms of the exact value, against 0.4-0.5 eps for the fp32 MFMA chain
//     (profiles/microbench/mfma_bf16_split.hip): the stage tolerances of the fp32 path hold.
//   * fp16 has 5 exponent bits, so both operands are kept in range by exact power-of-two scalings:
//     each layer's weights by one factor chosen from the layer's largest |w| (convert kernel below; the
//     inverse factor travels in `descale`), each POINT's activation vector by its own factor chosen
//     from that point's largest activation after every layer. A point is a column of the MFMA, so its
//     factor is a per-lane multiplier folded into the fma that adds the bias: no activation of any
//     magnitude overflows, and the low pieces never fall into the subnormal range.
//   * the bias is added, ReLU applied and the next scale chosen in fp32 on the accumulator values;
//     the alpha and rgb heads stay fp32 VALU dot products on those values.
//
// Cost per layer and wave: 384 MFMAs of 32 cycles against 1024 of 64.
#include "mlp_inputs.h"

namespace nerf {

typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// one 32-feature activation tile as MFMA B operands: k-slice s covers accumulator registers 8s..8s+7
struct XT {
    u32x4 hi[2], lo[2];
};

__device__ __forceinline__ f32x16 mma(const f32x4& a, const u32x4& b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, a), __builtin_bit_cast(h16x8, b), c, 0, 0, 0);
}

// ---- weight-stream pipeline ------------------------------------------------------------
// Four 32 KiB LDS buffers form a ring: while chunk c is consumed (48 MFMAs, ~1500 cycles), chunk c+1
// is resident and chunks c+2, c+3 are in flight, so a load has two chunk periods to land. One barrier
// per chunk, mid-chunk:
//   vmcnt(8)  -> this wave's share of chunk c+1 has landed (only chunk c+2's 8 loads may be pending)
//   s_barrier -> every wave's share has, and every wave has finished chunk c-1
//   then issue chunk c+3 into the buffer chunk c-1 occupied.
constexpr int kRingH = 4;

struct PipeH {
    const char* stream;
    char* lds;
    int c, b, n, wave, lane;
};

__device__ __forceinline__ int ringh_next(int b, int k) {
    b += k;
    return b >= kRingH ? b - kRingH : b;
}

__device__ __forceinline__ void prefetch_chunk(const PipeH& p, int chunk, int slot) {
    const char* g = p.stream + (size_t)chunk * kChunkBytes + p.wave * 8192 + p.lane * 16;
    char* l = p.lds + slot * kChunkBytes + p.wave * 8192;
#pragma unroll
    for (int i = 0; i < 8; ++i)
        __builtin_amdgcn_global_load_lds(GLB_PTR(g + i * 1024), LDS_PTR(l + i * 1024), 16, 0, 0);
}

__device__ __forceinline__ const f32x4* ring_frags(const PipeH& p, int slot) {
    return (const f32x4*)(p.lds + slot * kChunkBytes) + p.lane;
}

// A-fragments of one step (one output tile x one k-tile): [k-slice 0 hi, k-slice 0 lo, k-slice 1 hi, k-slice 1 lo]
struct Frag4 {
    f32x4 q[4];
};
__device__ __forceinline__ Frag4 read_frags(const f32x4* fr, int group) {
    Frag4 f;
#pragma unroll
    for (int i = 0; i < 4; ++i) f.q[i] = fr[(group + i) * 64];
    return f;
}

// the six products of one step, small terms first; PART 0 = the first product, PART 1 = the other five
template <int PART>
__device__ __forceinline__ void mma_step(f32x16& acc, const Frag4& f, const XT& x) {
    if constexpr (PART == 0) {
        acc = mma(f.q[1], x.hi[0], acc);
    } else {
#ifndef NERF_ABLATE_MFMA
        acc = mma(f.q[0], x.lo[0], acc);
#endif
        acc = mma(f.q[0], x.hi[0], acc);
#ifndef NERF_ABLATE_MFMA
        acc = mma(f.q[3], x.hi[1], acc);
        acc = mma(f.q[2], x.lo[1], acc);
#endif
        acc = mma(f.q[2], x.hi[1], acc);
    }
}

template <int S>
struct StepTag {
    static constexpr int value = S;
};
template <int P>
struct PartTag {
    static constexpr int value = P;
};

// Consume the current chunk in NSTEP steps of 6 MFMAs; `cur` holds the fragments of step 0 on entry and
// of the NEXT chunk's step 0 on exit. One wave per SIMD has nobody to hide LDS latency behind, so the order
// is pinned with scheduling fences: first MFMA of step n, the four fragment reads of step n+1 (into the
// other half of a double buffer), the other five MFMAs (160 matrix-pipe cycles for the reads to return).
// Left to itself hipcc sinks the reads below the step's last MFMA to share registers and waits for them.
template <int S, int NSTEP, class Body>
__device__ __forceinline__ void run_steps(PipeH& p, Frag4& cur, const f32x4* fr, const f32x4* fr_next, Body& body) {
    if constexpr (S < NSTEP) {
        body(StepTag<S>{}, PartTag<0>{}, cur);
        __builtin_amdgcn_sched_barrier(0);
        Frag4 nxt = (S + 1 < NSTEP) ? read_frags(fr, (S + 1) * 4) : read_frags(fr_next, 0);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (S == NSTEP / 2) {
            // the step after the barrier also issues the 8 LDS-DMA pieces of chunk c+3, two per MFMA
            const int nx = p.c + 3 < p.n ? p.c + 3 : p.c + 3 - p.n;   // wraps into the next tile's stream
#ifndef NERF_ABLATE_DMA
            prefetch_chunk(p, nx, ringh_next(p.b, 3));
#endif
            body(StepTag<S>{}, PartTag<1>{}, cur);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        } else {
            body(StepTag<S>{}, PartTag<1>{}, cur);
        }
        __builtin_amdgcn_sched_barrier(0);
        cur = nxt;
        if constexpr (S == NSTEP / 2 - 1) {
            asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        }
        run_steps<S + 1, NSTEP>(p, cur, fr, fr_next, body);
    }
}

template <int NSTEP, class Body>
__device__ __forceinline__ void consume_chunk(PipeH& p, Frag4& cur, Body body) {
    const f32x4* fr = ring_frags(p, p.b);
    const f32x4* fr_next = ring_frags(p, ringh_next(p.b, 1));
    run_steps<0, NSTEP>(p, cur, fr, fr_next, body);
    ++p.c;
    p.b = ringh_next(p.b, 1);
}

// chunk kinds: the group order is the fp32 stream's (pack_weights.cpp) with each unit of four groups
// re-cut into [k-slice][hi|lo] by convert_stream_h2 below
__device__ __forceinline__ void chunk_ktile8(PipeH& p, Frag4& cur, f32x16 (&acc)[8], const XT& x) {
    consume_chunk<8>(p, cur, [&](auto tag, auto part, const Frag4& f) {
        mma_step<decltype(part)::value>(acc[decltype(tag)::value], f, x);
    });
}
__device__ __forceinline__ void chunk_ktile4(PipeH& p, Frag4& cur, f32x16 (&acc)[8], const XT& x) {
    consume_chunk<4>(p, cur, [&](auto tag, auto part, const Frag4& f) {
        mma_step<decltype(part)::value>(acc[decltype(tag)::value], f, x);
    });
}
__device__ __forceinline__ void chunk_pair4(PipeH& p, Frag4& cur, f32x16 (&acc)[8], const XT& x0, const XT& x1) {
    consume_chunk<8>(p, cur, [&](auto tag, auto part, const Frag4& f) {
        constexpr int s = decltype(tag)::value;
        mma_step<decltype(part)::value>(acc[s & 3], f, s < 4 ? x0 : x1);
    });
}
template <int NKT>
__device__ __forceinline__ void chunk_row(PipeH& p, Frag4& cur, f32x16& acc, const XT (&x)[8]) {
    consume_chunk<NKT>(p, cur, [&](auto tag, auto part, const Frag4& f) {
        mma_step<decltype(part)::value>(acc, f, x[decltype(tag)::value]);
    });
}

// ---- per-point scaling and the fp16 split --------------------------------------------------------
// exponent t such that max * 2^t lies in [2^9, 2^10): headroom of 64 below the fp16 maximum, and the low
// piece of anything within 2^-13 of the point's largest activation is a normal fp16 number
__device__ __forceinline__ int pick_exponent(float m) {
    const int t = 10 - __builtin_amdgcn_frexp_expf(m);   // frexp_exp(0) = 0
    return t < -60 ? -60 : (t > 60 ? 60 : t);   // keeps descale * 2^-t finite
}
__device__ __forceinline__ float pow2f(int t) { return __builtin_ldexpf(1.0f, t); }

__device__ __forceinline__ float half_max(float m) { return fmaxf(m, __shfl_xor(m, 32)); }

__device__ __forceinline__ float tile_absmax(const f32x16& v, float m) {
#pragma unroll
    for (int r = 0; r < 16; ++r) m = fmaxf(m, fabsf(v[r]));
    return m;
}

// v_cvt_pk_f16_f32: both halves rounded to nearest even. hi = rn16(v) leaves |v - hi| <= 2^-12 |v| (exact in fp32),
// lo = rn16(v - hi) leaves 2^-24 |v|: the pair carries as many bits as the fp32 it came from.
__device__ __forceinline__ h16x2 round_pair(float a, float b) {
    const f32x2 v = {a, b};
    return __builtin_convertvector(v, h16x2);
}

// v * sc -> (hi, lo) for the 16 registers of one tile
__device__ __forceinline__ void split_tile(XT& out, const f32x16& v, float sc) {
#ifdef NERF_ABLATE_SPLIT
    out.hi[0][0] = __float_as_uint(v[0] * sc);
    return;
#endif
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float a = v[8 * s + 2 * q] * sc, b = v[8 * s + 2 * q + 1] * sc;
            const h16x2 hi = round_pair(a, b);
            const h16x2 lo = round_pair(a - (float)hi[0], b - (float)hi[1]);
            out.hi[s][q] = __builtin_bit_cast(unsigned, hi);
            out.lo[s][q] = __builtin_bit_cast(unsigned, lo);
        }
}

// multiply a split tile by 2^d (exact while nothing leaves the fp16 range; d <= 0 by construction)
__device__ __forceinline__ void rescale_tile(XT& x, int d) {
    const _Float16 f = (_Float16)pow2f(d < -30 ? -30 : d);
    const h16x2 ff = {f, f};
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            // (bit_cast straight from a vector element reads the wrong lane with this hipcc: go through a scalar)
            const unsigned wh = x.hi[s][q], wl = x.lo[s][q];
            const h16x2 ph = __builtin_bit_cast(h16x2, wh) * ff, pl = __builtin_bit_cast(h16x2, wl) * ff;
            x.hi[s][q] = __builtin_bit_cast(unsigned, ph);
            x.lo[s][q] = __builtin_bit_cast(unsigned, pl);
        }
}

// 64 bytes of the bias block per lane. hipcc guards every LDS load it can see with s_waitcnt vmcnt(0) while an
// LDS-DMA write is in flight (it cannot tell the bias block from the ring), which would drain the weight pipeline
// twice per layer; these reads are therefore issued from inline asm, with their own lgkmcnt wait (LDS returns in
// order, and the waits hipcc computes for its own reads can only become stricter by the extra entries).
struct Tile16 {
    f32x4 q[4];
};
__device__ __forceinline__ Tile16 lds_tile_issue(const float* p) {
    Tile16 t;
    const unsigned addr = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) float*)p;
    asm volatile(
        "ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:16\n\tds_read_b128 %2, %4 offset:32\n\t"
        "ds_read_b128 %3, %4 offset:48"
        : "=&v"(t.q[0]), "=&v"(t.q[1]), "=&v"(t.q[2]), "=&v"(t.q[3])
        : "v"(addr)
        : "memory");
    return t;
}
__device__ __forceinline__ void lds_tile_wait(Tile16& t) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(t.q[0]), "+v"(t.q[1]), "+v"(t.q[2]), "+v"(t.q[3])::"memory");
}
__device__ __forceinline__ float lds_scalar(const float* p) {
    float v;
    const unsigned addr = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) float*)p;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(addr) : "memory");
    return v;
}

// y = acc * c + bias (and ReLU above `floor`); returns this lane's largest |y|
template <int N>
__device__ __forceinline__ float finish_layer(f32x16 (&y)[8], const f32x16 (&acc)[8], const float* bias_lds, int tile,
                                              int h, float c, float floor) {
    float m0 = 0.0f, m1 = 0.0f;
    Tile16 nxt = lds_tile_issue(bias_lds + (tile * 2 + h) * 16);
#pragma unroll
    for (int t = 0; t < N; ++t) {
        Tile16 b = nxt;
        lds_tile_wait(b);
        if (t + 1 < N) nxt = lds_tile_issue(bias_lds + ((tile + t + 1) * 2 + h) * 16);
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            const float v0 = fmaxf(fmaf(acc[t][r], c, b.q[r >> 2][r & 3]), floor);
            const float v1 = fmaxf(fmaf(acc[t][r + 1], c, b.q[(r + 1) >> 2][(r + 1) & 3]), floor);
            y[t][r] = v0;
            y[t][r + 1] = v1;
            m0 = fmaxf(m0, fabsf(v0));
            m1 = fmaxf(m1, fabsf(v1));
        }
    }
    return fmaxf(m0, m1);
}

template <int N>
__device__ __forceinline__ void zero_tiles(f32x16 (&acc)[8]) {
#pragma unroll
    for (int t = 0; t < N; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
}

// One output row of a Linear over NKT fp32 activation tiles (weights per register in the bias block)
template <int NKT>
__device__ __forceinline__ float row_dot(const f32x16 (&x)[8], const float* bias_lds, int tile, int h) {
    float s0 = 0.0f, s1 = 0.0f;
    Tile16 nxt = lds_tile_issue(bias_lds + (tile * 2 + h) * 16);
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
        Tile16 w = nxt;
        lds_tile_wait(w);
        if (kt + 1 < NKT) nxt = lds_tile_issue(bias_lds + ((tile + kt + 1) * 2 + h) * 16);
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            s0 = fmaf(w.q[r >> 2][r & 3], x[kt][r], s0);
            s1 = fmaf(w.q[(r + 1) >> 2][(r + 1) & 3], x[kt][r + 1], s1);
        }
    }
    const float s = s0 + s1;
    return s + __shfl_xor(s, 32);
}

// ---- the kernel -------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void nerf_mlp_h2_kernel(const MlpLaunch a) {
    // The ring is the dynamic LDS allocation, the bias block a separate static one: hipcc guards every LDS read that
    // may alias an in-flight LDS-DMA write with s_waitcnt vmcnt(0), which would drain the weight pipeline at each
    // bias read; two distinct LDS objects cannot alias.
    extern __shared__ __attribute__((aligned(16))) char ring_lds[];
    __shared__ __attribute__((aligned(16))) float bias_lds[kBiasLdsBytes / 4];
    __shared__ float descale_lds[kMaxDepth + 2];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane >> 5;

    PipeH pipe{(const char*)a.stream_h2, ring_lds, 0, 0, a.n_chunks, wave, lane};
    prefetch_chunk(pipe, 0, 0);
    prefetch_chunk(pipe, 1, 1);
    prefetch_chunk(pipe, 2, 2);
    for (int i = threadIdx.x; i < a.n_bias_tiles * kBiasTileFloats; i += 256) bias_lds[i] = a.bias[i];
    if (threadIdx.x < a.D + 2) descale_lds[threadIdx.x] = a.descale[threadIdx.x];
    __syncthreads();   // chunks 0..2, the bias block and the layer scales are in LDS
    Frag4 cur = read_frags(ring_frags(pipe, 0), 0);

    const int64_t n_tiles = (a.n_points + kPointsPerGroup - 1) / kPointsPerGroup;
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        pipe.c = 0;
        const int64_t tile0 = tile * kPointsPerGroup + wave * kPointsPerWave;
        const int64_t pt_raw = tile0 + (lane & 31);
        const int64_t pt = pt_raw < a.n_points ? pt_raw : a.n_points - 1;

        XT xp0, xp1, xd;
        float m_pe, m_dd;
        int t_pe, t_dd;
        {
            f32x16 x0, x1, dd;
            load_inputs<MODE>(a, pt, h, x0, x1, dd);
            m_pe = half_max(tile_absmax(x1, tile_absmax(x0, 0.0f)));
            m_dd = half_max(tile_absmax(dd, 0.0f));
            t_pe = pick_exponent(m_pe);
            t_dd = pick_exponent(m_dd);
            const float spe = pow2f(t_pe), sdd = pow2f(t_dd);
            split_tile(xp0, x0, spe);
            split_tile(xp1, x1, spe);
            split_tile(xd, dd, sdd);
        }

        XT hid[8];
        f32x16 acc[8], y[8];
        int t_cur;
        float sigma = 0.0f;

        // layer 0: gamma(xyz) -> W (nerf.py:70-73)
        zero_tiles<8>(acc);
        chunk_ktile8(pipe, cur, acc, xp0);
        chunk_ktile8(pipe, cur, acc, xp1);
        {
            float m = half_max(finish_layer<8>(y, acc, bias_lds, 0, h, lds_scalar(descale_lds) * pow2f(-t_pe), 0.0f));
            if (a.D == 1 && a.use_viewdirs) sigma = row_dot<8>(y, bias_lds, 8 * a.D + 14, h) + lds_scalar(bias_lds + (8 * a.D) * 32);
            if ((a.skip_in_mask >> 1) & 1) m = fmaxf(m, m_pe);
            t_cur = pick_exponent(m);
            const float sc = pow2f(t_cur);
#pragma unroll
            for (int t = 0; t < 8; ++t) split_tile(hid[t], y[t], sc);
        }

        // trunk layers 1..D-1, then (with viewdirs) feature_linear as layer D without ReLU
        const int n_layers = a.use_viewdirs ? a.D + 1 : a.D;
        for (int i = 1; i < n_layers; ++i) {
            const bool is_feature = (i == a.D);
            const float c = lds_scalar(descale_lds + i) * pow2f(-t_cur);
            zero_tiles<8>(acc);
            if (!is_feature && ((a.skip_in_mask >> i) & 1)) {
                // h = cat[input_pts, h] (nerf.py:79-80): bring the encoded inputs to this layer's scale
                rescale_tile(xp0, t_cur - t_pe);
                rescale_tile(xp1, t_cur - t_pe);
                t_pe = t_cur;
                chunk_ktile8(pipe, cur, acc, xp0);
                chunk_ktile8(pipe, cur, acc, xp1);
            }
#pragma unroll
            for (int kt = 0; kt < 8; ++kt) chunk_ktile8(pipe, cur, acc, hid[kt]);
            float m = half_max(finish_layer<8>(y, acc, bias_lds, is_feature ? 8 * a.D + 1 : 8 * i, h, c,
                                               is_feature ? -__builtin_inff() : 0.0f));
            if (i == a.D - 1 && a.use_viewdirs) {
                // alpha_linear reads the post-ReLU trunk output (nerf.py:86): one row, as a dot product
                sigma = row_dot<8>(y, bias_lds, 8 * a.D + 14, h) + lds_scalar(bias_lds + (8 * a.D) * 32);
            }
            if (is_feature) m = fmaxf(m, m_dd);
            else if ((a.skip_in_mask >> (i + 1)) & 1) m = fmaxf(m, m_pe);
            t_cur = pick_exponent(m);
            const float sc = pow2f(t_cur);
#pragma unroll
            for (int t = 0; t < 8; ++t) split_tile(hid[t], y[t], sc);
        }

        const bool live = pt_raw < a.n_points;
        if (a.use_viewdirs) {
            // views_linears[0] on cat[feature, gamma(dir)] (nerf.py:93-98): 4 output tiles
            rescale_tile(xd, t_cur - t_dd);
            zero_tiles<4>(acc);
#pragma unroll
            for (int kp = 0; kp < 4; ++kp) chunk_pair4(pipe, cur, acc, hid[2 * kp], hid[2 * kp + 1]);
            chunk_ktile4(pipe, cur, acc, xd);
            finish_layer<4>(y, acc, bias_lds, 8 * a.D + 9, h, lds_scalar(descale_lds + a.D + 1) * pow2f(-t_cur), 0.0f);
            // rgb_linear (nerf.py:101): three rows over the 128-wide view layer
            const float* rb = bias_lds + (8 * a.D + 13) * 32;
            const float r0 = row_dot<4>(y, bias_lds, 8 * a.D + 22, h) + lds_scalar(rb);
            const float r1 = row_dot<4>(y, bias_lds, 8 * a.D + 26, h) + lds_scalar(rb + 1);
            const float r2 = row_dot<4>(y, bias_lds, 8 * a.D + 30, h) + lds_scalar(rb + 2);
            if (live && h == 0) {
                f32x4 o = {r0, r1, r2, sigma};   // outputs = cat[rgb, alpha] (nerf.py:106)
                *(f32x4*)(a.out + pt * 4) = o;
            }
        } else {
            // output_linear (nerf.py:109): rows 0..out_ch-1 of one tile
            f32x16 o;
#pragma unroll
            for (int r = 0; r < 16; ++r) o[r] = 0.0f;
            chunk_row<8>(pipe, cur, o, hid);
            Tile16 b = lds_tile_issue(bias_lds + ((8 * a.D) * 2 + h) * 16);
            lds_tile_wait(b);
            const float c = lds_scalar(descale_lds + a.D) * pow2f(-t_cur);
            if (live) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
                    if (row < a.out_ch) a.out[pt * a.out_ch + row] = fmaf(o[r], c, b.q[r >> 2][r & 3]);
                }
            }
        }
    }   // tile loop
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

hipError_t launch_mlp_h2(const MlpLaunch& a, int mode, hipStream_t s) {
    if (a.n_points <= 0) return hipSuccess;
    if (!a.stream_h2 || !a.descale) return hipErrorInvalidValue;
    const int64_t tiles = (a.n_points + kPointsPerGroup - 1) / kPointsPerGroup;
    static int n_cu[64] = {};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (!n_cu[dev]) {
        e = hipDeviceGetAttribute(&n_cu[dev], hipDeviceAttributeMultiprocessorCount, dev);
        if (e != hipSuccess) return e;
        if (n_cu[dev] <= 0) n_cu[dev] = 256;
    }
    const dim3 grid((unsigned)(tiles < n_cu[dev] ? tiles : n_cu[dev])), block(256);
    const size_t lds = kRingH * kChunkBytes;   // + 20.5 KiB static (bias block, layer scales)
    static bool raised[64][3] = {};
    if (mode < 0 || mode > 2) return hipErrorInvalidValue;
    if (!raised[dev][mode]) {
        const void* fn = mode == kInputEmbedded ? (const void*)nerf_mlp_h2_kernel<kInputEmbedded>
                         : mode == kInputPoints ? (const void*)nerf_mlp_h2_kernel<kInputPoints>
                                                : (const void*)nerf_mlp_h2_kernel<kInputRays>;
        e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        raised[dev][mode] = true;
    }
    switch (mode) {
        case kInputEmbedded:
            hipLaunchKernelGGL(nerf_mlp_h2_kernel<kInputEmbedded>, grid, block, lds, s, a);
            break;
        case kInputPoints:
            hipLaunchKernelGGL(nerf_mlp_h2_kernel<kInputPoints>, grid, block, lds, s, a);
            break;
        default:
            hipLaunchKernelGGL(nerf_mlp_h2_kernel<kInputRays>, grid, block, lds, s, a);
            break;
    }
    return hipGetLastError();
}

// ---- fp32 stream -> fp16-pair stream -----------------------------------------------------------------
// The fp32 stream's unit of four groups (one output tile x one k-tile: group t4 holds this lane's k-steps
// 4*t4..4*t4+3) becomes [k-slice s][hi|lo] with k-slice s = old groups 2s, 2s+1: exactly the 8 fp16 one lane
// feeds a 32x32x16 MFMA. All chunks of a layer share one power-of-two scale.
__global__ __launch_bounds__(256) void chunk_absmax_kernel(const float* stream, float* chunk_max) {
    __shared__ float red[4];
    const float* c = stream + (size_t)blockIdx.x * kChunkFloats;
    float m = 0.0f;
    for (int i = threadIdx.x; i < kChunkFloats; i += 256) m = fmaxf(m, fabsf(c[i]));
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) chunk_max[blockIdx.x] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

__global__ __launch_bounds__(256) void convert_stream_h2_kernel(const float* stream, const int* chunk_layer,
                                                                const float* chunk_max, int n_chunks,
                                                                uint32_t* out, float* descale) {
    const int chunk = blockIdx.x, layer = chunk_layer[chunk];
    float m = 0.0f;
    bool first = true;
    for (int i = 0; i < n_chunks; ++i)
        if (chunk_layer[i] == layer) {
            m = fmaxf(m, chunk_max[i]);
            if (i < chunk) first = false;
        }
    // largest weight of the layer -> [2^12, 2^13); an all-zero or non-finite layer keeps scale 1
    int e = (m > 0.0f && m < __builtin_inff()) ? 13 - __builtin_amdgcn_frexp_expf(m) : 0;
    e = e < -60 ? -60 : (e > 60 ? 60 : e);
    const float sc = __builtin_ldexpf(1.0f, e);
    if (first && threadIdx.x == 0) descale[layer] = __builtin_ldexpf(1.0f, -e);
    const float* src = stream + (size_t)chunk * kChunkFloats;
    uint32_t* dst = out + (size_t)chunk * kChunkFloats;
    // thread = (unit u, k-slice s, lane): 8 units x 2 x 64 = 1024 work items, 4 per thread
    for (int w = threadIdx.x; w < 1024; w += 256) {
        const int lane = w & 63, s = (w >> 6) & 1, u = w >> 7;
        const f32x4 a = *(const f32x4*)(src + ((4 * u + 2 * s) * 64 + lane) * 4);
        const f32x4 b = *(const f32x4*)(src + ((4 * u + 2 * s + 1) * 64 + lane) * 4);
        const float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
        u32x4 hi, lo;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float x0 = v[2 * q] * sc, x1 = v[2 * q + 1] * sc;
            const h16x2 p = round_pair(x0, x1);
            const h16x2 r = round_pair(x0 - (float)p[0], x1 - (float)p[1]);
            hi[q] = __builtin_bit_cast(unsigned, p);
            lo[q] = __builtin_bit_cast(unsigned, r);
        }
        *(u32x4*)(dst + ((4 * u + 2 * s) * 64 + lane) * 4) = hi;
        *(u32x4*)(dst + ((4 * u + 2 * s + 1) * 64 + lane) * 4) = lo;
    }
}

hipError_t launch_convert_stream_h2(const float* stream, const int* chunk_layer, int n_chunks, float* chunk_max,
                                    uint32_t* out, float* descale, hipStream_t s) {
    if (n_chunks <= 0) return hipSuccess;
    hipLaunchKernelGGL(chunk_absmax_kernel, dim3(n_chunks), dim3(256), 0, s, stream, chunk_max);
    hipLaunchKernelGGL(convert_stream_h2_kernel, dim3(n_chunks), dim3(256), 0, s, stream, chunk_layer, chunk_max,
                       n_chunks, out, descale);
    return hipGetLastError();
}

}  // namespace nerf
