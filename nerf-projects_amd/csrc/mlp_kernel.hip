// Fused positional-encoding + NeRF MLP kernel for gfx950 (MI355X, CDNA4).
//
// Replaces, per sample point, Embedder.embed (nerf/embedder.py:72-80), the per-sample
// viewdir broadcast + concat of run_network (nerf.ipynb:827-843) and NeRF.forward
// (nerf/nerf.py:57-111). One launch evaluates the whole network for every point; no
// encoded features or hidden activations ever reach HBM.
//
// Mapping to the hardware
//   * A wavefront owns 32 consecutive points. Activations are kept feature-major in
//     registers: the 256-wide hidden vector of those 32 points is 8 MFMA 32x32 accumulator
//     tiles (8 x 16 = 128 registers); a second set of 128 registers receives the next
//     layer. With `amdgpu_waves_per_eu(1,1)` a wave may use the whole 512-entry unified
//     VGPR/AGPR file, so both sets, the encoded inputs (32 + 16 registers, kept for the
//     skip connection) and the operand staging fit without spilling.
//   * Every product is v_mfma_f32_32x32x2_f32 (fp32 in, fp32 accumulate: bit-for-bit an
//     fmaf chain, the only MFMA that can hold the 1e-6 stage tolerance). Weights are the A
//     operand, activations the B operand; because the accumulator layout of one layer is
//     already the B-operand layout of the next (pack_weights.cpp explains the permutation),
//     layers chain with no LDS round trip and no cross-lane traffic: the only per-layer
//     vector work is the ReLU (v_max) that also moves the tile into the operand set.
//   * Weights (2.4 MB per network) are streamed L2 -> LDS in 32 KiB chunks by LDS-DMA
//     (`global_load_lds_dwordx4`, 8 wave-instructions per wave per chunk) through a ring of
//     three buffers: chunk c+2 is in flight and chunk c+1 resident while the four waves of
//     the workgroup run the 128 MFMAs of chunk c (8192 matrix-pipe cycles per SIMD). One
//     barrier per chunk, placed mid-chunk where its counters are already drained, so the MFMA
//     stream runs across chunk boundaries. A-fragments are read with `ds_read_b128`
//     (4 k-steps per lane per read, lane-linear => conflict-free).
//   * Biases are pre-arranged per accumulator register and read from LDS straight into
//     the accumulator tile (no MFMA, no VALU).
//
// Roofline: 593,408 MAC/point -> 9,280 MFMA per 32 points (99.8 % of them useful; the rest
// is K padding of the 63- and 27-wide encodings; the 1-row alpha and 3-row rgb heads are VALU
// dot products). Bound: fp32 MFMA, 64 cycles per v_mfma_f32_32x32x2_f32 per SIMD,
// 157.3 TFLOP/s per chip.
#include "mlp_inputs.h"

namespace nerf {

__device__ __forceinline__ f32x16 mfma(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// ---- weight-stream pipeline ------------------------------------------------------------
// Three 32 KiB LDS buffers form a ring. While chunk c is being consumed, chunk c+1 is already
// resident and chunk c+2 is in flight. The ONE barrier per chunk sits in the middle of the
// chunk's MFMA sequence, at a step boundary where the in-order LDS counter is already drained:
//   vmcnt(0)  -> this wave's share of chunk c+1 (issued a whole chunk ago) has landed
//   s_barrier -> every wave's share has, and every wave has finished chunk c-1
//   then issue chunk c+2 into the buffer chunk c-1 occupied.
// Because chunk c+1 is visible from that point on, the last step of chunk c prefetches the first
// A-fragments of chunk c+1, so the MFMA stream runs across chunk boundaries without a bubble.
constexpr int kRing = 3;

struct Pipe {
    const char* stream;   // packed chunks in HBM/L2
    char* lds;            // kRing chunk buffers
    int c;                // chunk being consumed
    int b;                // ring slot of chunk c
    int n;                // chunks in the stream
    int wave;             // wave-uniform
    int lane;
};

__device__ __forceinline__ int ring_next(int b, int k) {
    b += k;
    return b >= kRing ? b - kRing : b;
}

__device__ __forceinline__ void prefetch_chunk(const Pipe& p, int chunk, int slot) {
    const char* g = p.stream + (size_t)chunk * kChunkBytes + p.wave * 8192 + p.lane * 16;
    char* l = p.lds + slot * kChunkBytes + p.wave * 8192;
    // two address pairs per chunk, the pieces selected by the instruction's immediate offset (<= 3 KiB): with one pointer
    // per piece hipcc precomputes (and spills) eight 64-bit addresses for every unrolled chunk
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const char* gh = g + half * 4096;
        char* lh = l + half * 4096;
        __builtin_amdgcn_global_load_lds(GLB_PTR(gh), LDS_PTR(lh), 16, 0, 0);
        __builtin_amdgcn_global_load_lds(GLB_PTR(gh), LDS_PTR(lh), 16, 1024, 0);
        __builtin_amdgcn_global_load_lds(GLB_PTR(gh), LDS_PTR(lh), 16, 2048, 0);
        __builtin_amdgcn_global_load_lds(GLB_PTR(gh), LDS_PTR(lh), 16, 3072, 0);
    }
}

__device__ __forceinline__ const f32x4* ring_frags(const Pipe& p, int slot) {
    return (const f32x4*)(p.lds + slot * kChunkBytes) + p.lane;
}

// A-fragments of one step: 4 x ds_read_b128 = 16 k-steps per lane
struct Frag16 {
    f32x4 q[4];
};
__device__ __forceinline__ Frag16 read_frags(const f32x4* fr, int group) {
    Frag16 f;
#pragma unroll
    for (int t4 = 0; t4 < 4; ++t4) f.q[t4] = fr[(group + t4) * 64];
    return f;
}
// k-steps [LO, HI) of one step's 16
template <int LO, int HI>
__device__ __forceinline__ void mma_range(f32x16& acc, const Frag16& f, const f32x16& b) {
#pragma unroll
    for (int k = LO; k < HI; ++k) acc = mfma(f.q[k >> 2][k & 3], b[k], acc);
}

template <int S>
struct StepTag {
    static constexpr int value = S;
};
template <int LO, int HI>
struct Range {
    static constexpr int lo = LO, hi = HI;
};

// Consume the current chunk in NSTEP steps of 16 MFMAs. `body(StepTag<s>, Range<lo,hi>, frags)`
// issues k-steps [lo,hi) of step s; `cur` holds the fragments of step 0 on entry and of the NEXT
// chunk's step 0 on exit.
//
// The barrier follows step NSTEP/2-1 (see Pipe); the 8 LDS-DMA issues of chunk c+2 come right
// after it.
struct NoPost {
    __device__ __forceinline__ void operator()() const {}
};
// `post()` runs once per chunk, at the end of the step that follows the mid-chunk barrier: the training kernels issue a
// tile's global stores there, a whole barrier period (7 of 8 steps) before the next barrier's vmcnt(0) asks for their
// acknowledgement. Issued in one burst at a layer boundary, 32 stores per lane were still in flight at the next barrier.
template <int S, int NSTEP, class Body, class Post>
__device__ __forceinline__ void run_steps(Pipe& p, Frag16& cur, const f32x4* fr, const f32x4* fr_next, Body& body,
                                          Post& post) {
    if constexpr (S < NSTEP) {
        // order pinned with scheduling fences: the step's first MFMA, the four fragment reads of step S+1 into the
        // other half of a double buffer, then the other 15 MFMAs (960 matrix-pipe cycles for the reads to return).
        // Left to itself hipcc sinks the reads below the step's last MFMA to share registers and waits for them.
        body(StepTag<S>{}, Range<0, 1>{}, cur);
        __builtin_amdgcn_sched_barrier(0);
        Frag16 nxt = (S + 1 < NSTEP) ? read_frags(fr, (S + 1) * 4) : read_frags(fr_next, 0);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (S == NSTEP / 2) {
            // the step after the barrier also issues the 8 LDS-DMA pieces of chunk c+2, one per MFMA
            const int nx = p.c + 2 < p.n ? p.c + 2 : p.c + 2 - p.n;   // wraps into the next tile's stream
            prefetch_chunk(p, nx, ring_next(p.b, 2));
            body(StepTag<S>{}, Range<1, 16>{}, cur);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 7, 0);
            __builtin_amdgcn_sched_barrier(0);
            post();
        } else {
            body(StepTag<S>{}, Range<1, 16>{}, cur);
        }
        __builtin_amdgcn_sched_barrier(0);
        cur = nxt;
        if constexpr (S == NSTEP / 2 - 1) {
#ifndef NERF_ABLATE_BARRIER
            __syncthreads();
#endif
            __builtin_amdgcn_sched_barrier(0);
        }
        run_steps<S + 1, NSTEP>(p, cur, fr, fr_next, body, post);
    }
}

template <int NSTEP, class Body, class Post = NoPost>
__device__ __forceinline__ void consume_chunk(Pipe& p, Frag16& cur, Body body, Post post = Post{}) {
    const f32x4* fr = ring_frags(p, p.b);
    const f32x4* fr_next = ring_frags(p, ring_next(p.b, 1));
    run_steps<0, NSTEP>(p, cur, fr, fr_next, body, post);
    ++p.c;
    p.b = ring_next(p.b, 1);
}

// chunk kinds (group orders fixed by pack_weights.cpp) -------------------------------------------
// one k-tile against 8 output tiles: step s = output tile s
template <class Post = NoPost>
__device__ __forceinline__ void chunk_ktile8(Pipe& p, Frag16& cur, f32x16 (&acc)[8], const f32x16& b, Post post = Post{}) {
    consume_chunk<8>(p, cur, [&](auto tag, auto rng, const Frag16& f) {
        mma_range<decltype(rng)::lo, decltype(rng)::hi>(acc[decltype(tag)::value], f, b);
    }, post);
}
// one k-tile against 4 output tiles (direction part of the view layer)
__device__ __forceinline__ void chunk_ktile4(Pipe& p, Frag16& cur, f32x16 (&acc)[8], const f32x16& b) {
    consume_chunk<4>(p, cur, [&](auto tag, auto rng, const Frag16& f) {
        mma_range<decltype(rng)::lo, decltype(rng)::hi>(acc[decltype(tag)::value], f, b);
    });
}
// two k-tiles against 4 output tiles (feature part of the view layer): steps 0-3 use b0, 4-7 use b1
template <class Post = NoPost>
__device__ __forceinline__ void chunk_pair4(Pipe& p, Frag16& cur, f32x16 (&acc)[8], const f32x16& b0,
                                            const f32x16& b1, Post post = Post{}) {
    consume_chunk<8>(p, cur, [&](auto tag, auto rng, const Frag16& f) {
        constexpr int s = decltype(tag)::value;
        mma_range<decltype(rng)::lo, decltype(rng)::hi>(acc[s & 3], f, s < 4 ? b0 : b1);
    }, post);
}
// NKT k-tiles against ONE output tile: step s = k-tile s
template <int NKT>
__device__ __forceinline__ void chunk_row(Pipe& p, Frag16& cur, f32x16& acc, const f32x16 (&b)[8]) {
    consume_chunk<NKT>(p, cur, [&](auto tag, auto rng, const Frag16& f) {
        mma_range<decltype(rng)::lo, decltype(rng)::hi>(acc, f, b[decltype(tag)::value]);
    });
}

template <int N>
__device__ __forceinline__ void load_bias(f32x16 (&acc)[8], const float* bias_lds, int tile, int h) {
#pragma unroll
    for (int ot = 0; ot < N; ++ot)
        acc[ot] = *(const f32x16*)(bias_lds + ((tile + ot) * 2 + h) * 16);
}

// One output row of a Linear over NKT activation tiles held in registers: this lane's 16*NKT products
// (weights arranged per register in the bias block, see pack_weights.cpp row_tiles) plus the other
// half-wave's. Both half-waves return the full sum.
template <int NKT>
__device__ __forceinline__ float row_dot(const f32x16 (&x)[8], const float* bias_lds, int tile, int h) {
    float s0 = 0.0f, s1 = 0.0f;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
        const f32x16 w = *(const f32x16*)(bias_lds + ((tile + kt) * 2 + h) * 16);
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            s0 = fmaf(w[r], x[kt][r], s0);
            s1 = fmaf(w[r + 1], x[kt][r + 1], s1);
        }
    }
    const float s = s0 + s1;
    return s + __shfl_xor(s, 32);
}

// Returns the largest activation (its magnitude without ReLU) of this lane: +inf means the layer overflowed fp32 here,
// which poisons what follows in the reference - F.relu keeps +inf and NaN (nerf.py:72) and the next Linear mixes inf - inf -
// while v_max_f32 drops the NaNs that mix produces. One v_max3 per register pair: +1.1 % on this kernel (a compare per value
// into a scalar mask cost 4 %; ReLU as an integer maximum with 0 on the bit pattern - which keeps a POSITIVE NaN for free -
// lost them all: the matrix pipe's inf - inf carries the sign bit).
template <int N, bool RELU>
__device__ __forceinline__ float activate(f32x16 (&dst)[8], const f32x16 (&src)[8]) {
    float top = 0.0f;
#pragma unroll
    for (int t = 0; t < N; ++t)
#pragma unroll
#ifdef NERF_ABLATE_RELU
        for (int r = 0; r < 16; ++r) dst[t][r] = (r == 0 && t == 0) ? src[t][r] : dst[t][r];
#else
        for (int r = 0; r < 16; r += 2) {
            dst[t][r] = RELU ? fmaxf(src[t][r], 0.0f) : src[t][r];
            dst[t][r + 1] = RELU ? fmaxf(src[t][r + 1], 0.0f) : src[t][r + 1];
            top = fmaxf(fmaxf(top, fabsf(dst[t][r])), fabsf(dst[t][r + 1]));
        }
#endif
    return top;
}

// Training forward (STORE): N activation tiles of this wave's 32 points to a row-major [points, channels] buffer. A lane
// holds, per tile t and register quad q, the four consecutive features 32 t + 8 q + 4 h .. + 3 of its point; the two
// half-waves of a point write adjacent 16-byte pieces, so one store instruction covers 32 bytes of each of 32 rows.
// Rows of the concat buffers ([gamma(x) | h], [feature | gamma(dir)]) are not 16-byte aligned: dword stores there.
template <int N>
__device__ __forceinline__ void store_tiles(float* base, int ld, const f32x16 (&t)[8], int64_t pt, int h, bool live) {
    if (base == nullptr || !live) return;
    // wave-uniform base + 32-bit element offset (the buffers of a pass stay below 2^32 bytes: train_api.cpp checks):
    // one VGPR per row address instead of a 64-bit pair per buffer, which hipcc otherwise precomputes for every
    // buffer of the launch record and spills
    float* row = base + pt * (int64_t)ld + 4 * h;
    const bool vec = ((ld & 3) == 0) && ((reinterpret_cast<uintptr_t>(base) & 15) == 0);   // wave-uniform
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float* p = row + 32 * i + 8 * q;
            if (vec) {
                *(f32x4*)p = f32x4{t[i][4 * q], t[i][4 * q + 1], t[i][4 * q + 2], t[i][4 * q + 3]};
            } else {
                p[0] = t[i][4 * q];
                p[1] = t[i][4 * q + 1];
                p[2] = t[i][4 * q + 2];
                p[3] = t[i][4 * q + 3];
            }
        }
}

// one tile (feature columns 32 i .. 32 i + 31) of the same
__device__ __forceinline__ void store_tile(float* base, int ld, const f32x16& t, int i, int64_t pt, int h, bool live) {
    if (base == nullptr || !live) return;
    float* row = base + (uint32_t)((uint32_t)pt * (uint32_t)ld + 4u * (uint32_t)h) + 32 * i;
    const bool vec = ((ld & 3) == 0) && ((reinterpret_cast<uintptr_t>(base) & 15) == 0);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float* p = row + 8 * q;
        if (vec) {
            *(f32x4*)p = f32x4{t[4 * q], t[4 * q + 1], t[4 * q + 2], t[4 * q + 3]};
        } else {
            p[0] = t[4 * q];
            p[1] = t[4 * q + 1];
            p[2] = t[4 * q + 2];
            p[3] = t[4 * q + 3];
        }
    }
}

// The hooks that ride inside the chunk loop use these instead: straight-line code. store_tile()'s tests (buffer present?
// rows 16-byte aligned? lane past the end?) are branches in the middle of the pinned MFMA stream - 490 of them in the
// training forward kernel, which ran a tile in 306 us against 256 us for the store-free inference kernel while the stores
// themselves accounted for 1 % (ablation). Here the launcher guarantees present, 16-byte aligned buffers (training_rows_ok)
// and lanes past the end write what the lane of the last point writes (they recompute that point: same values, same
// address). base: wave-uniform; off: this lane's element offset of (point, 4 h), 32 bits (the launchers bound n_points).
struct RowRef {
    float* base;
    uint32_t off;
};
__device__ __forceinline__ RowRef row_ref(const float* base, int ld, int64_t pt, int h) {
    return RowRef{const_cast<float*>(base), (uint32_t)pt * (uint32_t)ld + 4u * (uint32_t)h};
}
__device__ __forceinline__ void store_tile_at(const RowRef& r, const f32x16& t, int i) {
    float* p = r.base + r.off + 32 * i;
#pragma unroll
    for (int q = 0; q < 4; ++q) *(f32x4*)(p + 8 * q) = f32x4{t[4 * q], t[4 * q + 1], t[4 * q + 2], t[4 * q + 3]};
}
__device__ __forceinline__ void load_tile_at(const RowRef& r, f32x16& t, int i) {
    const float* p = r.base + r.off + 32 * i;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 v = *(const f32x4*)(p + 8 * q);
        t[4 * q] = v[0];
        t[4 * q + 1] = v[1];
        t[4 * q + 2] = v[2];
        t[4 * q + 3] = v[3];
    }
}
static bool training_rows_ok(const void* base, int ld) {
    return base != nullptr && (ld & 3) == 0 && (reinterpret_cast<uintptr_t>(base) & 15) == 0;
}

// ---- the kernel -------------------------------------------------------------------------
template <int MODE, bool STORE = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void nerf_mlp_kernel(const MlpLaunch a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* bias_lds = (float*)smem;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane >> 5;

    Pipe pipe{(const char*)a.stream, smem + kBiasLdsBytes, 0, 0, a.n_chunks, wave, lane};
    prefetch_chunk(pipe, 0, 0);
    prefetch_chunk(pipe, a.n_chunks > 1 ? 1 : 0, 1);
    for (int i = threadIdx.x; i < a.n_bias_tiles * kBiasTileFloats; i += 256) bias_lds[i] = a.bias[i];
    __syncthreads();   // chunks 0 and 1 and the bias block are in LDS
    Frag16 cur = read_frags(ring_frags(pipe, 0), 0);

    // Persistent workgroup: one per CU, walking 128-point tiles with stride gridDim.x. The weight
    // ring never drains between tiles: the tail of one tile's stream prefetches the head of the
    // next (prefetch wraps modulo n_chunks), and the bias block stays in LDS.
    const int64_t n_tiles = (a.n_points + kPointsPerGroup - 1) / kPointsPerGroup;
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    pipe.c = 0;
    const int64_t tile0 = tile * kPointsPerGroup + wave * kPointsPerWave;
    const int64_t pt_raw = tile0 + (lane & 31);
    const int64_t pt = pt_raw < a.n_points ? pt_raw : a.n_points - 1;   // clamp: padded lanes recompute the last point
    const bool live = pt_raw < a.n_points;

    f32x16 x0, x1, dd;
    load_inputs<MODE>(a, pt, h, x0, x1, dd);

    f32x16 hid[8], acc[8];

    // layer 0: gamma(xyz) -> W (nerf.py:70-73)
    load_bias<8>(acc, bias_lds, 0, h);
    chunk_ktile8(pipe, cur, acc, x0);
    chunk_ktile8(pipe, cur, acc, x1);
    float top = activate<8, true>(hid, acc);      // +inf: an fp32 overflow somewhere in the trunk, in this lane's half of the point

    // trunk layers 1..D-1, then (with viewdirs) feature_linear as layer D without ReLU
    const int n_layers = a.use_viewdirs ? a.D + 1 : a.D;
    float sigma = 0.0f;
    for (int i = 1; i < n_layers; ++i) {
        const bool is_feature = (i == a.D);
        if (is_feature) {
            // alpha_linear reads the post-ReLU trunk output before feature_linear (nerf.py:86-89): one output
            // row, evaluated as a dot product over the 128 activations this lane holds + the other half-wave's
            sigma = row_dot<8>(hid, bias_lds, 8 * a.D + 14, h) + bias_lds[(8 * a.D) * 32];
            if (!(fmaxf(top, __shfl_xor(top, 32)) < __builtin_inff())) sigma = __builtin_nanf("");      // (either half of the point)
        }
        load_bias<8>(acc, bias_lds, is_feature ? 8 * a.D + 1 : 8 * i, h);
        // (training: tile kt of the previous layer's output - this chunk's B operand - goes to memory behind this chunk's
        // barrier)
        const RowRef keep = STORE ? row_ref(a.st.h[i - 1], a.st.h_ld[i - 1], pt, h) : RowRef{nullptr, 0u};
#pragma unroll
        for (int kt = 0; kt < 8; ++kt) {
            if constexpr (STORE)
                chunk_ktile8(pipe, cur, acc, hid[kt], [&]() { store_tile_at(keep, hid[kt], kt); });
            else
                chunk_ktile8(pipe, cur, acc, hid[kt]);
        }
        if (!is_feature && ((a.skip_in_mask >> i) & 1)) {
            // h = cat[input_pts, h] (nerf.py:79-80): the encoded inputs are still in registers
            chunk_ktile8(pipe, cur, acc, x0);
            chunk_ktile8(pipe, cur, acc, x1);
        }
        if (is_feature) {
            // the stream carries alpha_linear as an MFMA tile here for the fp16-pair kernel; this kernel has it from
            // row_dot above and only keeps the ring turning
            consume_chunk<8>(pipe, cur, [&](auto, auto, const Frag16&) {});
            top = fmaxf(top, activate<8, false>(hid, acc));
        } else {
            top = fmaxf(top, activate<8, true>(hid, acc));
        }
    }
    // without a view layer the last trunk output has no following chunks to ride behind
    if constexpr (STORE)
        if (!a.use_viewdirs) store_tiles<8>(a.st.h[a.D - 1], a.st.h_ld[a.D - 1], hid, pt, h, live);

    unsigned bad;      // NaN / Inf inputs propagate as through F.relu (mlp_inputs.h, kBadXyz): raw inputs re-read here
    {
        f32x16 t0, t1, t2;
        load_inputs<MODE, false, false>(a, pt, h, t0, t1, t2, nullptr, &bad);
    }
    const bool poisoned = !(fmaxf(top, __shfl_xor(top, 32)) < __builtin_inff());      // both halves of the point
    if (a.use_viewdirs) {
        // views_linears[0] on cat[feature, gamma(dir)] (nerf.py:93-98): 4 output tiles
        load_bias<4>(acc, bias_lds, 8 * a.D + 9, h);
        const RowRef keep = STORE ? row_ref(a.st.feat, a.st.feat_ld, pt, h) : RowRef{nullptr, 0u};
#pragma unroll
        for (int kp = 0; kp < 4; ++kp) {
            if constexpr (STORE)
                chunk_pair4(pipe, cur, acc, hid[2 * kp], hid[2 * kp + 1], [&]() {
                    store_tile_at(keep, hid[2 * kp], 2 * kp);
                    store_tile_at(keep, hid[2 * kp + 1], 2 * kp + 1);
                });
            else
                chunk_pair4(pipe, cur, acc, hid[2 * kp], hid[2 * kp + 1]);
        }
        chunk_ktile4(pipe, cur, acc, dd);
        activate<4, true>(hid, acc);
        if constexpr (STORE) store_tiles<4>(a.st.hv, a.st.hv_ld, hid, pt, h, live);
        // rgb_linear (nerf.py:101): three output rows over the 128-wide view layer, as dot products
        const float* rb = bias_lds + (8 * a.D + 13) * 32;
        const float r0 = row_dot<4>(hid, bias_lds, 8 * a.D + 22, h) + rb[0];
        const float r1 = row_dot<4>(hid, bias_lds, 8 * a.D + 26, h) + rb[1];
        const float r2 = row_dot<4>(hid, bias_lds, 8 * a.D + 30, h) + rb[2];
        if (live && h == 0) {
            // outputs = cat[rgb, alpha] (nerf.py:106)
            f32x4 o = {r0, r1, r2, sigma};
            if (bad || poisoned) {      // (poisoned: the trunk or feature_linear overflowed; sigma has its own mark above)
                const float qnan = __builtin_nanf("");
                o = f32x4{qnan, qnan, qnan, (bad & kBadXyz) ? qnan : sigma};
            }
            *(f32x4*)(a.out + pt * 4) = o;
        }
    } else {
        // output_linear (nerf.py:109): rows 0..out_ch-1 of one tile
        f32x16 o = *(const f32x16*)(bias_lds + ((8 * a.D) * 2 + h) * 16);
        chunk_row<8>(pipe, cur, o, hid);
        if (live) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
                if (row < a.out_ch) a.out[pt * a.out_ch + row] = ((bad & kBadXyz) || poisoned) ? __builtin_nanf("") : o[r];
            }
        }
    }
    }   // tile loop
    // the ring's trailing prefetches must land before this workgroup's LDS is released
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ---- fused backward-data pass (training) --------------------------------------------------
// The chain loss.backward() runs through the MLP, d raw -> d(pre-activation of every layer), in the forward kernel's
// orientation: d H_in^T [in x points] = W^T [in x out] * d Z^T, the transposed weights streamed by LDS-DMA
// (pack_backward_stream), the running gradient chained in registers from layer to layer. Per layer the only memory
// traffic is the ReLU mask (the post-ReLU activation the forward pass kept: 1 KB per point) and the masked gradient
// written for the dW GEMM (gemm_tn, dW = dZ^T X): 2 KB per point and layer against 131 kFLOP.
//   d(view pre)  = (d rgb . W_rgb) * [hv > 0]                       vector dot products, W_rgb rows from the bias block
//   d feature    = W_views[:, :W]^T d(view pre)                     4 chunks
//   d h_{D-1}    = W_feature^T d feature + d sigma * w_alpha        8 chunks + a rank-1 update
//   d z_i        = d h_i * [h_i > 0];  d h_{i-1} = W_i[:, hidden]^T d z_i      8 chunks per trunk layer
template <int N>
__device__ __forceinline__ void load_tiles(const float* base, int ld, f32x16 (&t)[8], int64_t pt, int h) {
    const float* row = base + (uint32_t)((uint32_t)pt * (uint32_t)ld + 4u * (uint32_t)h);
    const bool vec = ((ld & 3) == 0) && ((reinterpret_cast<uintptr_t>(base) & 15) == 0);   // wave-uniform
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float* p = row + 32 * i + 8 * q;
            if (vec) {
                const f32x4 v = *(const f32x4*)p;
                t[i][4 * q] = v[0];
                t[i][4 * q + 1] = v[1];
                t[i][4 * q + 2] = v[2];
                t[i][4 * q + 3] = v[3];
            } else {
                t[i][4 * q] = p[0];
                t[i][4 * q + 1] = p[1];
                t[i][4 * q + 2] = p[2];
                t[i][4 * q + 3] = p[3];
            }
        }
}
// g *= [post-ReLU activation > 0], the activation read from `base`
template <int N>
__device__ __forceinline__ void mask_tiles(f32x16 (&dst)[8], const f32x16 (&g)[8], const float* base, int ld, int64_t pt,
                                           int h) {
    const float* row = base + (uint32_t)((uint32_t)pt * (uint32_t)ld + 4u * (uint32_t)h);
    const bool vec = ((ld & 3) == 0) && ((reinterpret_cast<uintptr_t>(base) & 15) == 0);
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float* p = row + 32 * i + 8 * q;
            f32x4 v;
            if (vec) v = *(const f32x4*)p;
            else v = f32x4{p[0], p[1], p[2], p[3]};
#pragma unroll
            for (int e = 0; e < 4; ++e) dst[i][4 * q + e] = v[e] > 0.0f ? g[i][4 * q + e] : 0.0f;
        }
}
// one tile of the kept activation (feature columns 32 i .. 32 i + 31), requested early: see the backward kernel
__device__ __forceinline__ void load_tile(const float* base, int ld, f32x16& t, int i, int64_t pt, int h) {
    const float* row = base + (uint32_t)((uint32_t)pt * (uint32_t)ld + 4u * (uint32_t)h) + 32 * i;
    const bool vec = ((ld & 3) == 0) && ((reinterpret_cast<uintptr_t>(base) & 15) == 0);   // wave-uniform
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float* p = row + 8 * q;
        if (vec) {
            const f32x4 v = *(const f32x4*)p;
            t[4 * q] = v[0];
            t[4 * q + 1] = v[1];
            t[4 * q + 2] = v[2];
            t[4 * q + 3] = v[3];
        } else {
            t[4 * q] = p[0];
            t[4 * q + 1] = p[1];
            t[4 * q + 2] = p[2];
            t[4 * q + 3] = p[3];
        }
    }
}
// dst = g * [kept > 0], in the registers the activation was loaded into
__device__ __forceinline__ void mask_tile(f32x16& kept_then_dst, const f32x16& g) {
#pragma unroll
    for (int r = 0; r < 16; ++r) kept_then_dst[r] = kept_then_dst[r] > 0.0f ? g[r] : 0.0f;
}
// largest |value| of N tiles over the wavefront, into a float-bits slot (MlpBwdLaunch::maxes); post-ReLU values and
// magnitudes are non-negative, so the integer maximum of the bit patterns is the float maximum. The wave's maximum by six
// DPP folds (no LDS round trips), and the atomic only when the slot - read through the scalar cache, possibly stale, which
// only costs an atomic more - does not hold as much already: a thousand waves otherwise queue on one address per layer
// (+12 % on the kernel); after the first tiles almost none is issued.
template <int N>
__device__ __forceinline__ void track_max(unsigned* slot, const f32x16 (&t)[8]) {
    const unsigned known = *(const unsigned*)slot;
    float m = 0.0f;
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
        for (int r = 0; r < 16; r += 2) m = fmaxf(m, fmaxf(fabsf(t[i][r]), fabsf(t[i][r + 1])));
    int v = __float_as_int(m);
#define NERF_FOLD(ctrl, rows)                                                       \
    {                                                                               \
        const int o = __builtin_amdgcn_update_dpp(0, v, ctrl, rows, 0xf, false);    \
        v = o > v ? o : v;                                                          \
    }
    NERF_FOLD(0xB1, 0xf)      // quad_perm [1,0,3,2]
    NERF_FOLD(0x4E, 0xf)      // quad_perm [2,3,0,1]
    NERF_FOLD(0x141, 0xf)     // row_half_mirror
    NERF_FOLD(0x140, 0xf)     // row_mirror
    NERF_FOLD(0x142, 0xa)     // row_bcast:15 into rows 1 and 3
    NERF_FOLD(0x143, 0xc)     // row_bcast:31 into rows 2 and 3
#undef NERF_FOLD
    const unsigned top = (unsigned)__builtin_amdgcn_readlane(v, 63);
    if ((threadIdx.x & 63) == 0 && top > known && top < 0x7f800000u) atomicMax(slot, top);
}
template <int N>
__device__ __forceinline__ void zero_tiles(f32x16 (&t)[8]) {
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) t[i][r] = 0.0f;
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void nerf_mlp_bwd_kernel(const MlpBwdLaunch b) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* bias_lds = (float*)smem;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane >> 5;

    Pipe pipe{(const char*)b.stream, smem + kBiasLdsBytes, 0, 0, b.n_chunks, wave, lane};
    prefetch_chunk(pipe, 0, 0);
    prefetch_chunk(pipe, b.n_chunks > 1 ? 1 : 0, 1);
    for (int i = threadIdx.x; i < b.n_bias_tiles * kBiasTileFloats; i += 256) bias_lds[i] = b.bias[i];
    __syncthreads();
    Frag16 cur = read_frags(ring_frags(pipe, 0), 0);

    const int64_t n_tiles = (b.n_points + kPointsPerGroup - 1) / kPointsPerGroup;
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        pipe.c = 0;
        const int64_t pt_raw = tile * kPointsPerGroup + wave * kPointsPerWave + (lane & 31);
        const int64_t pt = pt_raw < b.n_points ? pt_raw : b.n_points - 1;
        const bool live = pt_raw < b.n_points;
        const float* dr = b.d_raw + pt * (b.d_raw_ld ? b.d_raw_ld : b.C);
        f32x16 hid[8], acc[8];
        if (b.use_viewdirs) {
            const float d0 = dr[0], d1 = dr[1], d2 = dr[2], dsig = dr[3];
            // d(view pre-activation): rgb_linear^T (3 rows, per-register weights: bias-block tiles 8D+22+4c+t) and the mask
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const f32x16 w0 = *(const f32x16*)(bias_lds + ((8 * b.D + 22 + t) * 2 + h) * 16);
                const f32x16 w1 = *(const f32x16*)(bias_lds + ((8 * b.D + 26 + t) * 2 + h) * 16);
                const f32x16 w2 = *(const f32x16*)(bias_lds + ((8 * b.D + 30 + t) * 2 + h) * 16);
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][r] = fmaf(d2, w2[r], fmaf(d1, w1[r], d0 * w0[r]));
            }
            mask_tiles<4>(hid, acc, b.fwd.hv, b.fwd.hv_ld, pt, h);
            if (b.maxes) track_max<4>(b.maxes + kBwdMaxViews, hid);

            // every gradient tile goes to memory behind the barrier of the chunk that contracts over it (see run_steps)
            // d feature = W_views[:, :W]^T d(view pre-activation)
            zero_tiles<8>(acc);
            {
                const RowRef out = row_ref(b.out.hv, b.out.hv_ld, pt, h);
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
                    chunk_ktile8(pipe, cur, acc, hid[kt], [&]() { store_tile_at(out, hid[kt], kt); });
            }
            activate<8, false>(hid, acc);
            if (b.maxes) track_max<8>(b.maxes + kBwdMaxFeat, hid);

            // d h_{D-1} = W_feature^T d feature + d sigma * w_alpha (alpha row: bias-block tiles 8D+14+t), then its mask
            // The ReLU mask of the layer below is the activation the forward pass kept: 1 KB per point, 128 KB per workgroup
            // and layer. Read at the layer boundary it stalls the matrix pipe for as long as HBM takes to deliver it (a fifth
            // of a layer's time, measured); instead tile kt of it is requested behind the barrier of chunk kt + 1, into the
            // registers of the gradient tile that chunk kt has just finished with, and only tile 7 is waited for in the open.
            zero_tiles<8>(acc);
            {
                const RowRef out = row_ref(b.out.feat, b.out.feat_ld, pt, h);
                const RowRef kept = row_ref(b.fwd.h[b.D - 1], b.fwd.h_ld[b.D - 1], pt, h);
#pragma unroll
                for (int kt = 0; kt < 8; ++kt)
                    chunk_ktile8(pipe, cur, acc, hid[kt], [&]() {
                        store_tile_at(out, hid[kt], kt);
                        if (kt >= 1) load_tile_at(kept, hid[kt - 1], kt - 1);
                    });
                load_tile_at(kept, hid[7], 7);
            }
            // (the stream carries the alpha row as an MFMA column here for the fp16-pair kernel; this kernel adds the rank-1 term
            // below and only keeps the ring turning)
            consume_chunk<8>(pipe, cur, [&](auto, auto, const Frag16&) {});
            if (b.maxes) track_max<8>(b.maxes + kBwdMaxKept + b.D - 1, hid);      // the kept h_{D-1}, before it becomes d z_{D-1}
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const f32x16 wa = *(const f32x16*)(bias_lds + ((8 * b.D + 14 + t) * 2 + h) * 16);
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][r] = fmaf(dsig, wa[r], acc[t][r]);
                mask_tile(hid[t], acc[t]);
            }
            if (b.maxes) track_max<8>(b.maxes + b.D - 1, hid);
        } else {
            // output_linear (nerf.py:109): d h_{D-1} = W_output^T d raw over the C <= kBwdMaxOutRows channels (rows per register:
            // bias-block tiles 8D+1+8c+t), masked by the kept h_{D-1}. (The stream carries the head as one MFMA chunk for the
            // fp16-pair kernel; this kernel only keeps the ring turning over it.)
            consume_chunk<8>(pipe, cur, [&](auto, auto, const Frag16&) {});
            zero_tiles<8>(acc);
            for (int c = 0; c < b.C; ++c) {
                const float dc = dr[c];
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    const f32x16 w = *(const f32x16*)(bias_lds + ((8 * b.D + 1 + 8 * c + t) * 2 + h) * 16);
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[t][r] = fmaf(dc, w[r], acc[t][r]);
                }
            }
            load_tiles<8>(b.fwd.h[b.D - 1], b.fwd.h_ld[b.D - 1], hid, pt, h);
            if (b.maxes) track_max<8>(b.maxes + kBwdMaxKept + b.D - 1, hid);
#pragma unroll
            for (int t = 0; t < 8; ++t) mask_tile(hid[t], acc[t]);
            if (b.maxes) track_max<8>(b.maxes + b.D - 1, hid);
        }

        // trunk: d h_{i-1} = W_i[:, hidden]^T d z_i, masked by layer i-1's ReLU
        for (int i = b.D - 1; i >= 1; --i) {
            zero_tiles<8>(acc);
            const RowRef out = row_ref(b.out.h[i], b.out.h_ld[i], pt, h);
            const RowRef kept = row_ref(b.fwd.h[i - 1], b.fwd.h_ld[i - 1], pt, h);
#pragma unroll
            for (int kt = 0; kt < 8; ++kt)
                chunk_ktile8(pipe, cur, acc, hid[kt], [&]() {
                    store_tile_at(out, hid[kt], kt);
                    if (kt >= 1) load_tile_at(kept, hid[kt - 1], kt - 1);
                });
            load_tile_at(kept, hid[7], 7);
            if (b.maxes) track_max<8>(b.maxes + kBwdMaxKept + i - 1, hid);
#pragma unroll
            for (int t = 0; t < 8; ++t) mask_tile(hid[t], acc[t]);
            if (b.maxes) track_max<8>(b.maxes + i - 1, hid);
        }
        store_tiles<8>(b.out.h[0], b.out.h_ld[0], hid, pt, h, live);     // d z_0: nothing left to ride behind
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

hipError_t launch_mlp_bwd(const MlpBwdLaunch& b, hipStream_t s) {
    if (b.n_points <= 0) return hipSuccess;
    if (b.n_chunks != (b.use_viewdirs ? 13 : 1) + 8 * (b.D - 1) || b.C < 4) return hipErrorInvalidValue;
    if (!b.use_viewdirs && (b.C > kBwdMaxOutRows || b.n_bias_tiles < 8 * b.D + 1 + 8 * b.C || b.D < 2)) return hipErrorInvalidValue;
    if (b.n_points > (int64_t)1 << 22) return hipErrorInvalidValue;      // 32-bit element offsets in load/store_tiles
    // the hooks inside the chunk loop are unconditional 16-byte accesses (RowRef)
    bool rows_ok = !b.use_viewdirs || (training_rows_ok(b.out.hv, b.out.hv_ld) && training_rows_ok(b.out.feat, b.out.feat_ld) &&
                                       training_rows_ok(b.fwd.hv, b.fwd.hv_ld));
    for (int i = 0; i < b.D; ++i) rows_ok = rows_ok && training_rows_ok(b.out.h[i], b.out.h_ld[i]) && training_rows_ok(b.fwd.h[i], b.fwd.h_ld[i]);
    if (!rows_ok) return hipErrorInvalidValue;
    const int64_t tiles = (b.n_points + kPointsPerGroup - 1) / kPointsPerGroup;
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
    const size_t lds = kBiasLdsBytes + kRing * kChunkBytes;
    static bool raised[64] = {};
    if (!raised[dev]) {
        e = hipFuncSetAttribute((const void*)nerf_mlp_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        raised[dev] = true;
    }
    hipLaunchKernelGGL(nerf_mlp_bwd_kernel, grid, block, lds, s, b);
    return hipGetLastError();
}

hipError_t launch_mlp(const MlpLaunch& a, int mode, hipStream_t s) {
    if (a.n_points <= 0) return hipSuccess;
    const int64_t tiles = (a.n_points + kPointsPerGroup - 1) / kPointsPerGroup;
    // persistent grid: the kernel's 112 KiB of LDS and 512 registers per lane admit exactly one
    // workgroup per CU, so one workgroup per CU walks the tiles
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
    const size_t lds = kBiasLdsBytes + kRing * kChunkBytes;
    static bool raised[64][4] = {};
    if (mode < 0 || mode > 2) return hipErrorInvalidValue;
    if (a.store && (mode != kInputRays || a.n_points > (int64_t)1 << 22)) return hipErrorInvalidValue;   // training forward: ray records; 32-bit element offsets in store_tiles
    if (a.store) {   // the hooks inside the chunk loop are unconditional 16-byte stores (RowRef)
        bool rows_ok = !a.use_viewdirs || (training_rows_ok(a.st.feat, a.st.feat_ld) && training_rows_ok(a.st.hv, a.st.hv_ld));
        for (int i = 0; i < a.D; ++i) rows_ok = rows_ok && training_rows_ok(a.st.h[i], a.st.h_ld[i]);
        if (!rows_ok) return hipErrorInvalidValue;
    }
    typedef void (*kernel_t)(const MlpLaunch);
    static const kernel_t table[4] = {nerf_mlp_kernel<kInputEmbedded>, nerf_mlp_kernel<kInputPoints>,
                                      nerf_mlp_kernel<kInputRays>, nerf_mlp_kernel<kInputRays, true>};
    static_assert(kInputEmbedded == 0 && kInputPoints == 1 && kInputRays == 2, "kernel table order");
    const int which = a.store ? 3 : mode;
    // 112 KiB of dynamic LDS is above the 64 KiB default cap: raise it once per device and kernel
    if (!raised[dev][which]) {
        e = hipFuncSetAttribute((const void*)table[which], hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        raised[dev][which] = true;
    }
    hipLaunchKernelGGL(table[which], grid, block, lds, s, a);
    return hipGetLastError();
}

}  // namespace nerf
