// Machinery of the fp16-pair MLP kernel (mlp_kernel_h2.hip, v_mfma_f32_32x32x16_f16): the LDS-DMA weight ring, the hand-placed step with its counted LDS waits, the power-of-two
// scaling and the exact (hi, lo) fp16 split, LDS reads outside hipcc's LDS-DMA guard.
#pragma once
#include "mlp_inputs.h"

namespace nerf {

typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// one 32-feature activation tile as MFMA B operands: two u32x4 of packed halves each for hi and lo ([k-slice])
struct XT {
    u32x4 hi[2], lo[2];
};
#ifdef NERF_STAMPS
// diagnostic build: wave 0 of workgroup 0 records (tag, s_memtime) through its first tile. A sample is taken at the
// start of a step and stored at the start of the NEXT one, so that reading the counter (an SMEM return) adds no wait
// of its own.
struct Stamper {
    unsigned long long* buf;
    unsigned long long t_prev;
    int tag_prev;
    int n;
    bool on;
};
__device__ __forceinline__ void stamp(Stamper& st, int tag) {
    if (st.on) {
        if (st.n < 4090 && (threadIdx.x & 63) == 0) {
            st.buf[2 * st.n] = (unsigned long long)st.tag_prev;
            st.buf[2 * st.n + 1] = st.t_prev;
        }
        ++st.n;
        st.t_prev = __builtin_amdgcn_s_memtime();
        st.tag_prev = tag;
    }
}
#define STAMP(p, tag) stamp((p).st, (tag))
#else
#define STAMP(p, tag) do {} while (0)
#endif

// ---- weight-stream pipeline -------------------------------------------------------------
// Four 32 KiB LDS buffers form a ring: while chunk c is consumed, chunk c+1 is resident, chunk c+2 half issued and
// chunk c+3 about to be. A chunk travels as 8 LDS-DMA pieces per wave (1 KiB each), one per step: the first-half steps
// of chunk c issue pieces 4..7 of chunk c+2, the second-half steps pieces 0..3 of chunk c+3. One barrier per chunk,
// mid-chunk:
//   vmcnt(8)  -> this wave's share of chunk c+1 has landed (only chunk c+2's 8 pieces may be pending)
//   s_barrier -> every wave's share has, and every wave has finished chunk c-1, whose buffer chunk c+3 takes.
constexpr int kRingH = 4;

struct PipeH {
    const char* stream;
    char* lds;
    int c, b, n, wave, lane;
    const char* g_first;
    const char* g_second;
    char* l_first;
    char* l_second;
    // carried from chunk to chunk (consume_chunk): what a chunk needs beyond its predecessor's values is one ring slot and
    // one stream position - recomputing every address from (c, b) was 27 instructions between the last MFMA of a chunk and
    // the first of the next, with a single MFMA left in the matrix pipe to cover them
    unsigned fr;            // LDS address of this lane's fragments in the current chunk's slot
    const char* g2;         // this lane's source of chunk c + 2 and of chunk c + 3 (middle of the wave's share)
    const char* g3;         // the stream is followed by a copy of its first three chunks (launch_convert_stream_h2): within a
                            // tile the positions only advance, into the head of the next tile's stream; a tile resets them
    int c3;
#ifdef NERF_STAMPS
    Stamper st;
#endif
};

__device__ __forceinline__ int ringh_next(int b, int k) {
    b += k;
    return b >= kRingH ? b - kRingH : b;
}
__device__ __forceinline__ const char* piece_src(const PipeH& p, int chunk) {
    return p.stream + (size_t)chunk * kChunkBytes + p.wave * 8192 + p.lane * 16;
}
__device__ __forceinline__ char* piece_dst(const PipeH& p, int slot) { return p.lds + slot * kChunkBytes + p.wave * 8192; }

template <int J>
__device__ __forceinline__ void prefetch_piece(const char* g, char* l) {
    __builtin_amdgcn_global_load_lds(GLB_PTR(g), LDS_PTR(l), 16, J * 1024, 0);
}
template <int LO, int HI>
__device__ __forceinline__ void prefetch_pieces(const char* g, char* l) {
    if constexpr (LO < HI) {
        prefetch_piece<LO>(g, l);
        prefetch_pieces<LO + 1, HI>(g, l);
    }
}

// A-fragments of one step: [T0 hi, T0 lo, T1 hi, T1 lo]
struct Frag4 {
    f32x4 q[4];
};

// ---- hand-placed step ------------------------------------------------------------------------------------------
// Every LDS read of a step is its own inline-asm statement in its own MFMA gap and the waits are counted by hand (LDS
// returns in order; hipcc would guard reads it can see with vmcnt(0) while an LDS-DMA is in flight, and four
// ds_read_b128 issued together stall the wave, profiles/microbench/step_mix.hip). LDS operations of a step, in issue
// order:   q0' (after MFMA pair 0)  q1' (after 1)  q2' (after 2)  q3' (after 3)  then NB bias reads (after pair 4)
// so that
//     pair 0 wants q1 (and pair 1, 2 q0) of this step: newer are q2, q3 and the NB bias reads -> lgkmcnt(2 + NB)
//     the conversion wants the bias reads: newer is q0'                                       -> lgkmcnt(1), covers q2, q3
//     without bias reads pair 3 wants q3: newer are q0', q1', q2'                              -> lgkmcnt(3)
// A count that is too small only waits longer; the pattern is kept across chunk boundaries (a chunk's last step issues
// no bias read, the next chunk issues its first ones before its step 0), and everything else that reads LDS between
// chunks waits for lgkmcnt(0).
template <int OFF>
__device__ __forceinline__ void frag_issue(f32x4& q, unsigned addr) {
#ifdef NERF_FRAG_VGPR
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(q) : "v"(addr), "n"(OFF) : "memory");
#else
    // into the accumulator half of the register file: an MFMA takes its A operand from there as well, and the 32
    // registers of the double-buffered fragments are 32 vector registers the conversion does not have to fight for
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&a"(q) : "v"(addr), "n"(OFF) : "memory");
#endif
}
template <int N>
__device__ __forceinline__ void lgkm_wait() {
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
}
#define NERF_FENCE() __builtin_amdgcn_sched_barrier(0)

template <int S>
struct StepTag {
    static constexpr int value = S;
};
template <int P>
struct PartTag {
    static constexpr int value = P;
};

// body(step, part, frags): part 0..5 = that MFMA pair; part 11, 12, 13 = the vector work placed behind pairs 1, 2, 3;
// part 14 = the bias requests for the next step, behind pair 4
template <int S, int NSTEP, int NB, class Body>
__device__ __forceinline__ void run_steps(PipeH& p, Frag4& cur, unsigned fr, unsigned fr_next, Body& body) {
    if constexpr (S < NSTEP) {
        constexpr bool last = S + 1 == NSTEP;
        constexpr int G = last ? 0 : (S + 1) * 4;
        const unsigned ad = last ? fr_next : fr;
        Frag4 nxt;
        NERF_FENCE();
        lgkm_wait<2 + NB>();
        NERF_FENCE();
        body(StepTag<S>{}, PartTag<0>{}, cur);
        NERF_FENCE();
        STAMP(p, (p.c << 8) | (S << 4) | NB);
        frag_issue<(G + 0) * 1024>(nxt.q[0], ad);
#ifndef NERF_ABLATE_DMA
        {
            constexpr int per = 8 / NSTEP;
            if constexpr (S < NSTEP / 2) prefetch_pieces<S * per, (S + 1) * per>(p.g_first, p.l_first);
            else prefetch_pieces<(S - NSTEP / 2) * per - 4, (S - NSTEP / 2 + 1) * per - 4>(p.g_second, p.l_second);
        }
#endif
        NERF_FENCE();
        body(StepTag<S>{}, PartTag<1>{}, cur);
        NERF_FENCE();
        if constexpr (NB > 0) {
            lgkm_wait<1>();
            NERF_FENCE();
        }
        body(StepTag<S>{}, PartTag<11>{}, cur);
        frag_issue<(G + 1) * 1024>(nxt.q[1], ad);
        NERF_FENCE();
        body(StepTag<S>{}, PartTag<2>{}, cur);
        NERF_FENCE();
        body(StepTag<S>{}, PartTag<12>{}, cur);
        frag_issue<(G + 2) * 1024>(nxt.q[2], ad);
        NERF_FENCE();
        if constexpr (NB == 0) {
            lgkm_wait<3>();
            NERF_FENCE();
        }
        body(StepTag<S>{}, PartTag<3>{}, cur);
        NERF_FENCE();
        body(StepTag<S>{}, PartTag<13>{}, cur);
        frag_issue<(G + 3) * 1024>(nxt.q[3], ad);
        NERF_FENCE();
        body(StepTag<S>{}, PartTag<4>{}, cur);
        NERF_FENCE();
        body(StepTag<S>{}, PartTag<14>{}, cur);
        NERF_FENCE();
        body(StepTag<S>{}, PartTag<5>{}, cur);
        NERF_FENCE();
        cur = nxt;
        if constexpr (S == NSTEP / 2 - 1) {
#if defined(NERF_ABLATE_BARRIER)      // timing-only builds: what the chunk's synchronisation costs (profiles/r02_kernel_ab.md)
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
#elif defined(NERF_ABLATE_VMWAIT)
            asm volatile("s_barrier" ::: "memory");
#else
#ifdef NERF_EXP_VMCNT      // timing experiment only (profiles/r02_kernel_ab.md): UNSAFE for the ring
#ifdef NERF_EXP_VMCNT_ALL
            if constexpr (true)
#else
            if constexpr (NB > 0)
#endif
                asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(NERF_EXP_VMCNT) : "memory");
            else
#endif
            asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
#endif
            NERF_FENCE();
        }
        run_steps<S + 1, NSTEP, NB>(p, cur, fr, fr_next, body);
    }
}

__device__ __forceinline__ unsigned lds_byte_addr(const void* p) {
    return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)p;
}

__device__ __forceinline__ void pipe_tile_start(PipeH& p) {
    p.g2 = piece_src(p, 2) + 4096;      // (biased by half a wave's share: see consume_chunk)
    p.g3 = piece_src(p, 3) + 4096;
}
__device__ __forceinline__ void pipe_start(PipeH& p) {
    static_assert(kRingH == 4, "ring slots are advanced with & 3");
    p.b = 0;
    p.fr = lds_byte_addr(p.lds) + p.lane * 16;
    pipe_tile_start(p);
}
template <int NSTEP, int NB, class Body>
__device__ __forceinline__ void consume_chunk(PipeH& p, Frag4& cur, Body body) {
    const int b1 = (p.b + 1) & 3, b2 = (p.b + 2) & 3, b3 = (p.b + 3) & 3;
    const unsigned fr = p.fr;
    const unsigned fr_next = lds_byte_addr(p.lds) + p.lane * 16 + b1 * kChunkBytes;
    // both halves of a wave's share are addressed from its MIDDLE: pieces 4..7 at immediate offsets 0..3 KiB, pieces 0..3 at
    // -4..-1 KiB (the offset field is signed, 13 bits, and moves the LDS address along) - no pointer arithmetic per chunk
    p.g_first = p.g2;
    p.l_first = piece_dst(p, b2) + 4096;
    p.g_second = p.g3;
    p.l_second = piece_dst(p, b3) + 4096;
    run_steps<0, NSTEP, NB>(p, cur, fr, fr_next, body);
#ifdef NERF_STAMPS
    ++p.c;
#endif
    p.b = b1;
    p.fr = fr_next;
    p.g2 = p.g3;
    p.g3 += kChunkBytes;
}

// ---- per-point scaling and the fp16 split --------------------------------------------------
// exponent t such that max * 2^t lies in [2^9, 2^10): headroom of 64 below the fp16 maximum
__device__ __forceinline__ int pick_exponent(float m) {
    const int t = 10 - __builtin_amdgcn_frexp_expf(m);   // frexp_exp(0) = 0
    return t < -60 ? -60 : (t > 60 ? 60 : t);   // keeps descale * 2^-t finite
}
__device__ __forceinline__ float pow2f(int t) { return __builtin_ldexpf(1.0f, t); }

// largest value over the wavefront (of non-negative values), as a wave-uniform number (lives in an SGPR). Six DPP steps
// (within quads, half rows, rows, then row_bcast:15 / :31 into the last row) and a v_readlane instead of six
// ds_bpermute round trips through the LDS crossbar with a wait each: non-negative floats order like their bit patterns.
__device__ __forceinline__ float wave_max(float m) {
    int v = __builtin_bit_cast(int, m);
    auto fold = [&](auto ctrl, auto rows) {
        const int t = __builtin_amdgcn_update_dpp(0, v, decltype(ctrl)::value, decltype(rows)::value, 0xf, false);
        v = t > v ? t : v;
    };
    fold(PartTag<0xB1>{}, PartTag<0xf>{});     // quad_perm [1,0,3,2]
    fold(PartTag<0x4E>{}, PartTag<0xf>{});     // quad_perm [2,3,0,1]
    fold(PartTag<0x141>{}, PartTag<0xf>{});    // row_half_mirror
    fold(PartTag<0x140>{}, PartTag<0xf>{});    // row_mirror: every lane of a row holds the row's maximum
    fold(PartTag<0x142>{}, PartTag<0xa>{});    // row_bcast:15 into rows 1 and 3
    fold(PartTag<0x143>{}, PartTag<0xc>{});    // row_bcast:31 into rows 2 and 3
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(v, 63));
}
__device__ __forceinline__ float tile_absmax(const f32x16& v, float m) {
#pragma unroll
    for (int r = 0; r < 16; ++r) m = fmaxf(m, fabsf(v[r]));
    return m;
}

// v_cvt_pk_f16_f32: both halves rounded to nearest even. hi = rn16(v) leaves |v - hi| <= 2^-11 |v| (exact in fp32, up
// to 12 significant bits), lo = rn16(v - hi) keeps 11 of them: |v - hi - lo| <= 2^-23 |v|, at most one fp32 ulp.
__device__ __forceinline__ h16x2 round_pair(float a, float b) {
    const f32x2 v = {a, b};
    return __builtin_convertvector(v, h16x2);
}
// values 2 Pp, 2 Pp + 1 of a tile in [point group][position] order (already scaled) -> packed (hi, lo)
template <int Pp>
__device__ __forceinline__ void split_pair(XT& out, float a, float b) {
    const h16x2 hi = round_pair(a, b);
    const h16x2 lo = {(_Float16)__builtin_fmaf((float)hi[0], -1.0f, a), (_Float16)__builtin_fmaf((float)hi[1], -1.0f, b)};
    out.hi[Pp >> 2][Pp & 3] = __builtin_bit_cast(unsigned, hi);
    out.lo[Pp >> 2][Pp & 3] = __builtin_bit_cast(unsigned, lo);
}
template <int Pp>
__device__ __forceinline__ void split_pairs(XT& out, const f32x16& v, float sc) {
    if constexpr (Pp < 8) {
        split_pair<Pp>(out, v[2 * Pp] * sc, v[2 * Pp + 1] * sc);
        split_pairs<Pp + 1>(out, v, sc);
    }
}
__device__ __forceinline__ void split_tile(XT& out, const f32x16& v, float sc) { split_pairs<0>(out, v, sc); }

// ---- LDS reads outside hipcc's LDS-DMA guard ---------------------------------------------------------------
// hipcc guards every LDS load it can see with s_waitcnt vmcnt(0) while an LDS-DMA write is in flight (it cannot tell the
// bias block from the ring), which would drain the weight pipeline at every bias read; these reads are issued from
// inline asm with their own lgkmcnt wait (LDS returns in order).
__device__ __forceinline__ float lds_scalar(const float* p) {
    float v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(lds_byte_addr(p)) : "memory");
    return v;
}
__device__ __forceinline__ f32x4 lds_vec4(const float* p) {
    f32x4 v;
    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(lds_byte_addr(p)) : "memory");
    return v;
}
template <int OFF>
__device__ __forceinline__ f32x2 lds_pair_issue(unsigned addr) {
    f32x2 r;
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=&v"(r) : "v"(addr), "n"(OFF) : "memory");
    return r;
}

// the last slice of a register pair's conversion: (a0, a1), already scaled -> packed (hi, lo), lo = rn16(a - hi) by one fma
// with an fp16 source and an fp16 result per half (hipcc does not form these itself). One statement: the two
// half-register writes want an instruction between them.
struct ConvTmp {
    float y0, y1, a0, a1;
};
template <int P>
__device__ __forceinline__ void conv_slice2(XT& dst, const ConvTmp& t) {
#ifdef NERF_ABLATE_CONV
    if (P == 0) dst.hi[0][0] = __float_as_uint(t.y0);
    return;
#endif
    unsigned hi, lo;
    asm("v_cvt_pk_f16_f32 %0, %2, %3\n\t"
        "v_fma_mixlo_f16 %1, %0, -1.0, %2 op_sel_hi:[1,0,0]\n\t"
        "s_nop 0\n\t"
        "v_fma_mixhi_f16 %1, %0, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
        : "=&v"(hi), "=&v"(lo)
        : "v"(t.a0), "v"(t.a1));
    dst.hi[P >> 2][P & 3] = hi;
    dst.lo[P >> 2][P & 3] = lo;
}

// ---- shared by the forward (mlp_kernel_h2.hip) and backward-data (mlp_bwd_kernel_h2.hip) kernels -------------------
__device__ __forceinline__ f32x16 mma(const f32x4& a, const u32x4& b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, a), __builtin_bit_cast(h16x8, b), c, 0, 0, 0);
}

// MFMA K (0..5) of a step, small terms first
template <int K, bool FIRST>
__device__ __forceinline__ void mma_one(f32x16& acc, const Frag4& f, const XT& x) {
#ifdef NERF_MMA_SHARED_OPERANDS
    // neighbours share an operand: (q0,lo0) (q0,hi0) (q1,hi0) | (q3,hi1) (q2,hi1) (q2,lo1)
    constexpr int A[6] = {0, 0, 1, 3, 2, 2};
    constexpr bool LO[6] = {true, false, false, false, false, true};
#else
    constexpr int A[6] = {1, 0, 0, 3, 2, 2};
    constexpr bool LO[6] = {false, true, false, false, true, false};
#endif
    constexpr int s = K / 3;
    const u32x4& b = LO[K] ? x.lo[s] : x.hi[s];
    if constexpr (K == 0 && FIRST) {
        const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        acc = mma(f.q[A[K]], b, zero);
    } else {
        acc = mma(f.q[A[K]], b, acc);
    }
}

// max with the partner lane of the other half-wave: v_permlane32_swap on two copies yields (lo, lo) and (hi, hi)
// - a vector instruction, no trip through the LDS crossbar and no lgkmcnt wait. From inline asm: through
// __builtin_amdgcn_permlane32_swap(u, u, ...) hipcc folds max(r[0], r[1]) to r[0] (it takes the two results of a swap
// of equal inputs for equal), which silently made every lane use the LOWER half-wave's value - harmless while both
// halves of a point have similar maxima, an fp16 overflow (NaN) when one row of the upper half dominates
// (tests: test_mlp_precisions_vs_fp64, "one row x2^16").
__device__ __forceinline__ float half_max(float m) {
    float a = m, b = m;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// ---- LDS reads outside hipcc's LDS-DMA guard ---------------------------------------------------------------
// hipcc guards every LDS load it can see with s_waitcnt vmcnt(0) while an LDS-DMA write is in flight (it cannot
// tell the bias block from the ring), which would drain the weight pipeline at every bias read; these reads are
// issued from inline asm with their own lgkmcnt wait (LDS returns in order, and the waits hipcc computes for its
// own reads can only become stricter by the extra entries).
__device__ __forceinline__ unsigned lds_addr(const float* p) {
    return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) float*)p;
}
struct Tile16 {
    f32x4 q[4];
};
__device__ __forceinline__ Tile16 lds_tile_issue(unsigned addr) {
    Tile16 t;
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

// a pointer the compiler cannot prove wave-uniform (picked by a loop variable out of the launch record) as one that is:
// the "s" operands of the stores below must be scalar registers
template <class P>
__device__ __forceinline__ P* wave_uniform(P* p) {
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (P*)(((unsigned long long)hi << 32) | lo);
}

// Running maxima for the weight-gradient kernel's scales (MlpBwdLaunch::maxes: float bits, non-negative, so the integer
// maximum is the float maximum). A lane hands its point's maximum to an LDS atomic on the workgroup's record - one
// instruction, nothing returned, nothing waited for - and the workgroup enters its records into the global slots when it
// ends (flush_maxes). Earlier forms and what they cost the training forward pass: a C++ atomicMax per wave and layer (a
// thousand waves queue on one address); a wave reduction by DPP folds plus a compare against the slot read through the
// scalar cache (6 % of the pass: all of it in the open at a layer boundary); the same with a C++ atomicMax on a
// reconstructed pointer (a FLAT atomic behind s_waitcnt vmcnt(0): the weight ring drained at every boundary, 2x).
__device__ __forceinline__ void enter_max(unsigned* record_lds, float m_point) {
#ifdef NERF_ABLATE_ENTER_MAX
    return;
#endif
    asm volatile("ds_max_u32 %0, %1" : : "v"(lds_byte_addr(record_lds)), "v"(__float_as_uint(m_point)) : "memory");
}
// at the end of the kernel, behind a barrier: thread t enters record t
__device__ __forceinline__ void flush_maxes(unsigned* global_slots, const unsigned* record_lds, int n) {
    const int t = threadIdx.x;
    if (global_slots && t < n) {
        const unsigned v = record_lds[t];
        if (v != 0u && v < 0x7f800000u) atomicMax(global_slots + t, v);
    }
}

// ---- what the training kernels keep ----------------------------------------------------------------------------
// 16 bytes of a row-major [points, channels] buffer in ONE instruction: wave-uniform base, one 32-bit byte offset per
// lane, the rest an immediate (the launchers bound the buffers). (s_nop 1: a store of more than 8 bytes reads its data
// registers late - two wait states before a vector instruction may overwrite them on gfx950; hipcc's hazard recogniser
// does not look inside inline asm.)
template <int IMM>
__device__ __forceinline__ void keep_quad(float* base, unsigned off, const f32x4& v) {
    asm volatile("global_store_dwordx4 %0, %1, %2 offset:%3\n\ts_nop 1" : : "v"(off), "v"(v), "s"(base), "n"(IMM) : "memory");
}
// cache-policy bits of the blocked layout's stores (timing experiments set others: profiles/r04_ab_notes.txt)
#if defined(NERF_STORE_BITS_SEL) && NERF_STORE_BITS_SEL == 1
#define NERF_STORE_BITS "sc1 nt"
#elif defined(NERF_STORE_BITS_SEL) && NERF_STORE_BITS_SEL == 2
#define NERF_STORE_BITS "sc0 nt"
#elif defined(NERF_STORE_BITS_SEL) && NERF_STORE_BITS_SEL == 3
#define NERF_STORE_BITS "sc1"
#elif defined(NERF_STORE_BITS_SEL) && NERF_STORE_BITS_SEL == 4
#define NERF_STORE_BITS "sc0 sc1 nt"
#else
#define NERF_STORE_BITS "nt"
#endif
// the same with the nt bit, for stores that write whole lines of data nobody reads soon (the layout blocked by 32 points)
template <int IMM>
__device__ __forceinline__ void keep_quad_nt(float* base, unsigned off, const f32x4& v) {
#ifdef NERF_EXP_NOSTORE      // timing experiments (profiles/r04_ab_notes.txt)
    asm volatile("" ::"v"(v), "v"(off), "s"(base));
    return;
#endif
    asm volatile("global_store_dwordx4 %0, %1, %2 offset:%3 " NERF_STORE_BITS "\n\ts_nop 1" : : "v"(off), "v"(v), "s"(base), "n"(IMM) : "memory");
}
template <int IMM>
__device__ __forceinline__ void keep_word(unsigned* base, unsigned off, unsigned v) {
    asm volatile("global_store_dword %0, %1, %2 offset:%3" : : "v"(off), "v"(v), "s"(base), "n"(IMM) : "memory");
}

// ReLU masks, one bit per unit (MlpStore::mask). The forward kernel's conversion hooks meet a tile's values pair by pair;
// `hi` is the pair's packed fp16 high halves (non-negative behind a ReLU), so "unit active" is "half non-zero": a packed
// 16-bit minimum with 1 turns both into flags, and the word collects them by shifting: after the 16 pairs of two tiles,
// value 2s of tile T sits at bit 15 - 8 (T & 1) - s and value 2s + 1 at bit 31 - 8 (T & 1) - s. (A unit whose value is
// below 2^-25 of its point's scale rounds to a zero half and counts as inactive: its pre-activation is zero to 2^-35 of
// the layer's range, where the reference's own sign is rounding noise.)
__device__ __forceinline__ unsigned mask_push(unsigned field, unsigned hi) {
    unsigned t;
    asm("v_pk_min_u16 %0, %1, %2" : "=v"(t) : "v"(hi), "s"(0x00010001u));
    return (field << 1) | t;
}
// g where the unit's bit is set, else 0: value 2s (ODD = 0) / 2s + 1 (ODD = 1) of tile T, from the word of tiles 2 (T / 2), + 1.
// Two instructions from inline asm (the bit sign-extended to a mask, then an AND): from C++ hipcc makes it an AND, a
// compare into vcc, a wait state and a select.
template <int T, int S, int ODD>
__device__ __forceinline__ float mask_apply(unsigned word, float g) {
    constexpr int bit = (ODD ? 31 : 15) - 8 * (T & 1) - S;
    float y;
    asm("v_bfe_i32 %0, %1, %2, 1\n\tv_and_b32 %0, %0, %3" : "=&v"(y) : "v"(word), "n"(bit), "v"(g));
    return y;
}

}  // namespace nerf
