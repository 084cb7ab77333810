// nerf_mlp_h2a_kernel: the fp16-pair encode+MLP kernel (mlp_kernel_h2.hip: same arithmetic, same weight ring, same
// hand-placed step) with the register file allocated BY HAND for everything that lives across steps.
//
// Why: the three things hipcc could not be brought to do in mlp_kernel_h2.hip (profiles/r02_kernel_ab.md: eight forms tried,
// each spilling hundreds of registers) all need an allocation it does not find at 256 + 256 registers:
//   * a layer's output tile 0 converted inside the layer's OWN last chunk (it was converted in the open at the boundary),
//   * the finished sums leaving the accumulator file inside that chunk as well (16 v_accvgpr_read per step in its idle
//     vector slots instead of 128 in the open),
//   * one accumulator set instead of two.
// Here every matrix operand has a fixed home, named in the instruction text (templates compute the register numbers, the
// compiler never sees these values), and hipcc keeps what it is good at - addresses, the positional encoding, the scale
// bookkeeping, the epilogue - in the registers left to it (amdgpu_num_vgpr: v0..v127 incl. its SGPR spill lanes, a0..a31;
// tests/test_kernel_audit.py checks the generated code for any other use).
//
//   AGPR  a[  0: 31]  hipcc's (it parks vector registers there where its own 128 do not suffice: encoding, epilogue)
//         a[ 32:159]  the running layer's 8 accumulator tiles (32 features x 32 points each, 16 registers)
//         a[160:191]  A fragments of the current / next step (two buffers of 4 x ds_read_b128, step parity)
//         a[192:223]  gamma(xyz) as MFMA B operands (two tiles), kept for the skip layer
//         a[224:239]  gamma(dir) as B operands
//         a[240:255]  the one-row tile (alpha_linear; output_linear without view directions)
//   VGPR  v[128:255]  eight tile slots of 16: slot t holds tile t of the PENDING layer - first its raw sums (registers r
//                     and r + 8 = the two values of pair r), then, converted IN PLACE, the packed operands of the next
//                     layer: r = hi word of pair r, r + 8 = its lo word; hi words 0-3 | 4-7 and lo words 0-3 | 4-7 are the
//                     four B operands of the tile's two k-slices.
// The in-place conversion is what the pairing (r, r + 8) is for: mlp_kernel_h2.hip packs registers (2r, 2r + 1), whose
// words land on registers other pairs still have to read. The pairing only permutes the order of a k-slice's eight
// features, so the weight stream is re-cut to match (convert_stream_h2_kernel, `pairing` = 1).
//
// A layer now runs:  chunks 0..6   contract over tile c, convert pending tile c + 1 (slot c + 1, in place)
//                    last chunk    contract over tile 7 (or gamma(xyz); the alpha row for feature_linear); step s >= 1:
//                                  convert pair s - 1 of the layer's OWN tile 0 (read from accumulator tile 0) into slot 0, and copy
//                                  the own tile s (final since step s... read from step s + 1 on) to slot s
//                    boundary      close / open the scale bookkeeping; tile 7 is copied (its slot was in use)
// Hazards hipcc would have handled and this file handles itself (CDNA3 ISA 4.5): an MFMA result is read by
// v_accvgpr_read no earlier than two MFMA issues later (>= 11 wait states for an 8-pass MFMA); vector writes of an MFMA
// operand are at least a step away from the MFMA; the two half-register writes of v_fma_mix* have an instruction between.
#include "mlp_pair_common.h"

namespace nerf {

constexpr int kAcc = 32, kFragA = 160, kXp0 = 192, kXp1 = 208, kXd = 224, kRowT = 240;   // AGPR map (a[0:31]: hipcc's)
constexpr int kSlot = 128;                                                               // VGPR map: slot t = kSlot + 16 t

// ---- instruction emitters (register numbers are template constants) -----------------------------------------------
template <int ACC, int FRAG, bool B_AGPR, int B, bool ZERO>
__device__ __forceinline__ void mfma_at() {
    if constexpr (B_AGPR) {
        if constexpr (ZERO)
            asm volatile("v_mfma_f32_32x32x16_f16 a[%0:%1], a[%2:%3], a[%4:%5], 0" ::"n"(ACC), "n"(ACC + 15), "n"(FRAG),
                         "n"(FRAG + 3), "n"(B), "n"(B + 3));
        else
            asm volatile("v_mfma_f32_32x32x16_f16 a[%0:%1], a[%2:%3], a[%4:%5], a[%0:%1]" ::"n"(ACC), "n"(ACC + 15),
                         "n"(FRAG), "n"(FRAG + 3), "n"(B), "n"(B + 3));
    } else {
        if constexpr (ZERO)
            asm volatile("v_mfma_f32_32x32x16_f16 a[%0:%1], a[%2:%3], v[%4:%5], 0" ::"n"(ACC), "n"(ACC + 15), "n"(FRAG),
                         "n"(FRAG + 3), "n"(B), "n"(B + 3));
        else
            asm volatile("v_mfma_f32_32x32x16_f16 a[%0:%1], a[%2:%3], v[%4:%5], a[%0:%1]" ::"n"(ACC), "n"(ACC + 15),
                         "n"(FRAG), "n"(FRAG + 3), "n"(B), "n"(B + 3));
    }
}
// MFMA K (0..5) of a step, small terms first: (W_lo x_hi, W_hi x_lo, W_hi x_hi) for k-slice 0, then for k-slice 1.
// BUF: fragment buffer of this step; the tile at B_BASE is [hi 0 | hi 1 | lo 0 | lo 1], four registers each.
template <int K, int ACC, int BUF, bool B_AGPR, int B_BASE, bool FIRST>
__device__ __forceinline__ void mma_k() {
    constexpr int F = kFragA + 16 * BUF;
    if constexpr (K == 0) mfma_at<ACC, F + 4, B_AGPR, B_BASE + 0, FIRST>();
    else if constexpr (K == 1) mfma_at<ACC, F + 0, B_AGPR, B_BASE + 8, false>();
    else if constexpr (K == 2) mfma_at<ACC, F + 0, B_AGPR, B_BASE + 0, false>();
    else if constexpr (K == 3) mfma_at<ACC, F + 12, B_AGPR, B_BASE + 4, false>();
    else if constexpr (K == 4) mfma_at<ACC, F + 8, B_AGPR, B_BASE + 12, false>();
    else mfma_at<ACC, F + 8, B_AGPR, B_BASE + 4, false>();
}
template <int BUF, int Q, int OFF>
__device__ __forceinline__ void frag_issue_a(unsigned addr) {
    constexpr int r = kFragA + 16 * BUF + 4 * Q;
    asm volatile("ds_read_b128 a[%0:%1], %2 offset:%3" ::"n"(r), "n"(r + 3), "v"(addr), "n"(OFF) : "memory");
}

// finished sums of accumulator register R -> slot register R (same index), one instruction
template <int R>
__device__ __forceinline__ void readout_reg() {
    asm volatile("v_accvgpr_read_b32 v[%0], a[%1]" ::"n"(kSlot + R), "n"(kAcc + R));
}
template <int T, int LO, int HI>
__device__ __forceinline__ void readout_range() {
    if constexpr (LO < HI) {
        readout_reg<16 * T + LO>();
        readout_range<T, LO + 1, HI>();
    }
}

// ---- conversion of one register pair (values r and r + 8 of a tile), in place, in three slices ---------------------
// slice 0 (conv0a): y = max(sum * c + bias, floor), the sums from the slot (FROM_ACC: from accumulator tile T);
// slice 1 (conv1a): running maximum, scaling; slice 2 (conv2): the split, written over the sums
template <int T, int P>
__device__ __forceinline__ void conv2(const ConvTmp& t) {
    constexpr int hi = kSlot + 16 * T + P, lo = hi + 8;
    asm volatile("v_cvt_pk_f16_f32 v[%2], %0, %1\n\t"
                 "v_fma_mixlo_f16 v[%3], v[%2], -1.0, %0 op_sel_hi:[1,0,0]\n\t"
                 "s_nop 0\n\t"
                 "v_fma_mixhi_f16 v[%3], v[%2], -1.0, %1 op_sel:[1,0,0] op_sel_hi:[1,0,0]" ::"v"(t.a0),
                 "v"(t.a1), "n"(hi), "n"(lo));
}

struct Tile16 {
    f32x4 q[4];
};
struct PendingA {
    float c;       // raw sum -> activation: descale * 2^-t_in (per point)
    float floor;   // 0 for ReLU, -inf for feature_linear
    float sc;      // activation -> operand: 2^t_out (per point)
    int t_out;
    unsigned bias_addr;    // LDS address of this half-wave's bias entries of tile 0 (tile t: + 128 t)
    float m;       // running max |y|
};
// The bias block keeps a tile's 16 entries in accumulator-register order; pair P wants entries P and P + 8: one
// ds_read2_b32 (two dword offsets), so a step still issues one LDS operation per converted pair.
template <int DW>
__device__ __forceinline__ f32x2 bias_pair_issue(unsigned addr) {
    f32x2 r;
    asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=&v"(r) : "v"(addr), "n"(DW), "n"(DW + 8) : "memory");
    return r;
}
template <int T, int P, bool FROM_ACC>
__device__ __forceinline__ void conv0a(ConvTmp& t, const PendingA& pd, const f32x2& bb) {
    // (the maxima inside the statement: on a value that comes out of inline asm hipcc first issues a canonicalising
    // v_max_f32 v, v, v - two more vector instructions in a step that has none to spare)
    if constexpr (FROM_ACC) {
        asm volatile("v_accvgpr_read_b32 %0, a[%2]\n\tv_accvgpr_read_b32 %1, a[%3]\n\t"
                     "v_fma_f32 %0, %0, %4, %5\n\tv_fma_f32 %1, %1, %4, %6\n\t"
                     "v_max_f32 %0, %0, %7\n\tv_max_f32 %1, %1, %7"
                     : "=&v"(t.y0), "=&v"(t.y1)
                     : "n"(kAcc + 16 * T + P), "n"(kAcc + 16 * T + P + 8), "v"(pd.c), "v"(bb[0]), "v"(bb[1]), "v"(pd.floor));
    } else {
        asm volatile("v_fma_f32 %0, v[%2], %4, %5\n\tv_fma_f32 %1, v[%3], %4, %6\n\t"
                     "v_max_f32 %0, %0, %7\n\tv_max_f32 %1, %1, %7"
                     : "=&v"(t.y0), "=&v"(t.y1)
                     : "n"(kSlot + 16 * T + P), "n"(kSlot + 16 * T + P + 8), "v"(pd.c), "v"(bb[0]), "v"(bb[1]), "v"(pd.floor));
    }
}
__device__ __forceinline__ void conv1a(ConvTmp& t, PendingA& pd) {
    pd.m = fmaxf(fmaxf(pd.m, fabsf(t.y0)), fabsf(t.y1));
    t.a0 = t.y0 * pd.sc;
    t.a1 = t.y1 * pd.sc;
}
// a whole tile in the open (tile 1 before the view layer; every tile before output_linear): 16 bias entries, 8 pairs
template <int T, int P>
__device__ __forceinline__ void convert_pairs_open(PendingA& pd, const Tile16& b) {
    if constexpr (P < 8) {
        ConvTmp t;
        conv0a<T, P, false>(t, pd, f32x2{b.q[P >> 2][P & 3], b.q[(P + 8) >> 2][P & 3]});
        conv1a(t, pd);
        conv2<T, P>(t);
        convert_pairs_open<T, P + 1>(pd, b);
    }
}
__device__ __forceinline__ Tile16 lds_tile_issue_a(unsigned addr) {
    Tile16 t;
    asm volatile(
        "ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:16\n\tds_read_b128 %2, %4 offset:32\n\t"
        "ds_read_b128 %3, %4 offset:48"
        : "=&v"(t.q[0]), "=&v"(t.q[1]), "=&v"(t.q[2]), "=&v"(t.q[3])
        : "v"(addr)
        : "memory");
    return t;
}
__device__ __forceinline__ void lds_tile_wait_a(Tile16& t) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(t.q[0]), "+v"(t.q[1]), "+v"(t.q[2]), "+v"(t.q[3])::"memory");
}
template <int T>
__device__ __forceinline__ void convert_tile_open(PendingA& pd) {
    Tile16 b = lds_tile_issue_a(pd.bias_addr + 128 * T);
    lds_tile_wait_a(b);
    convert_pairs_open<T, 0>(pd, b);
}

// ---- operands that hipcc computes (the encodings) into their AGPR homes ---------------------------------------------
template <int R>
__device__ __forceinline__ void agpr_put(unsigned w) {
    asm volatile("v_accvgpr_write_b32 a[%0], %1" ::"n"(R), "v"(w));
}
template <int R>
__device__ __forceinline__ unsigned agpr_get() {
    unsigned w;
    asm volatile("v_accvgpr_read_b32 %0, a[%1]" : "=v"(w) : "n"(R));
    return w;
}
// 16 values (already in tile order) * sc -> packed (hi, lo) words of pairs (r, r + 8) at a[BASE + r], a[BASE + 8 + r]
template <int BASE, int P>
__device__ __forceinline__ void split_into(const f32x16& v, float sc) {
    if constexpr (P < 8) {
        const float a0 = v[P] * sc, a1 = v[P + 8] * sc;
        const h16x2 hi = round_pair(a0, a1);
        const h16x2 lo = {(_Float16)__builtin_fmaf((float)hi[0], -1.0f, a0), (_Float16)__builtin_fmaf((float)hi[1], -1.0f, a1)};
        agpr_put<BASE + P>(__builtin_bit_cast(unsigned, hi));
        agpr_put<BASE + 8 + P>(__builtin_bit_cast(unsigned, lo));
        split_into<BASE, P + 1>(v, sc);
    }
}
template <int BASE, int R>
__device__ __forceinline__ void rescale_at(h16x2 ff) {
    if constexpr (R < 16) {
        const unsigned w = agpr_get<BASE + R>();
        const h16x2 p = __builtin_bit_cast(h16x2, w) * ff;
        agpr_put<BASE + R>(__builtin_bit_cast(unsigned, p));
        rescale_at<BASE, R + 1>(ff);
    }
}
// multiply a split tile by 2^d (exact while nothing leaves the fp16 range)
template <int BASE>
__device__ __forceinline__ void rescale_tile_a(int d) {
    const _Float16 f = (_Float16)pow2f(d < -30 ? -30 : (d > 15 ? 15 : d));
    rescale_at<BASE, 0>(h16x2{f, f});
}

template <int S, int NSTEP>
__device__ __forceinline__ void dma_piece(PipeH& p) {
    constexpr int per = 8 / NSTEP;
    if constexpr (S < NSTEP / 2) prefetch_pieces<S * per, (S + 1) * per>(p.g_first, p.l_first);
    else prefetch_pieces<(S - NSTEP / 2) * per, (S - NSTEP / 2 + 1) * per>(p.g_second, p.l_second);
}

// ---- the step -------------------------------------------------------------------------------------------------------
// Same order and the same counted waits as run_steps (mlp_pair_common.h); fragments of step S sit in buffer S & 1 (every
// chunk has an even number of steps, so a chunk always starts on buffer 0).
template <int S, int NSTEP, int NB, class Body>
__device__ __forceinline__ void run_steps_a(PipeH& p, unsigned fr, unsigned fr_next, Body& body) {
    if constexpr (S < NSTEP) {
        constexpr bool last = S + 1 == NSTEP;
        constexpr int G = last ? 0 : (S + 1) * 4;
        constexpr int NXT = (S + 1) & 1;
        const unsigned ad = last ? fr_next : fr;
        NERF_FENCE();
        lgkm_wait<2 + NB>();
        NERF_FENCE();
        body(StepTag<S>{}, PartTag<0>{});
        NERF_FENCE();
        frag_issue_a<NXT, 0, (G + 0) * 1024>(ad);
        dma_piece<S, NSTEP>(p);
        NERF_FENCE();
        body(StepTag<S>{}, PartTag<1>{});
        NERF_FENCE();
        if constexpr (NB > 0) {
            lgkm_wait<1>();
            NERF_FENCE();
        }
        body(StepTag<S>{}, PartTag<11>{});
        frag_issue_a<NXT, 1, (G + 1) * 1024>(ad);
        NERF_FENCE();
        body(StepTag<S>{}, PartTag<2>{});
        NERF_FENCE();
        body(StepTag<S>{}, PartTag<12>{});
        frag_issue_a<NXT, 2, (G + 2) * 1024>(ad);
        NERF_FENCE();
        if constexpr (NB == 0) {
            lgkm_wait<3>();
            NERF_FENCE();
        }
        body(StepTag<S>{}, PartTag<3>{});
        NERF_FENCE();
        body(StepTag<S>{}, PartTag<13>{});
        frag_issue_a<NXT, 3, (G + 3) * 1024>(ad);
        NERF_FENCE();
        body(StepTag<S>{}, PartTag<4>{});
        NERF_FENCE();
        body(StepTag<S>{}, PartTag<14>{});
        NERF_FENCE();
        body(StepTag<S>{}, PartTag<5>{});
        NERF_FENCE();
        body(StepTag<S>{}, PartTag<15>{});
        NERF_FENCE();
        if constexpr (S == NSTEP / 2 - 1) {
            asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
            NERF_FENCE();
        }
        run_steps_a<S + 1, NSTEP, NB>(p, fr, fr_next, body);
    }
}
template <int NSTEP, int NB, class Body>
__device__ __forceinline__ void consume_chunk_a(PipeH& p, Body body) {
    const unsigned fr = lds_byte_addr(p.lds + p.b * kChunkBytes) + p.lane * 16;
    const unsigned fr_next = lds_byte_addr(p.lds + ringh_next(p.b, 1) * kChunkBytes) + p.lane * 16;
    const int c2 = p.c + 2 < p.n ? p.c + 2 : p.c + 2 - p.n;   // wraps into the next tile's stream
    const int c3 = p.c + 3 < p.n ? p.c + 3 : p.c + 3 - p.n;
    p.g_first = piece_src(p, c2) + 4096;
    p.l_first = piece_dst(p, ringh_next(p.b, 2)) + 4096;
    p.g_second = piece_src(p, c3);
    p.l_second = piece_dst(p, ringh_next(p.b, 3));
    run_steps_a<0, NSTEP, NB>(p, fr, fr_next, body);
    ++p.c;
    p.b = ringh_next(p.b, 1);
}

// accumulator tile 7 of the layer that has just ended -> slot 7, four registers behind the last MFMA of steps 0..3 of the
// next layer's first chunk (which overwrites accumulator tile 7 in its step 7 and needs slot 7 six chunks later)
template <int S, int PT>
__device__ __forceinline__ void read7_part() {
    if constexpr (PT == 15 && S < 4) readout_range<7, 4 * S, 4 * S + 4>();
}
// ---- chunk kinds ------------------------------------------------------------------------------------------------------
// k-tile chunk: acc tile s += W[s][k] x, x = the tile at (B_AGPR, B_BASE). CONV >= 0: step s converts pair s of pending
// tile CONV in place (bias entries requested one step earlier, two ds_read_b32: entries s and s + 8).
template <int CONV, bool FIRST, bool B_AGPR, int B_BASE, bool READ7 = false>
__device__ __forceinline__ void chunk_k8(PipeH& p, PendingA& pd) {
    constexpr int C0 = CONV < 0 ? 0 : CONV;
    f32x2 bb = {0.0f, 0.0f};
    ConvTmp t;
    if constexpr (CONV >= 0) bb = bias_pair_issue<32 * C0>(pd.bias_addr);
    consume_chunk_a<8, (CONV >= 0 ? 1 : 0)>(p, [&](auto tag, auto part) {
        constexpr int s = decltype(tag)::value, pt = decltype(part)::value;
        if constexpr (pt < 6) mma_k<pt, kAcc + 16 * s, s & 1, B_AGPR, B_BASE, FIRST>();
        else if constexpr (CONV >= 0) {
            if constexpr (pt == 11) conv0a<C0, s, false>(t, pd, bb);
            else if constexpr (pt == 12) conv1a(t, pd);
            else if constexpr (pt == 13) conv2<C0, s>(t);
            else if constexpr (pt == 14 && s < 7) bb = bias_pair_issue<32 * C0 + s + 1>(pd.bias_addr);
        }
        if constexpr (READ7) read7_part<s, pt>();
    });
}
// The layer's LAST accumulating chunk, or the alpha chunk behind feature_linear's: besides its MFMAs it converts the layer's
// own tile 0 out of the accumulator file into slot 0 - pairs 0 and 1 in step 1 (tile 0 is final after step 0), pair s in
// step s = 2..7 - and from step 2 on copies the own tile s - 1 (final after step s - 1; its slot holds an operand tile no
// later step reads) to its slot, three to six registers behind each MFMA. Tile 7 is copied by the NEXT chunk (READ7).
// Bias entries: pair 0 before the chunk, pair s + 1 in step s - the pattern of every converting chunk (NB = 1).
struct OwnA {
    f32x2 e, o;                // bias entries (P, P + 8) of an even / an odd pair
    ConvTmp t, u;
};
template <int S, int PT>
__device__ __forceinline__ void own_part(OwnA& oc, PendingA& pd) {
    if constexpr (S == 1) {
        if constexpr (PT == 11) {
            conv0a<0, 0, true>(oc.t, pd, oc.e);
            conv0a<0, 1, true>(oc.u, pd, oc.o);
        } else if constexpr (PT == 12) {
            conv1a(oc.t, pd);
            conv1a(oc.u, pd);
        } else if constexpr (PT == 13) {
            conv2<0, 0>(oc.t);
            conv2<0, 1>(oc.u);
        }
    } else if constexpr (S >= 2) {
        if constexpr (PT == 11) conv0a<0, S, true>(oc.t, pd, (S & 1) ? oc.o : oc.e);
        else if constexpr (PT == 12) conv1a(oc.t, pd);
        else if constexpr (PT == 13) conv2<0, S>(oc.t);
        if constexpr (PT == 12) readout_range<S - 1, 0, 3>();
        else if constexpr (PT == 13) readout_range<S - 1, 3, 5>();
        else if constexpr (PT == 14) readout_range<S - 1, 5, 10>();
        else if constexpr (PT == 15) readout_range<S - 1, 10, 16>();
    }
    if constexpr (PT == 14 && S < 7) {
        if constexpr ((S + 1) & 1) oc.o = bias_pair_issue<S + 1>(pd.bias_addr);
        else oc.e = bias_pair_issue<S + 1>(pd.bias_addr);
    }
}
template <bool B_AGPR, int B_BASE>
__device__ __forceinline__ void chunk_k8_own(PipeH& p, PendingA& pd) {
    OwnA oc;
    oc.e = bias_pair_issue<0>(pd.bias_addr);
    consume_chunk_a<8, 1>(p, [&](auto tag, auto part) {
        constexpr int s = decltype(tag)::value, pt = decltype(part)::value;
        if constexpr (pt < 6) mma_k<pt, kAcc + 16 * s, s & 1, B_AGPR, B_BASE, false>();
        else own_part<s, pt>(oc, pd);
    });
}
// one output row over the 8 operand tiles in the slots, into a[kRowT ..]: step s = tile s
template <bool OWN>
__device__ __forceinline__ void chunk_row8_a(PipeH& p, PendingA& pd) {
    OwnA oc;
    if constexpr (OWN) oc.e = bias_pair_issue<0>(pd.bias_addr);
    consume_chunk_a<8, (OWN ? 1 : 0)>(p, [&](auto tag, auto part) {
        constexpr int s = decltype(tag)::value, pt = decltype(part)::value;
        if constexpr (pt < 6) {
            if constexpr (s == 0) mma_k<pt, kRowT, s & 1, false, kSlot + 16 * s, true>();
            else mma_k<pt, kRowT, s & 1, false, kSlot + 16 * s, false>();
        } else if constexpr (OWN) own_part<s, pt>(oc, pd);
    });
    asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");   // the row's last MFMA -> the read of its result
}
// view layer: two k-tiles against 4 output tiles (steps 0-3 tile X0, 4-7 tile X0 + 1), converting pending tiles CONV, CONV + 1
template <int CONV, bool FIRST, int X0, bool READ7 = false>
__device__ __forceinline__ void chunk_pair4_a(PipeH& p, PendingA& pd) {
    constexpr int C0 = CONV < 0 ? 0 : CONV;
    f32x2 bb = {0.0f, 0.0f}, cc = {0.0f, 0.0f};
    ConvTmp t0, t1;
    if constexpr (CONV >= 0) {
        bb = bias_pair_issue<32 * C0>(pd.bias_addr);
        cc = bias_pair_issue<32 * (C0 + 1)>(pd.bias_addr);
    }
    consume_chunk_a<8, (CONV >= 0 ? 2 : 0)>(p, [&](auto tag, auto part) {
        constexpr int s = decltype(tag)::value, pt = decltype(part)::value;
        if constexpr (pt < 6) {
            if constexpr (s < 4) mma_k<pt, kAcc + 16 * (s & 3), s & 1, false, kSlot + 16 * X0, FIRST>();
            else mma_k<pt, kAcc + 16 * (s & 3), s & 1, false, kSlot + 16 * (X0 + 1), false>();
        } else if constexpr (CONV >= 0) {
            if constexpr (pt == 11) {
                conv0a<C0, s, false>(t0, pd, bb);
                conv0a<C0 + 1, s, false>(t1, pd, cc);
            } else if constexpr (pt == 12) {
                conv1a(t0, pd);
                conv1a(t1, pd);
            } else if constexpr (pt == 13) {
                conv2<C0, s>(t0);
                conv2<C0 + 1, s>(t1);
            } else if constexpr (pt == 14 && s < 7) {
                bb = bias_pair_issue<32 * C0 + s + 1>(pd.bias_addr);
                cc = bias_pair_issue<32 * (C0 + 1) + s + 1>(pd.bias_addr);
            }
        }
        if constexpr (READ7) read7_part<s, pt>();
    });
}
// gamma(dir) against the view layer's 4 output tiles
__device__ __forceinline__ void chunk_k4_a(PipeH& p) {
    consume_chunk_a<4, 0>(p, [&](auto tag, auto part) {
        constexpr int s = decltype(tag)::value, pt = decltype(part)::value;
        if constexpr (pt < 6) mma_k<pt, kAcc + 16 * s, s & 1, true, kXd, false>();
    });
}

// accumulator tile T -> sixteen compiler registers (the epilogue's arithmetic is hipcc's)
template <int BASE, int R>
__device__ __forceinline__ void acc_fetch(f32x16& y) {
    if constexpr (R < 16) {
        float v;
        asm volatile("v_accvgpr_read_b32 %0, a[%1]" : "=v"(v) : "n"(BASE + R));
        y[R] = v;
        acc_fetch<BASE, R + 1>(y);
    }
}

template <int BASE, int R>
__device__ __forceinline__ void acc_fetch(f32x16& y);
// max with the partner lane of the other half-wave (inline asm: see half_max in mlp_kernel_h2.hip)
__device__ __forceinline__ float half_max_a(float m) {
    float x = m, y = m;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(x), "+v"(y));
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
    return r;
}
// the view layer's output tile T: y = relu(acc * c + bias), and its share of rgb_linear's three rows (weights per register
// in the bias block, tiles 8D + 22 + 4 row + T), summed in the order of row_dot4 (mlp_kernel_h2.hip)
template <int T>
__device__ __forceinline__ void epilogue_tile(float (&s0)[3], float (&s1)[3], unsigned bias0, int D, float cv) {
    f32x16 y;
    acc_fetch<kAcc + 16 * T, 0>(y);
    Tile16 b = lds_tile_issue_a(bias0 + 128 * (8 * D + 9 + T));
    lds_tile_wait_a(b);
#pragma unroll
    for (int r = 0; r < 16; ++r) y[r] = fmaxf(fmaf(y[r], cv, b.q[r >> 2][r & 3]), 0.0f);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        Tile16 w = lds_tile_issue_a(bias0 + 128 * (8 * D + 22 + 4 * c + T));
        lds_tile_wait_a(w);
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            s0[c] = fmaf(w.q[r >> 2][r & 3], y[r], s0[c]);
            s1[c] = fmaf(w.q[(r + 1) >> 2][(r + 1) & 3], y[r + 1], s1[c]);
        }
    }
}

// ---- the kernel -------------------------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_vgpr(128)))   // (with amdgpu_waves_per_eu(1, 1) the request is dropped)
void nerf_mlp_h2a_kernel(const MlpLaunch a) {
    // registers beyond hipcc's own: the kernel descriptor must cover them (accum_offset 256, 512 in all)
    asm volatile("; hand-allocated: v[128:255], a[32:255]" ::: "v255", "a255");
    extern __shared__ __attribute__((aligned(16))) char ring_lds[];
    __shared__ __attribute__((aligned(16))) float bias_lds[kBiasLdsBytes / 4];
    __shared__ __attribute__((aligned(16))) float layer_tab[4 * (kMaxDepth + 3)];   // per layer [descale, gain, max|b|, -]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane >> 5;

    PipeH pipe{(const char*)a.stream_h2, ring_lds, 0, 0, a.n_chunks, wave, lane, nullptr, nullptr, nullptr, nullptr};
    for (int k = 0; k < 2; ++k) {
        prefetch_pieces<0, 4>(piece_src(pipe, k), piece_dst(pipe, k));
        prefetch_pieces<0, 4>(piece_src(pipe, k) + 4096, piece_dst(pipe, k) + 4096);
    }
    prefetch_pieces<0, 4>(piece_src(pipe, 2), piece_dst(pipe, 2));   // chunk 0's first-half steps issue the other four
    for (int i = threadIdx.x; i < a.n_bias_tiles * kBiasTileFloats; i += 256) bias_lds[i] = a.bias[i];
    if (threadIdx.x < a.D + 3) {
        const int l = threadIdx.x;
        const bool has_gain = l <= (a.use_viewdirs ? a.D : a.D - 1);
        layer_tab[4 * l] = a.descale[l];
        layer_tab[4 * l + 1] = has_gain ? a.gain[2 * l] : 0.0f;
        layer_tab[4 * l + 2] = has_gain ? a.gain[2 * l + 1] : 0.0f;
        layer_tab[4 * l + 3] = 0.0f;
    }
    __syncthreads();   // chunks 0, 1, the bias block and the layer tables are in LDS
    {
        const unsigned fr0 = lds_byte_addr(ring_lds) + lane * 16;
        frag_issue_a<0, 0, 0>(fr0);
        frag_issue_a<0, 1, 1024>(fr0);
        frag_issue_a<0, 2, 2048>(fr0);
        frag_issue_a<0, 3, 3072>(fr0);
    }

    const unsigned bias0 = lds_byte_addr(bias_lds) + 64 * h;   // this half-wave's entries of bias-block tile 0
    const int n_layers = a.use_viewdirs ? a.D + 1 : a.D;
    const int64_t n_tiles = (a.n_points + kPointsPerGroup - 1) / kPointsPerGroup;
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        pipe.c = 0;
        const int64_t tile0 = tile * kPointsPerGroup + wave * kPointsPerWave;
        const int64_t pt_raw = tile0 + (lane & 31);
        const int64_t pt = pt_raw < a.n_points ? pt_raw : a.n_points - 1;

        float m_pe, m_dir = 0.0f;
        int t_pe;
        unsigned bad;
        {
            f32x16 x0, x1, dd;
            float m_dd;
            load_inputs<MODE, true, true>(a, pt, h, x0, x1, dd, &m_dd, &bad);
            m_dir = wave_max(m_dd);
            // the ranges of the encoded inputs are taken over the whole wavefront (wave-uniform: SGPRs)
            m_pe = wave_max(tile_absmax(x1, tile_absmax(x0, 0.0f)));
            t_pe = pick_exponent(m_pe);
            split_into<kXp0, 0>(x0, pow2f(t_pe));
            split_into<kXp1, 0>(x1, pow2f(t_pe));
            // gamma(dir) is parked unscaled (scale 2^0: |gamma(dir)| <= max(|d|, 1)); the view layer brings it to its scale
            split_into<kXd, 0>(dd, 1.0f);
        }
        const unsigned long long bad_xyz = __ballot((bad & kBadXyz) != 0), bad_dir = __ballot((bad & kBadDir) != 0);

        PendingA pd;
        float sigma = 0.0f;
        float m_prev = 0.0f;

        // what the raw sums of layer l become. m_in = largest |input| of layer l (true units), t_in = exponent its inputs
        // were scaled by
        auto make_pending = [&](int l, float m_in, int t_in) {
            const bool is_feature = a.use_viewdirs && l == a.D;
            const f32x4 tab = lds_vec4(layer_tab + 4 * l);
            pd.c = tab[0] * pow2f(-t_in);
            pd.floor = is_feature ? -__builtin_inff() : 0.0f;
            float bound = fmaf(tab[1], m_in, tab[2]) * 1.001f;
            // the next layer may concatenate these outputs with inputs that must fit the same scale
            if (is_feature) bound = fmaxf(bound, m_dir);
            else if ((a.skip_in_mask >> (l + 1)) & 1) bound = fmaxf(bound, m_pe);
            pd.t_out = pick_exponent(bound);
            pd.sc = pow2f(pd.t_out);
            pd.bias_addr = bias0 + 128 * (is_feature ? 8 * a.D + 1 : 8 * l);
            pd.m = 0.0f;
        };
        auto close_pending = [&]() {
            m_prev = half_max_a(pd.m);
            const int slack = 10 - pd.t_out - __builtin_amdgcn_frexp_expf(m_prev);
            if (m_prev > 0.0f && slack >= 12 && pd.t_out > -60 && a.loose) atomicAdd(a.loose, 1u);
        };

        // layer 0: gamma(xyz) -> W (nerf.py:70-73); its second chunk converts its own tile 0 and copies the others out
        make_pending(0, m_pe, t_pe);
        asm volatile("s_nop 4" ::: "memory");            // the encodings' v_accvgpr_write -> first MFMA
        chunk_k8<-1, true, true, kXp0>(pipe, pd);
        chunk_k8_own<true, kXp1>(pipe, pd);

        // trunk layers 1..D-1, then (with viewdirs) feature_linear as layer D without ReLU
        for (int l = 1; l < n_layers; ++l) {
            const bool is_feature = a.use_viewdirs && l == a.D;
            const bool skip = !is_feature && ((a.skip_in_mask >> l) & 1);
            chunk_k8<1, true, false, kSlot + 16 * 0, true>(pipe, pd);
            chunk_k8<2, false, false, kSlot + 16 * 1>(pipe, pd);
            chunk_k8<3, false, false, kSlot + 16 * 2>(pipe, pd);
            chunk_k8<4, false, false, kSlot + 16 * 3>(pipe, pd);
            chunk_k8<5, false, false, kSlot + 16 * 4>(pipe, pd);
            chunk_k8<6, false, false, kSlot + 16 * 5>(pipe, pd);
            chunk_k8<7, false, false, kSlot + 16 * 6>(pipe, pd);
            close_pending();                                     // all 8 tiles of the pending layer are converted
            const int t_in = pd.t_out;
            make_pending(l, skip ? fmaxf(m_prev, m_pe) : m_prev, t_in);
            if (is_feature) {
                // alpha_linear reads the post-ReLU trunk output (nerf.py:86), i.e. this layer's input: one more chunk, a
                // single-row tile; it also does this layer's own conversions and copies
                chunk_k8<-1, false, false, kSlot + 16 * 7>(pipe, pd);
                chunk_row8_a<true>(pipe, pd);
                sigma = fmaf(__uint_as_float(agpr_get<kRowT>()), lds_scalar(layer_tab + 4 * (a.D + 2)) * pow2f(-t_in),
                             lds_scalar(bias_lds + (8 * a.D) * 32));
            } else if (skip) {
                // h = cat[input_pts, h] (nerf.py:79-80): bring the encoded inputs to this layer's scale
                chunk_k8<-1, false, false, kSlot + 16 * 7>(pipe, pd);
                rescale_tile_a<kXp0>(t_in - t_pe);
                rescale_tile_a<kXp1>(t_in - t_pe);
                t_pe = t_in;
                asm volatile("s_nop 4" ::: "memory");
                chunk_k8<-1, false, true, kXp0>(pipe, pd);
                chunk_k8_own<true, kXp1>(pipe, pd);
            } else {
                chunk_k8_own<false, kSlot + 16 * 7>(pipe, pd);
            }
        }

        const bool live = pt_raw < a.n_points;
        if (a.use_viewdirs) {
            // views_linears[0] on cat[feature, gamma(dir)] (nerf.py:93-98): 4 output tiles; the pending layer is
            // feature_linear (tile 0 converted by its last chunk)
            convert_tile_open<1>(pd);
            chunk_pair4_a<2, true, 0, true>(pipe, pd);
            chunk_pair4_a<4, false, 2>(pipe, pd);
            chunk_pair4_a<6, false, 4>(pipe, pd);
            chunk_pair4_a<-1, false, 6>(pipe, pd);
            close_pending();
            rescale_tile_a<kXd>(pd.t_out);
            asm volatile("s_nop 4" ::: "memory");
            chunk_k4_a(pipe);
            asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
            // y = relu(acc * c + bias) tile by tile, and rgb_linear's three rows over it (nerf.py:101)
            const float cv = lds_scalar(layer_tab + 4 * (a.D + 1)) * pow2f(-pd.t_out);
            float s0[3] = {0.0f, 0.0f, 0.0f}, s1[3] = {0.0f, 0.0f, 0.0f};
            epilogue_tile<0>(s0, s1, bias0, a.D, cv);
            epilogue_tile<1>(s0, s1, bias0, a.D, cv);
            epilogue_tile<2>(s0, s1, bias0, a.D, cv);
            epilogue_tile<3>(s0, s1, bias0, a.D, cv);
            const float* rb = bias_lds + (8 * a.D + 13) * 32;
            float r[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float s = s0[c] + s1[c];
                r[c] = s + __shfl_xor(s, 32) + lds_scalar(rb + c);
            }
            if (live && h == 0) {
                f32x4 o = {r[0], r[1], r[2], sigma};   // outputs = cat[rgb, alpha] (nerf.py:106)
                const bool bx = (bad_xyz >> lane) & 1, bd = (bad_dir >> lane) & 1;
                if (bx || bd) {                          // NaN / Inf inputs propagate as through F.relu (see kBadXyz)
                    const float qnan = __builtin_nanf("");
                    o = f32x4{qnan, qnan, qnan, bx ? qnan : sigma};
                }
                *(f32x4*)(a.out + pt * 4) = o;
            }
        } else {
            // output_linear (nerf.py:109): rows 0..out_ch-1 of one tile; the pending layer is trunk layer D-1
            asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");   // tile 7's last MFMA -> its copy: 11 wait states
            readout_range<7, 0, 16>();
            convert_tile_open<1>(pd);
            convert_tile_open<2>(pd);
            convert_tile_open<3>(pd);
            convert_tile_open<4>(pd);
            convert_tile_open<5>(pd);
            convert_tile_open<6>(pd);
            convert_tile_open<7>(pd);
            chunk_row8_a<false>(pipe, pd);
            f32x16 o;
            acc_fetch<kRowT, 0>(o);
            Tile16 b = lds_tile_issue_a(bias0 + 128 * (8 * a.D));
            lds_tile_wait_a(b);
            const float c = lds_scalar(layer_tab + 4 * a.D) * pow2f(-pd.t_out);
            if (live) {
                const bool bx = (bad_xyz >> lane) & 1;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
                    if (row < a.out_ch)
                        a.out[pt * a.out_ch + row] = bx ? __builtin_nanf("") : fmaf(o[r], c, b.q[r >> 2][r & 3]);
                }
            }
        }
    }   // tile loop
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

hipError_t launch_mlp_h2a(const MlpLaunch& a, int mode, hipStream_t s) {
    if (a.n_points <= 0) return hipSuccess;
    if (!a.stream_h2 || !a.descale || !a.gain) return hipErrorInvalidValue;
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
    // rows of encoded inputs (mode 0) stay with mlp_kernel_h2.hip: hipcc's share of that instantiation does not fit a0..a31
    if (mode != kInputPoints && mode != kInputRays) return hipErrorInvalidValue;
    typedef void (*kernel_t)(const MlpLaunch);
    static const kernel_t table[3] = {nullptr, nerf_mlp_h2a_kernel<kInputPoints>, nerf_mlp_h2a_kernel<kInputRays>};
    static_assert(kInputEmbedded == 0 && kInputPoints == 1 && kInputRays == 2, "kernel table order");
    if (!raised[dev][mode]) {
        e = hipFuncSetAttribute((const void*)table[mode], hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        raised[dev][mode] = true;
    }
    hipLaunchKernelGGL(table[mode], grid, block, lds, s, a);
    return hipGetLastError();
}

}  // namespace nerf
