// Fused positional-encoding + NeRF MLP kernel, "fp16-pair" arithmetic (NERF_PRECISION_F16X2).
//
// Same job and same structure as mlp_kernel.hip (Embedder.embed nerf/embedder.py:72-80, the viewdir
// broadcast + concat of run_network nerf.ipynb:827-843, NeRF.forward nerf/nerf.py:57-111; a wavefront
// owns 32 points, activations stay feature-major in registers, weights stream L2 -> LDS by LDS-DMA),
// but the contraction runs on the half-precision matrix pipe, which on gfx950 has 16x the rate of
// v_mfma_f32_32x32x2_f32:
//
//   * every fp32 operand v is carried as two fp16 numbers, v ~= hi + lo with hi = rn16(v) and
//     lo = rn16(v - hi): |v - hi - lo| <= 2^-23 |v| (the remainder has up to 12 significant bits, lo keeps 11: at
//     most one fp32 ulp is lost). A product W*x is the three v_mfma_f32_32x32x16_f16 terms
//     W_lo*x_hi + W_hi*x_lo + W_hi*x_hi (fp16 x fp16 is exact in the fp32 accumulator); the dropped W_lo*x_lo is
//     <= 2^-22 |W x| per product. Per product that is ~4x the fp32 rounding unit; measured through the 8x256 network
//     against an fp64 evaluation the result is as close as the fp32 MFMA chain's, or closer (the errors are unbiased
//     and small next to the accumulated rounding of a 256-term fp32 sum): tests test_mlp_precisions_vs_fp64;
//     profiles/microbench/mfma_bf16_split.hip has the single-layer numbers and the schemes that were ruled out.
//   * fp16 has 5 exponent bits, so both operands are kept in range by exact power-of-two scalings:
//     each layer's weights by one factor chosen from the layer's largest |w| (convert kernel below; the
//     inverse factor travels in `descale`), each POINT's activation vector by its own factor, chosen
//     after every layer from an a-priori bound of that point's outputs (see Pending). A point is a
//     column of the MFMA, so its factor is a per-lane multiplier folded into the fma that adds the
//     bias: no activation of any magnitude overflows.
//   * the bias is added, ReLU applied and the operands re-split in fp32 on the accumulator values, one
//     output tile per chunk of the NEXT layer, in the shadow of its MFMAs; alpha_linear is one more MFMA
//     tile, rgb_linear an fp32 dot product on the last layer's values.
//
// Cost per layer and wave: 384 MFMAs of 32 cycles against 1024 of 64.
#include "mlp_pair_common.h"

namespace nerf {

// multiply a split tile by 2^d (exact while nothing leaves the fp16 range)
__device__ __forceinline__ void rescale_tile(XT& x, int d) {
    const _Float16 f = (_Float16)pow2f(d < -30 ? -30 : (d > 15 ? 15 : d));
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

// ---- the layer whose raw sums wait to become the next layer's operands -------------------------------------
// A layer's 8 accumulator tiles are not converted when the layer ends but one tile per chunk of the NEXT layer
// (which accumulates into the other accumulator set), one register pair per MFMA step, in the shadow of the
// MFMAs. That needs the output scale before the outputs exist: it is chosen from the bound
//   |y_j| <= max_j sum_k |W_jk| * max_k |x_k| + max_j |b_j|      (gain table, launch_layer_gains)
// which is loose by 2^4..2^5 on NeRF weights; the split tolerates 2^11 (the low halves of values more than 2^-2
// below the scaled maximum go subnormal with absolute error 2^-25, i.e. 2^-35 of the 2^10 the bound maps to).
struct Pending {
    float c;       // raw sum -> activation: descale * 2^-t_in (per point)
    float floor;   // 0 for ReLU, -inf for feature_linear
    float sc;      // activation -> operand: 2^t_out (per point)
    int t_out;
    unsigned bias_addr;    // LDS address of this half-wave's bias entries of tile 0 (tile t: + 128 t)
    float m;       // running max |y|
    float* keep_base;      // STORE kernels: the buffer that keeps this layer's activations for the backward pass ...
    unsigned keep_off;     // ... and this lane's BYTE offset of (its point, feature 4 h) in it
    f32x2 even;            // (convert_pair: the even pair of a register quad, until the odd one completes the 16 bytes)
    unsigned* mask_base;   // STORE kernels: this layer's ReLU-mask record (MlpStore::mask), this lane's BYTE offset in it,
    unsigned mask_off;     // the word being collected (mask_push) and the record's finished words: the record goes out as
    unsigned maskw;        // ONE 16-byte store when the layer is closed (a wave then writes a contiguous KiB; word by word,
    u32x4 maskq;           // 4-byte pieces of a line reached memory microseconds apart - partial-line writes - and cost
                           // the pass a fifth of its time)
};

// Training forward (STORE kernels): register pair P of tile T of the pending layer - features 32 T + 8 (P/2) + 2 (P%2) + 4 h
// + {0, 1} of this lane's point, true units - goes to the row-major [points, channels] buffer autograd would keep
// (MlpStore; the layout of mlp_kernel.hip's store_tile_at). Wave-uniform base, one 32-bit byte offset per lane (the
// launcher bounds the buffers), the rest an immediate: ONE instruction in a step whose issue slots are nearly all taken.
// Lanes past the end recompute the last point and write the same values to the same address.
template <int T, int P>
__device__ __forceinline__ void keep_pair(const Pending& pd, float y0, float y1) {
    const f32x2 v = {y0, y1};
    asm volatile("global_store_dwordx2 %0, %1, %2 offset:%3"
                 :
                 : "v"(pd.keep_off), "v"(v), "s"(pd.keep_base), "n"((32 * T + 8 * (P >> 1) + 2 * (P & 1)) * 4)
                 : "memory");
}
// two pairs (a register quad: four consecutive features) in one 16-byte store: half the instructions and half the write
// requests of keep_pair.
// STORE = 1: row-major [points, channels]: the 16 bytes at (point, feature 32 T + 8 Q + 4 h): a store instruction covers 32
// rows x 32 bytes - partial lines, completed by the three other quads of the tile steps later.
// STORE = 2: BLOCKED by 32 points (MlpStore::blocked): the buffer is [point / 32][tile T][quad Q][32 points][2 half-waves]
// [4 features], so the instruction writes ONE contiguous KiB - eight whole lines - and carries the nt bit (written once, read
// once by the weight-gradient kernel after gigabytes of other traffic: it need not displace the weight stream from L2).
// keep_off is the lane's place inside a piece; keep_base points at the start of the ODD tile of the pair being converted, so
// that the signed 13-bit immediate reaches both tiles (1024 Q - 4096 for the even one), and moves on by two tiles when the odd
// one is done (next_tile_pair: two scalar adds in place - a base per tile spilled scalar registers into vector ones and those
// into scratch, which tools/audit_lds_waits.py and the no-scratch rule of tests/test_kernel_audit.py both refuse).
// Measured with a timing-only build before any consumer could read the layout (profiles/r04_store_layout_ab.txt): forward
// 1.36 -> 0.93 ms per iteration, backward-data 1.01 -> 0.69; the nt bit on the row-major form DOUBLES both (partial lines
// written through).
template <int T, int Q, int STORE>
__device__ __forceinline__ void keep_pairs(const Pending& pd, const f32x2& even, float y0, float y1) {
    const f32x4 v = {even[0], even[1], y0, y1};
#ifdef NERF_EXP_NOSTORE      // timing experiments (profiles/r02_kernel_ab.md)
    asm volatile("" ::"v"(v));
    return;
#endif
    // (s_nop 1: a store of more than 8 bytes reads its data registers late - two wait states before a vector instruction
    // may overwrite them on gfx950; hipcc's hazard recogniser does not look inside inline asm)
    if constexpr (STORE == 2) {
        asm volatile("global_store_dwordx4 %0, %1, %2 offset:%3 " NERF_STORE_BITS "\n\ts_nop 1"
                     :
                     : "v"(pd.keep_off), "v"(v), "s"(pd.keep_base), "n"(1024 * Q - ((T & 1) ? 0 : 4096))
                     : "memory");
    } else {
        asm volatile("global_store_dwordx4 %0, %1, %2 offset:%3\n\ts_nop 1"
                     :
                     : "v"(pd.keep_off), "v"(v), "s"(pd.keep_base), "n"((32 * T + 8 * Q) * 4)
                     : "memory");
    }
}
// blocked stores: tiles 2 k, 2 k + 1 are done, the base moves to the next pair's odd tile
template <int STORE>
__device__ __forceinline__ void next_tile_pair(Pending& pd) {
    if constexpr (STORE == 2) pd.keep_base += 2048;
}
template <int P, int STORE = 0, int T = 0>
__device__ __forceinline__ void convert_pair(XT& dst, const f32x16& src, Pending& pd, const f32x2& b) {
#ifdef NERF_ABLATE_CONV
    if (P == 0) dst.hi[0][0] = __float_as_uint(src[0] + b[0]);
    return;
#endif
    const float y0 = fmaxf(fmaf(src[2 * P], pd.c, b[0]), pd.floor);
    const float y1 = fmaxf(fmaf(src[2 * P + 1], pd.c, b[1]), pd.floor);
    if constexpr (STORE) {
        if constexpr ((P & 1) == 0) pd.even = f32x2{y0, y1};
        else keep_pairs<T, (P >> 1), STORE>(pd, pd.even, y0, y1);
    }
    pd.m = fmaxf(fmaxf(pd.m, fabsf(y0)), fabsf(y1));
    const float a0 = y0 * pd.sc, a1 = y1 * pd.sc;
    const unsigned hi = __builtin_bit_cast(unsigned, round_pair(a0, a1));
    // lo = rn16(a - hi): one fma with an fp16 source and an fp16 result per half (hipcc does not form these itself)
    unsigned lo;
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(lo) : "v"(hi), "v"(a0));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lo) : "v"(hi), "v"(a1));
    dst.hi[P >> 2][P & 3] = hi;
    dst.lo[P >> 2][P & 3] = lo;
#ifndef NERF_ABLATE_MASKS
    if constexpr (STORE) {
        pd.maskw = mask_push(pd.maskw, hi);
        if constexpr (P == 7 && (T & 1) == 1) {      // tiles 2w, 2w + 1 done: word w of the record
            pd.maskq[T >> 1] = pd.maskw;
            pd.maskw = 0u;
        }
    }
#endif
}

// a whole tile at once (not hidden: tile 0 at the start of a layer)
template <int P, int STORE = 0, int T = 0>
__device__ __forceinline__ void convert_pairs(XT& dst, const f32x16& src, Pending& pd, const Tile16& b) {
    if constexpr (P < 8) {
        convert_pair<P, STORE, T>(dst, src, pd, f32x2{b.q[P >> 1][2 * (P & 1)], b.q[P >> 1][2 * (P & 1) + 1]});
        convert_pairs<P + 1, STORE, T>(dst, src, pd, b);
    }
}
template <int T, int STORE = 0>
__device__ __forceinline__ void convert_tile(XT& dst, const f32x16& src, Pending& pd) {
    Tile16 b = lds_tile_issue(pd.bias_addr + 128 * T);
    lds_tile_wait(b);
    convert_pairs<0, STORE, T>(dst, src, pd, b);
}
// tile 0 with its bias entries already fetched by make_pending, together with its scale-table row: the two LDS latencies
// of a layer boundary overlap instead of adding up (-0.5 % per launch). Requested earlier still - before the layer's last
// chunk - hipcc parks the in-flight destination registers in AGPRs (tools/audit_lds_waits.py rejects the build).
template <int STORE = 0>
__device__ __forceinline__ void convert_tile0_with(XT& dst, const f32x16& src, Pending& pd, Tile16& b) {
    convert_pairs<0, STORE, 0>(dst, src, pd, b);
}

// convert_pair cut into the three slices that ride behind MFMAs 1, 2 and 3 of a step
template <int P>
__device__ __forceinline__ void conv_slice0(ConvTmp& t, const f32x16& src, const Pending& pd, const f32x2& b) {
#ifdef NERF_ABLATE_CONV   // timing-only build: one instruction keeps the operands alive
    t.y0 = src[2 * P] + b[0]; t.y1 = 0.0f;
    return;
#endif
    t.y0 = fmaxf(fmaf(src[2 * P], pd.c, b[0]), pd.floor);
    t.y1 = fmaxf(fmaf(src[2 * P + 1], pd.c, b[1]), pd.floor);
}
__device__ __forceinline__ void conv_slice1(ConvTmp& t, Pending& pd) {
#ifdef NERF_ABLATE_CONV
    return;
#endif
    pd.m = fmaxf(fmaxf(pd.m, fabsf(t.y0)), fabsf(t.y1));
    t.a0 = t.y0 * pd.sc;
    t.a1 = t.y1 * pd.sc;
}
// chunk kinds (group order: pack_weights.cpp, each unit of four groups re-cut into [k-slice][hi|lo] by
// convert_stream_h2 below). CONV >= 0: while the chunk runs, step s converts register pair s of pending tile CONV;
// its two bias entries are requested one step earlier.
template <int CONV, bool FIRST, int STORE = 0>
__device__ __forceinline__ void chunk_ktile8(PipeH& p, Frag4& cur, f32x16 (&acc)[8], const XT& x, XT (&hid)[8],
                                             const f32x16 (&pend)[8], Pending& pd) {
    constexpr int C0 = CONV < 0 ? 0 : CONV;
    f32x2 r;
    ConvTmp t;
    f32x2 even;     // STORE: the even pair of a register quad waits a step for the odd one
    if constexpr (CONV >= 0) r = lds_pair_issue<128 * C0>(pd.bias_addr);
    consume_chunk<8, (CONV >= 0 ? 1 : 0)>(p, cur, [&](auto tag, auto part, const Frag4& f) {
        constexpr int s = decltype(tag)::value, pt = decltype(part)::value;
        if constexpr (pt < 6) mma_one<pt, FIRST>(acc[s], f, x);
        else if constexpr (CONV >= 0) {
            if constexpr (pt == 11) {
                conv_slice0<s>(t, pend[C0], pd, r);
                if constexpr (STORE) {
                    if constexpr ((s & 1) == 0) even = f32x2{t.y0, t.y1};
                    else keep_pairs<C0, (s >> 1), STORE>(pd, even, t.y0, t.y1);
                }
            } else if constexpr (pt == 12) conv_slice1(t, pd);
            else if constexpr (pt == 13) conv_slice2<s>(hid[C0], t);
            else if constexpr (pt == 14) {
                if constexpr (s < 7) r = lds_pair_issue<128 * C0 + 8 * (s + 1)>(pd.bias_addr);
#ifndef NERF_ABLATE_MASKS
                if constexpr (STORE) {      // the pair's ReLU flags (MlpStore::mask)
                    pd.maskw = mask_push(pd.maskw, hid[C0].hi[s >> 2][s & 3]);
                    if constexpr (s == 7 && (C0 & 1) == 1) {
                        pd.maskq[C0 >> 1] = pd.maskw;
                        pd.maskw = 0u;
                    }
                }
#endif
            }
        }
    });
}
__device__ __forceinline__ void chunk_ktile4(PipeH& p, Frag4& cur, f32x16 (&acc)[8], const XT& x) {
    consume_chunk<4, 0>(p, cur, [&](auto tag, auto part, const Frag4& f) {
        constexpr int s = decltype(tag)::value, pt = decltype(part)::value;
        if constexpr (pt < 6) mma_one<pt, false>(acc[s], f, x);
    });
}
template <int CONV, bool FIRST, int STORE = 0>
__device__ __forceinline__ void chunk_pair4(PipeH& p, Frag4& cur, f32x16 (&acc)[8], const XT& x0, const XT& x1,
                                            XT (&hid)[8], const f32x16 (&pend)[8], Pending& pd) {
    constexpr int C0 = CONV < 0 ? 0 : CONV;
    f32x2 r0, r1;
    ConvTmp t0, t1;
    f32x2 even0, even1;
    if constexpr (CONV >= 0) {
        r0 = lds_pair_issue<128 * C0>(pd.bias_addr);
        r1 = lds_pair_issue<128 * (C0 + 1)>(pd.bias_addr);
    }
    consume_chunk<8, (CONV >= 0 ? 2 : 0)>(p, cur, [&](auto tag, auto part, const Frag4& f) {
        constexpr int s = decltype(tag)::value, pt = decltype(part)::value;
        if constexpr (pt < 6) {
            if constexpr (s < 4) mma_one<pt, FIRST>(acc[s & 3], f, x0);
            else mma_one<pt, false>(acc[s & 3], f, x1);
        } else if constexpr (CONV >= 0) {
            if constexpr (pt == 11) {
                conv_slice0<s>(t0, pend[C0], pd, r0);
                conv_slice0<s>(t1, pend[C0 + 1], pd, r1);
                if constexpr (STORE) {
                    if constexpr ((s & 1) == 0) {
                        even0 = f32x2{t0.y0, t0.y1};
                        even1 = f32x2{t1.y0, t1.y1};
                    } else {
                        keep_pairs<C0, (s >> 1), STORE>(pd, even0, t0.y0, t0.y1);
                        keep_pairs<C0 + 1, (s >> 1), STORE>(pd, even1, t1.y0, t1.y1);
                    }
                }
            } else if constexpr (pt == 12) {
                conv_slice1(t0, pd);
                conv_slice1(t1, pd);
            } else if constexpr (pt == 13) {
                conv_slice2<s>(hid[C0], t0);
                conv_slice2<s>(hid[C0 + 1], t1);
            } else if constexpr (pt == 14 && s < 7) {
                r0 = lds_pair_issue<128 * C0 + 8 * (s + 1)>(pd.bias_addr);
                r1 = lds_pair_issue<128 * (C0 + 1) + 8 * (s + 1)>(pd.bias_addr);
            }
        }
    });
}
__device__ __forceinline__ void chunk_row8(PipeH& p, Frag4& cur, f32x16& acc, const XT (&x)[8]) {
    consume_chunk<8, 0>(p, cur, [&](auto tag, auto part, const Frag4& f) {
        constexpr int s = decltype(tag)::value, pt = decltype(part)::value;
        if constexpr (pt < 6) {
            if constexpr (s == 0) mma_one<pt, true>(acc, f, x[0]);
            else mma_one<pt, false>(acc, f, x[s]);
        }
    });
}
// y = relu(acc * c + bias) for the view layer's 4 tiles (the last layer: nothing to overlap with)
__device__ __forceinline__ void finish_views(f32x16 (&y)[4], const f32x16 (&acc)[8], unsigned bias_addr, float c) {
    Tile16 nxt = lds_tile_issue(bias_addr);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        Tile16 b = nxt;
        lds_tile_wait(b);
        if (t + 1 < 4) nxt = lds_tile_issue(bias_addr + 128 * (t + 1));
#pragma unroll
        for (int r = 0; r < 16; ++r) y[t][r] = fmaxf(fmaf(acc[t][r], c, b.q[r >> 2][r & 3]), 0.0f);
    }
}

// the mask word of two fp32 tiles (values >= 0) in mask_push's bit order: value 2s of the first tile at bit 15 - s, 2s + 1
// at bit 31 - s, the second tile's eight places lower
__device__ __forceinline__ unsigned relu_mask_word(const f32x16& t0, const f32x16& t1) {
    unsigned w = 0u;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        w |= (t0[2 * s] > 0.0f ? 1u : 0u) << (15 - s);
        w |= (t0[2 * s + 1] > 0.0f ? 1u : 0u) << (31 - s);
        w |= (t1[2 * s] > 0.0f ? 1u : 0u) << (7 - s);
        w |= (t1[2 * s + 1] > 0.0f ? 1u : 0u) << (23 - s);
    }
    return w;
}

// STORE: the view layer's four output tiles, 16 bytes per instruction
template <int I>
__device__ __forceinline__ void keep_tiles4(float* base, unsigned off, const f32x16 (&y)[4]) {
    if constexpr (I < 16) {
        constexpr int t = I >> 2, q = I & 3;
        keep_quad<(32 * t + 8 * q) * 4>(base, off, f32x4{y[t][4 * q], y[t][4 * q + 1], y[t][4 * q + 2], y[t][4 * q + 3]});
        keep_tiles4<I + 1>(base, off, y);
    }
}

// One output row of a Linear over 4 fp32 activation tiles (weights per register in the bias block)
__device__ __forceinline__ float row_dot4(const f32x16 (&x)[4], unsigned w_addr) {
    float s0 = 0.0f, s1 = 0.0f;
    Tile16 nxt = lds_tile_issue(w_addr);
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
        Tile16 w = nxt;
        lds_tile_wait(w);
        if (kt + 1 < 4) nxt = lds_tile_issue(w_addr + 128 * (kt + 1));
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
// STORE: the training forward pass - also writes what autograd would keep (MlpLaunch::st: every trunk layer's post-ReLU
// output, the feature vector, the view layer's output), each value as it leaves conv_slice0 / convert_pair / finish_views.
// The stream it is given is the PLAIN network's (no row equalisation): the kept activations are the reference's.
template <int MODE, int STORE = 0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void nerf_mlp_h2_kernel(const MlpLaunch a) {
    // The ring is the dynamic LDS allocation; the bias block and the small per-layer tables are static.
    extern __shared__ __attribute__((aligned(16))) char ring_lds[];
    __shared__ __attribute__((aligned(16))) float bias_lds[kBiasLdsBytes / 4];
    __shared__ __attribute__((aligned(16))) float layer_tab[4 * (kMaxDepth + 3)];   // per layer [descale, gain, max|b|, -]
    __shared__ unsigned max_record[kBwdMaxSlots];      // STORE: enter_max's records
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane >> 5;

    PipeH pipe{(const char*)a.stream_h2, ring_lds, 0, 0, a.n_chunks, wave, lane, nullptr, nullptr, nullptr, nullptr, 0, nullptr, nullptr, 0};
    pipe_start(pipe);
#ifdef NERF_STAMPS
    pipe.st = Stamper{a.stamps, 0, -1, 0, blockIdx.x == 0 && wave == 0 && a.stamps != nullptr};
#endif
    for (int k = 0; k < 2; ++k) {
        prefetch_pieces<0, 4>(piece_src(pipe, k), piece_dst(pipe, k));
        prefetch_pieces<0, 4>(piece_src(pipe, k) + 4096, piece_dst(pipe, k) + 4096);
    }
    prefetch_pieces<0, 4>(piece_src(pipe, 2), piece_dst(pipe, 2));   // chunk 0's first-half steps issue the other four
    for (int i = threadIdx.x; i < a.n_bias_tiles * kBiasTileFloats; i += 256) bias_lds[i] = a.bias[i];
    if (STORE != 0 && threadIdx.x < kBwdMaxSlots) max_record[threadIdx.x] = threadIdx.x == kBwdMaxGammaD ? 0x3f800000u : 0u;      // (|gamma(d)| <= 1)
    if (threadIdx.x < a.D + 3) {
        const int l = threadIdx.x;
        const bool has_gain = l <= (a.use_viewdirs ? a.D : a.D - 1);
        layer_tab[4 * l] = a.descale[l];
        layer_tab[4 * l + 1] = has_gain ? a.gain[2 * l] : 0.0f;
        layer_tab[4 * l + 2] = has_gain ? a.gain[2 * l + 1] : 0.0f;
        layer_tab[4 * l + 3] = 0.0f;
    }
    __syncthreads();   // chunks 0, 1, the bias block and the layer tables are in LDS
    Frag4 cur;
    {
        const unsigned fr0 = lds_byte_addr(ring_lds) + lane * 16;
        frag_issue<0>(cur.q[0], fr0);
        frag_issue<1024>(cur.q[1], fr0);
        frag_issue<2048>(cur.q[2], fr0);
        frag_issue<3072>(cur.q[3], fr0);
    }

    const unsigned bias0 = lds_addr(bias_lds) + 64 * h;   // this half-wave's entries of bias-block tile 0
    const int n_layers = a.use_viewdirs ? a.D + 1 : a.D;
    const int64_t n_tiles = (a.n_points + kPointsPerGroup - 1) / kPointsPerGroup;
#ifdef NERF_EXP_STAGGER      // timing experiment (profiles/r04_ab_notes.txt): workgroups out of phase with each other, so that the chip's
    // thousand waves do not issue their stores (and their weight-stream loads) in the same instants
    for (int k = 0; k < (int)((blockIdx.x >> 3) & 7); ++k) __builtin_amdgcn_s_sleep(NERF_EXP_STAGGER);
#endif
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
#ifdef NERF_STAMPS
        pipe.c = 0;
#endif
        pipe_tile_start(pipe);
        const int64_t tile0 = tile * kPointsPerGroup + wave * kPointsPerWave;
        const int64_t pt_raw = tile0 + (lane & 31);
        const int64_t pt = pt_raw < a.n_points ? pt_raw : a.n_points - 1;

        XT xp0, xp1;
        float m_pe;
        int t_pe;
        {
            f32x16 x0, x1, dd;
            load_inputs<MODE, true, false>(a, pt, h, x0, x1, dd);   // gamma(dir) waits for the view layer
            // The ranges of the encoded inputs are taken over the whole wavefront: wave-uniform, so they live in SGPRs
            // (the kernel has no vector register to spare: kept per lane, one of them was spilled to scratch, and its
            // reload - a VMEM load - drained the LDS-DMA weight pipeline with s_waitcnt vmcnt(0) twice per layer).
            // A wave's points share a ray or two, so the common scale costs the split nothing.
            m_pe = wave_max(tile_absmax(x1, tile_absmax(x0, 0.0f)));
            t_pe = pick_exponent(m_pe);
            if constexpr (STORE != 0) enter_max(&max_record[kBwdMaxGammaX], m_pe);      // the gamma(x) columns' weight gradients scale by it
            split_tile(xp0, x0, pow2f(t_pe));
            split_tile(xp1, x1, pow2f(t_pe));
        }

        XT hid[8];
        f32x16 accA[8], accB[8];
        Pending pd;
        float sigma = 0.0f;
        float m_prev = 0.0f;    // largest |activation| of the layer before the pending one... of its inputs

        // what the raw sums of layer l become: called when its chunks are done. m_in = largest |input| of layer l
        // (true units), t_in = exponent its inputs were scaled by
        Tile16 bias0_req;      // bias entries of the pending layer's tile 0, requested by make_pending
        auto make_pending = [&](int l, float m_in, int t_in) {
            const bool is_feature = a.use_viewdirs && l == a.D;
            const unsigned baddr = bias0 + 128 * (is_feature ? 8 * a.D + 1 : 8 * l);
            // the scale-table row and the bias entries of tile 0 in one go: five reads and ONE wait for all of them (one
            // exposed LDS latency per layer instead of two). One statement: with reads still in flight across the
            // arithmetic below, hipcc moved their destination registers whenever that arithmetic changed.
            f32x4 tab;
            asm volatile("ds_read_b128 %0, %5\n\tds_read_b128 %1, %6\n\tds_read_b128 %2, %6 offset:16\n\t"
                         "ds_read_b128 %3, %6 offset:32\n\tds_read_b128 %4, %6 offset:48\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(tab), "=&v"(bias0_req.q[0]), "=&v"(bias0_req.q[1]), "=&v"(bias0_req.q[2]), "=&v"(bias0_req.q[3])
                         : "v"(lds_byte_addr(layer_tab + 4 * l)), "v"(baddr)
                         : "memory");
            pd.c = tab[0] * pow2f(-t_in);
            pd.floor = is_feature ? -__builtin_inff() : 0.0f;
            float bound = fmaf(tab[1], m_in, tab[2]) * 1.001f;
            // the next layer may concatenate these outputs with inputs that must fit the same scale
            if (is_feature) {
                // range of gamma(dir), which the view layer concatenates: re-read from the ray record here (once per
                // tile) rather than kept in a register through the trunk
                f32x16 x0, x1, dd;
                float m_dd;
                load_inputs<MODE, false, false>(a, pt, h, x0, x1, dd, &m_dd);
                bound = fmaxf(bound, wave_max(m_dd));
            }
            else if ((a.skip_in_mask >> (l + 1)) & 1) bound = fmaxf(bound, m_pe);
            pd.t_out = pick_exponent(bound);
            pd.sc = pow2f(pd.t_out);
            pd.bias_addr = baddr;
            // The running maximum starts at 0 - or at +inf when the layer before had overflowed ON THIS POINT (m_prev = inf; not
            // m_in, which takes in the wave-wide range of the encodings): an activation beyond the fp32 range poisons
            // everything downstream in the reference (F.relu keeps +inf and NaN, nerf.py:72; the next Linear mixes
            // inf - inf), v_max_f32 would quietly drop the NaNs, and carrying the fact in the maximum costs no register:
            // the heads below turn m = inf into the reference's NaN.
            pd.m = fmaxf(m_prev - 3.4028234663852886e38f, 0.0f);
            if constexpr (STORE) {
                pd.maskw = 0u;
                pd.mask_base = wave_uniform(is_feature ? a.st.mask_hv : a.st.mask[l]);      // (feature_linear: not written)
                pd.mask_off = 16u * (2u * (unsigned)pt + (unsigned)h);
                pd.keep_base = (is_feature ? a.st.feat : a.st.h[l]) + (STORE == 2 ? 1024 : 0);      // (blocked: keep_pairs)
                pd.keep_off = 4u * ((unsigned)pt * (unsigned)(is_feature ? a.st.feat_ld : a.st.h_ld[l]) + 4u * (unsigned)h);
                if constexpr (STORE == 2)      // blocked by 32 points: 32 KiB per group of a 256-wide buffer, 32 bytes per point of a piece
                    pd.keep_off = ((unsigned)pt >> 5) * 32768u + ((unsigned)pt & 31u) * 32u + (unsigned)h * 16u;
            }
        };
        // all 8 tiles of the pending layer are converted: its true output range
        auto close_pending = [&](int slot) {
            m_prev = half_max(pd.m);
            if constexpr (STORE) {
                // this layer's ReLU-mask record (for feature_linear, which has no ReLU, the buffer is nullptr-free scratch:
                // see make_pending)
                if (slot != kBwdMaxFeatValue)
                    asm volatile("global_store_dwordx4 %0, %1, %2\n\ts_nop 1" : : "v"(pd.mask_off), "v"(pd.maskq), "s"(pd.mask_base) : "memory");
                // the largest kept activation of this layer, for the weight-gradient kernel's scale (MlpStore::maxes)
                enter_max(&max_record[slot], m_prev);
            }
            // the scale was chosen for a bound of 2^(10 - t_out); outputs 2^12 and more below it have begun to lose
            // low-half bits (see Pending). Counted, never silent: nerf_precision_status.
            const int slack = 10 - pd.t_out - __builtin_amdgcn_frexp_expf(m_prev);
            if (m_prev > 0.0f && m_prev < __builtin_inff() && slack >= 12 && pd.t_out > -60 && a.loose) atomicAdd(a.loose, 1u);
        };

        // layer 0: gamma(xyz) -> W (nerf.py:70-73)
        chunk_ktile8<-1, true>(pipe, cur, accA, xp0, hid, accB, pd);
        chunk_ktile8<-1, false>(pipe, cur, accA, xp1, hid, accB, pd);
        make_pending(0, m_pe, t_pe);

        // trunk layers 1..D-1, then (with viewdirs) feature_linear as layer D without ReLU. Layer l accumulates
        // into `out` while the pending layer l-1 is converted out of `pend`.
        auto layer_pass = [&](f32x16 (&pend)[8], f32x16 (&out)[8], int l) {
            convert_tile0_with<STORE>(hid[0], pend[0], pd, bias0_req);
            chunk_ktile8<1, true, STORE>(pipe, cur, out, hid[0], hid, pend, pd);
            next_tile_pair<STORE>(pd);
            chunk_ktile8<2, false, STORE>(pipe, cur, out, hid[1], hid, pend, pd);
            chunk_ktile8<3, false, STORE>(pipe, cur, out, hid[2], hid, pend, pd);
            next_tile_pair<STORE>(pd);
            chunk_ktile8<4, false, STORE>(pipe, cur, out, hid[3], hid, pend, pd);
            chunk_ktile8<5, false, STORE>(pipe, cur, out, hid[4], hid, pend, pd);
            next_tile_pair<STORE>(pd);
            chunk_ktile8<6, false, STORE>(pipe, cur, out, hid[5], hid, pend, pd);
            chunk_ktile8<7, false, STORE>(pipe, cur, out, hid[6], hid, pend, pd);
            chunk_ktile8<-1, false>(pipe, cur, out, hid[7], hid, pend, pd);
            close_pending(kBwdMaxKept + l - 1);
            const int t_in = pd.t_out;
            float m_in = m_prev;
            if (a.use_viewdirs && l == a.D) {
                // alpha_linear reads the post-ReLU trunk output (nerf.py:86), i.e. this layer's input: one more
                // chunk, a single-row tile accumulated into a pending tile that is no longer needed
                chunk_row8(pipe, cur, pend[0], hid);
                sigma = fmaf(pend[0][0], lds_scalar(layer_tab + 4 * (a.D + 2)) * pow2f(-t_in), lds_scalar(bias_lds + (8 * a.D) * 32));
                if (!(m_prev < __builtin_inff())) sigma = __builtin_nanf("");      // the trunk overflowed on this point (make_pending)
            }
            if (!(a.use_viewdirs && l == a.D) && ((a.skip_in_mask >> l) & 1)) {
                // h = cat[input_pts, h] (nerf.py:79-80): bring the encoded inputs to this layer's scale
                rescale_tile(xp0, t_in - t_pe);
                rescale_tile(xp1, t_in - t_pe);
                t_pe = t_in;
                chunk_ktile8<-1, false>(pipe, cur, out, xp0, hid, pend, pd);
                chunk_ktile8<-1, false>(pipe, cur, out, xp1, hid, pend, pd);
                m_in = fmaxf(m_in, m_pe);
            }
            make_pending(l, m_in, t_in);
        };
        int l = 1;
        bool pend_in_a = true;
        while (l < n_layers) {
            layer_pass(accA, accB, l);
            ++l;
            pend_in_a = false;
            if (l >= n_layers) break;
            layer_pass(accB, accA, l);
            ++l;
            pend_in_a = true;
        }
        if (!pend_in_a) {
#pragma unroll
            for (int t = 0; t < 8; ++t) accA[t] = accB[t];
        }

        const bool live = pt_raw < a.n_points;
        if (a.use_viewdirs) {
            // views_linears[0] on cat[feature, gamma(dir)] (nerf.py:93-98): 4 output tiles; the pending layer is
            // feature_linear
            convert_tile0_with<STORE>(hid[0], accA[0], pd, bias0_req);
            convert_tile<1, STORE>(hid[1], accA[1], pd);
            next_tile_pair<STORE>(pd);
            chunk_pair4<2, true, STORE>(pipe, cur, accB, hid[0], hid[1], hid, accA, pd);
            next_tile_pair<STORE>(pd);
            chunk_pair4<4, false, STORE>(pipe, cur, accB, hid[2], hid[3], hid, accA, pd);
            next_tile_pair<STORE>(pd);
            chunk_pair4<6, false, STORE>(pipe, cur, accB, hid[4], hid[5], hid, accA, pd);
            chunk_pair4<-1, false>(pipe, cur, accB, hid[6], hid[7], hid, accA, pd);
            close_pending(kBwdMaxFeatValue);
            const bool rgb_poisoned = !(m_prev < __builtin_inff());      // the trunk or feature_linear overflowed (make_pending)
            XT xd;
            {
                f32x16 x0, x1, dd;
                load_inputs<MODE, false, true>(a, pt, h, x0, x1, dd);
                split_tile(xd, dd, pd.sc);
            }
            chunk_ktile4(pipe, cur, accB, xd);
            unsigned bad;     // raw inputs re-read (a value kept across the view layer costs the step a register): requested
            {                 // here, behind the last MFMA chunk, so that the round trip runs under the colour head's arithmetic
                f32x16 x0, x1, dd;
                load_inputs<MODE, false, false>(a, pt, h, x0, x1, dd, nullptr, &bad);
            }
            f32x16 y[4];
            finish_views(y, accB, bias0 + 128 * (8 * a.D + 9), lds_scalar(layer_tab + 4 * (a.D + 1)) * pow2f(-pd.t_out));
            if constexpr (STORE) {
                const unsigned off = 4u * ((unsigned)pt * (unsigned)a.st.hv_ld + 4u * (unsigned)h);
                keep_tiles4<0>(a.st.hv, off, y);
                // the view layer's ReLU mask in the bit order of the trunk layers' (MlpStore::mask_hv, words 0 and 1)
                const unsigned moff = 16u * (2u * (unsigned)pt + (unsigned)h);
                const u32x4 rec = {relu_mask_word(y[0], y[1]), relu_mask_word(y[2], y[3]), 0u, 0u};
                asm volatile("global_store_dwordx4 %0, %1, %2\n\ts_nop 1" : : "v"(moff), "v"(rec), "s"(a.st.mask_hv) : "memory");
            }
            // rgb_linear (nerf.py:101): three rows over the 128-wide view layer
            const float* rb = bias_lds + (8 * a.D + 13) * 32;
            const float r0 = row_dot4(y, bias0 + 128 * (8 * a.D + 22)) + lds_scalar(rb);
            const float r1 = row_dot4(y, bias0 + 128 * (8 * a.D + 26)) + lds_scalar(rb + 1);
            const float r2 = row_dot4(y, bias0 + 128 * (8 * a.D + 30)) + lds_scalar(rb + 2);
            if (live && h == 0) {
                f32x4 o = {r0, r1, r2, sigma};   // outputs = cat[rgb, alpha] (nerf.py:106)
                if (bad || rgb_poisoned) {       // NaN / Inf inputs propagate as through F.relu (see kBadXyz)
                    const float qnan = __builtin_nanf("");
                    o = f32x4{qnan, qnan, qnan, (bad & kBadXyz) ? qnan : sigma};
                }
                *(f32x4*)(a.out + pt * 4) = o;
            }
        } else {
            // output_linear (nerf.py:109): rows 0..out_ch-1 of one tile; the pending layer is trunk layer D-1 (STORE: kept, with
            // its mask record and maximum, like every other trunk layer - the training pass of networks without view directions)
            convert_tile0_with<STORE>(hid[0], accA[0], pd, bias0_req);
            convert_tile<1, STORE>(hid[1], accA[1], pd);
            next_tile_pair<STORE>(pd);
            convert_tile<2, STORE>(hid[2], accA[2], pd);
            convert_tile<3, STORE>(hid[3], accA[3], pd);
            next_tile_pair<STORE>(pd);
            convert_tile<4, STORE>(hid[4], accA[4], pd);
            convert_tile<5, STORE>(hid[5], accA[5], pd);
            next_tile_pair<STORE>(pd);
            convert_tile<6, STORE>(hid[6], accA[6], pd);
            convert_tile<7, STORE>(hid[7], accA[7], pd);
            if constexpr (STORE != 0) close_pending(kBwdMaxKept + a.D - 1);
            const bool poisoned = !(half_max(pd.m) < __builtin_inff());      // the trunk overflowed on this point (make_pending)
            f32x16 o;
            chunk_row8(pipe, cur, o, hid);
            Tile16 b = lds_tile_issue(bias0 + 128 * (8 * a.D));
            lds_tile_wait(b);
            const float c = lds_scalar(layer_tab + 4 * a.D) * pow2f(-pd.t_out);
            unsigned bad;
            {
                f32x16 x0, x1, dd;
                load_inputs<MODE, false, false>(a, pt, h, x0, x1, dd, nullptr, &bad);
            }
            if (live) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
                    if (row < a.out_ch)
                        a.out[pt * a.out_ch + row] = ((bad & kBadXyz) || poisoned) ? __builtin_nanf("") : fmaf(o[r], c, b.q[r >> 2][r & 3]);
                }
            }
        }
#ifdef NERF_STAMPS
        STAMP(pipe, 0x7fffff00);
        pipe.st.on = false;   // first tile only
#endif
    }   // tile loop
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    if constexpr (STORE) {
        __syncthreads();
        flush_maxes(a.st.maxes, max_record, kBwdMaxSlots);
    }
}

hipError_t launch_mlp_h2(const MlpLaunch& a, int mode, hipStream_t s) {
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
    if (mode < 0 || mode > 2) return hipErrorInvalidValue;
    if (a.store) {
        // training forward: ray records, a view-dependent network, buffers the one-instruction stores can address
        // (16-byte aligned rows, byte offsets below 2^32)
        if (mode != kInputRays) return hipErrorInvalidValue;
        auto ok = [&](const float* b, int ld) {
            return b != nullptr && (ld & 3) == 0 && (reinterpret_cast<uintptr_t>(b) & 15) == 0 &&
                   (uint64_t)a.n_points * (uint64_t)ld * 4u < ((uint64_t)1 << 32);
        };
        // (without view directions - output_linear on the trunk, nerf.py:109 - there is no feature vector and no view layer)
        bool rows_ok = !a.use_viewdirs || (ok(a.st.feat, a.st.feat_ld) && ok(a.st.hv, a.st.hv_ld));
        for (int i = 0; i < a.D; ++i) rows_ok = rows_ok && ok(a.st.h[i], a.st.h_ld[i]);
        // the ReLU-mask records (32 bytes per point, byte offsets below 2^32) and the maxima are always written
        rows_ok = rows_ok && a.st.maxes && (a.st.mask_hv || !a.use_viewdirs) && (uint64_t)a.n_points * 32u < ((uint64_t)1 << 32);
        for (int i = 0; i < a.D; ++i) rows_ok = rows_ok && a.st.mask[i] && (reinterpret_cast<uintptr_t>(a.st.mask[i]) & 15) == 0;
        if (!rows_ok) return hipErrorInvalidValue;
        static bool raised_store[64][2] = {};
        const int blk = a.st.blocked ? 1 : 0;
        if (!raised_store[dev][blk]) {
            e = hipFuncSetAttribute(blk ? (const void*)nerf_mlp_h2_kernel<kInputRays, 2> : (const void*)nerf_mlp_h2_kernel<kInputRays, 1>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
            raised_store[dev][blk] = true;
        }
        if (blk) hipLaunchKernelGGL((nerf_mlp_h2_kernel<kInputRays, 2>), grid, block, lds, s, a);
        else hipLaunchKernelGGL((nerf_mlp_h2_kernel<kInputRays, 1>), grid, block, lds, s, a);
        return hipGetLastError();
    }
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

// gain[2l] = max over output rows of sum_k |W[row][k]|, gain[2l+1] = max |bias|, for the layers whose outputs
// are re-quantised (see Pending). Grid (layer, row group of 16): a wavefront takes a row at a time, its lanes along the
// row (four threads per row read 16 bytes of each of 16 rows per step and made 80 dependent steps: 27 us; the training
// step runs this after every optimiser step); the per-layer maxima are combined with integer atomicMax on the bit patterns
// (non-negative floats order like their bits; a maximum does not depend on the order of its operands, so the result is
// deterministic). `gain` is zeroed by the launcher.
constexpr int kGainRowsPerBlock = 16;
__global__ __launch_bounds__(256) void layer_gain_kernel(const float* params, const GainRefs refs, float* gain) {
    __shared__ float red[2][4];
    const int l = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n_in = refs.in[l];
    float g = 0.0f, bm = 0.0f;
    for (int r = wave; r < kGainRowsPerBlock; r += 4) {
        const int row = blockIdx.y * kGainRowsPerBlock + r;
        if (row >= refs.out[l]) break;
        const float* w = params + refs.w_off[l] + (size_t)row * n_in;
        float s = 0.0f;
        for (int k = lane; k < n_in; k += 64) s += fabsf(w[k]);
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        g = fmaxf(g, s);
        bm = fmaxf(bm, fabsf(params[refs.b_off[l] + row]));
    }
    if (lane == 0) {
        red[0][wave] = g;
        red[1][wave] = bm;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float gm = fmaxf(fmaxf(red[0][0], red[0][1]), fmaxf(red[0][2], red[0][3]));
        const float bb = fmaxf(fmaxf(red[1][0], red[1][1]), fmaxf(red[1][2], red[1][3]));
        atomicMax((int*)&gain[2 * l], __float_as_int(gm));
        atomicMax((int*)&gain[2 * l + 1], __float_as_int(bb));
    }
}

// Row equalisation (PackedNet::d_params_eq). ReLU commutes with positive factors and feature_linear is linear, so scaling
// hidden unit j of a layer by s_j = 2^e_j and dividing column j of every layer that reads it by s_j leaves the network's
// function unchanged - exactly, the factors being powers of two - while every row of a layer gets the same norm
// binade (the layer's median). The fp16-pair kernel scales a LAYER's weights by one factor and a POINT's activations by one factor: a unit
// with weights 2^13 below another's used to lose its low halves, and its small outputs theirs; now neither happens.
// Two kernels. row_exponents_kernel chooses the e_j: one workgroup walks the layers in order (a layer's column factors are
// its producer's row factors); a wavefront takes four rows at a time, its lanes along the columns, all of their loads in
// flight together. It only READS the parameters (0.6 MB) and writes the table of exponents; apply_row_exponents_kernel then
// writes the equalised copy element-wise, every row of every layer at once. (One kernel doing both on one workgroup took
// 0.4 ms per network; the training step equalises both networks after every optimiser step, so that a step stays a
// function of the parameters alone - a resumed run repeats the original bit for bit.)
struct EqualiseBatch {
    int n;
    const float* params[2];
    int* row_exp[2];
    float* out[2];
    EqualiseRefs refs[2];
    unsigned* flags[2];      // row_exponents_layers_kernel: the mailbox [kMaxLinears][256] per network: (epoch << 8) | (e_j + 128)
    unsigned epoch;
#ifdef NERF_ROWEXP_STAMPS      // profiles/microbench/row_exponents_bench.hip: wall_clock64() samples of thread 0 per phase
    unsigned long long* stamps;
#endif
};
#ifdef NERF_ROWEXP_STAMPS
#define ROWEXP_STAMP() if (threadIdx.x == 0 && blockIdx.x == 0) batch.stamps[n_stamp++] = wall_clock64()
#else
#define ROWEXP_STAMP()
#endif
constexpr int kRowExpThreads = 1024;
__global__ __launch_bounds__(kRowExpThreads) void row_exponents_kernel(const EqualiseBatch batch) {
    const float* params = batch.params[blockIdx.x];
    int* row_exp_out = batch.row_exp[blockIdx.x];
    const EqualiseRefs& r = batch.refs[blockIdx.x];
    __shared__ int expo[kMaxLinears][256];    // e_j of every linear (0 where rows are not scaled)
    __shared__ int row_exp[256];              // binade of a row's norm, -1000: leave the row alone
    constexpr int kExpBins = 320;             // frexp exponents of finite floats lie within -148 .. 128
    __shared__ int hist[kExpBins];
    __shared__ int median_exp, n_valid;
    // (wave: known to be uniform, so that what it indexes - rows, offsets - stays in scalar registers)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), n_waves = kRowExpThreads >> 6;
#ifdef NERF_ROWEXP_STAMPS
    int n_stamp = 0;
#endif
    for (int idx = 0; idx < r.n; ++idx) {
        ROWEXP_STAMP();
        const int k = r.order[idx];
        const int n_out = r.out[k], n_in = r.in[k], src = r.col_src[k], c0 = r.hid_col0[k], c1 = c0 + r.n_hid[k];
        if (r.scale_rows[k]) {
            // binade of every row's NORM: with inputs of comparable size a unit's output scales with
            // the l2 norm of its row and bias (the largest |w| misjudges a row that copies one input next to rows that
            // sum 256) - rows of zeros with a zero bias and non-finite rows keep factor 1
            for (int q = threadIdx.x; q < kExpBins; q += blockDim.x) hist[q] = 0;
            if (threadIdx.x == 0) n_valid = 0;
            // A wave takes kRows rows per round, a lane kCols columns of each; the rows' sums are folded together (a half
            // of the lanes keeps one half of the rows and hands over its sums of the other: kRows - 1 exchanges of a double
            // and log2(64 / kRows) more, not 6 kRows), and lane l ends up with row l / (64 / kRows) of the round.
            // profiles/microbench/row_exponents_bench.hip: 9 us per 256 x 256 layer, 0.10 ms for two networks; it was 18 and
            // 0.20 with a load behind every condition - each became a branch with its own wait - and a butterfly per row.
            constexpr int kRows = 4, kCols = 6;        // 64 x 6 = 384 columns cover in <= 383
            const int n_col_sets = (n_in + 63) >> 6;
            int nec[kCols];                            // this lane's columns: minus the producer's exponent
#pragma unroll
            for (int t = 0; t < kCols; ++t) {
                const int c = lane + 64 * t;
                nec[t] = (src >= 0 && c >= c0 && c < c1) ? -expo[src][c - c0] : 0;
            }
            const int my_row = lane / (64 / kRows);
            for (int j0 = wave * kRows; j0 < n_out; j0 += n_waves * kRows) {
                float v[kRows][kCols];
                // every load unconditional, at a clamped address, zeroed afterwards
#pragma unroll
                for (int t = 0; t < kCols; ++t)
                    if (t < n_col_sets) {              // (uniform: a 256-wide layer has four column sets)
#pragma unroll
                        for (int a = 0; a < kRows; ++a) {
                            const int j = j0 + a, c = lane + 64 * t;
                            const int jc = j < n_out ? j : n_out - 1, cc = c < n_in ? c : n_in - 1;
                            v[a][t] = params[r.w_off[k] + (size_t)jc * n_in + cc];
                        }
                    }
                const int jb = j0 + my_row < n_out ? j0 + my_row : n_out - 1;
                float bias = params[r.b_off[k] + jb];      // (with the weights, not where it is used)
                if (j0 + my_row >= n_out) bias = 0.0f;
                double m2[kRows];                              // (double: |w| up to FLT_MAX squares without overflow)
#pragma unroll
                for (int a = 0; a < kRows; ++a) m2[a] = 0.0;
#pragma unroll
                for (int t = 0; t < kCols; ++t)
                    if (t < n_col_sets) {
#pragma unroll
                        for (int a = 0; a < kRows; ++a) {
                            const float w = (j0 + a >= n_out || lane + 64 * t >= n_in) ? 0.0f : v[a][t];
                            const double x = (double)__builtin_ldexpf(w, nec[t]);
                            m2[a] = fma(x, x, m2[a]);
                        }
                    }
                int bit = 32;
#pragma unroll
                for (int w = kRows / 2; w >= 1; w >>= 1, bit >>= 1) {
                    const bool up = (lane & bit) != 0;
#pragma unroll
                    for (int a = 0; a < w; ++a) {
                        const double give = up ? m2[a] : m2[a + w], keep = up ? m2[a + w] : m2[a];
                        m2[a] = keep + __shfl_xor(give, bit);
                    }
                }
                double m1 = m2[0];
#pragma unroll
                for (; bit > 0; bit >>= 1) m1 += __shfl_xor(m1, bit);
                // the bias counts as a weight on a constant input: a row of zeros with a bias is a unit of size |b|
                m1 = fma((double)bias, (double)bias, m1);
                // binade of the norm sqrt(m1) from m1's own: m1 = f 2^E, f in [0.5, 1) -> ceil(E / 2)
                int e2 = 0;
                (void)frexp(m1, &e2);
                const bool valid = m1 > 0.0 && m1 < (double)__builtin_inff() * (double)__builtin_inff();
                if ((lane & (64 / kRows - 1)) == 0 && j0 + my_row < n_out) row_exp[j0 + my_row] = valid ? (e2 + 1) >> 1 : -1000;
            }
            ROWEXP_STAMP();
            __syncthreads();
            ROWEXP_STAMP();
            // towards the MEDIAN binade, not the largest: the ordinary units keep their scale - a skip layer concatenates
            // them with gamma(x), whose entries are not scaled, and one huge row must not push 255 others 2^20 above those
            // (a histogram over the binades a float's norm can have)
            for (int j = threadIdx.x; j < n_out; j += blockDim.x)
                if (row_exp[j] > -1000) {
                    const int q = row_exp[j] + kExpBins / 2;
                    atomicAdd(&hist[q < 0 ? 0 : (q >= kExpBins ? kExpBins - 1 : q)], 1);
                    atomicAdd(&n_valid, 1);
                }
            __syncthreads();
            ROWEXP_STAMP();
            if (wave == 0) {
                // the smallest binade with more than half of the rows at or below it: lane l owns bins 5 l .. 5 l + 4
                int own = 0;
                for (int q = 5 * lane; q < 5 * lane + 5; ++q) own += hist[q];
                int incl = own;
                for (int o = 1; o < 64; o <<= 1) {
                    const int t = __shfl_up(incl, o);
                    if (lane >= o) incl += t;
                }
                const int half = n_valid / 2;
                const unsigned long long over = __ballot(incl > half);
                if (over != 0ull && lane == __builtin_ctzll(over)) {
                    int seen = incl - own, q = 5 * lane;
                    for (; q < 5 * lane + 5; ++q) {
                        seen += hist[q];
                        if (seen > half) break;
                    }
                    median_exp = q - kExpBins / 2;
                }
                if (over == 0ull && lane == 0) median_exp = 0;      // (no valid row)
            }
            __syncthreads();
            ROWEXP_STAMP();
            for (int j = threadIdx.x; j < 256; j += blockDim.x) {
                int e = (j < n_out && row_exp[j] > -1000) ? median_exp - row_exp[j] : 0;
                e = e > 30 ? 30 : (e < -30 ? -30 : e);      // a unit 2^30 off the median is not brought all the way
                expo[k][j] = e;
                row_exp_out[k * 256 + j] = e;
            }
        } else {
            for (int j = threadIdx.x; j < 256; j += blockDim.x) {
                expo[k][j] = 0;
                row_exp_out[k * 256 + j] = 0;
            }
        }
        __syncthreads();
    }
}

// The same table with ONE WORKGROUP PER LINEAR (round 4). The chain of layers is a chain of dependencies - a layer's column
// factors are its producer's row factors - but only the arithmetic depends on them, not the loads: every workgroup fetches its
// layer's weights into registers at once (sixteen rows per wave, all of them in flight), THEN waits for its producer's flag in
// memory, reads the producer's exponents and finishes in a few microseconds. The chain's latency is a flag, sixteen rows of
// arithmetic, a histogram and a median per layer instead of a layer's worth of memory latency: 105 -> 83 us per training
// iteration for both networks (profiles/r04_train_kernel_stats.csv). Same expressions in the same order per row as row_exponents_kernel: the same table, bit for bit. All r.n x n
// workgroups are resident at once (two dozen on 256 CUs), so the waiting ones cannot starve their producers; the mailbox words
// carry the launch's epoch (24 bits) and need no reset.
__global__ __launch_bounds__(kRowExpThreads) void row_exponents_layers_kernel(const EqualiseBatch batch) {
    const int net = blockIdx.y;
    const float* params = batch.params[net];
    int* row_exp_out = batch.row_exp[net];
    unsigned* flags = batch.flags[net];
    const EqualiseRefs& r = batch.refs[net];
    const int k = blockIdx.x;
    if (k >= r.n) return;
    __shared__ int expo_src[256];             // the producer's e_j
    __shared__ int row_exp[256];              // binade of a row's norm, -1000: leave the row alone
    constexpr int kExpBins = 320;
    __shared__ int hist[kExpBins];
    __shared__ int median_exp, n_valid;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), n_waves = kRowExpThreads >> 6;
    const int n_out = r.out[k], n_in = r.in[k], src = r.col_src[k], c0 = r.hid_col0[k], c1 = c0 + r.n_hid[k];
    // Producer and consumer may sit on different XCDs, whose L2s are not coherent with each other. A release fence at agent scope
    // writes the whole L2's dirty lines back - gigabytes of the step's traffic pass through it: measured 126 us for the chain,
    // slower than one workgroup walking it - so nothing here is a fence: the exponents and the flag are device-scope atomic stores
    // (written through to where every XCD sees them), the flag goes out after every thread's stores have been acknowledged
    // (s_waitcnt + barrier), and the consumer reads both with device-scope atomic loads, flag first.
    // Each exponent travels with its own flag: the mailbox word (epoch << 8 | e + 128) of unit j is what the consumer's thread j
    // polls - one round trip through memory per layer of the chain, no separate flag, no barrier on the producer's side (the
    // table itself is written with plain stores for the kernels that follow this launch).
    auto put = [&](int j, int e) {
        row_exp_out[k * 256 + j] = e;
        __hip_atomic_store(&flags[k * 256 + j], (batch.epoch << 8) | (unsigned)(e + 128), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    if (!r.scale_rows[k]) {
        for (int j = threadIdx.x; j < 256; j += blockDim.x) put(j, 0);
        return;
    }
    constexpr int kRows = 4, kCols = 6, kRounds = 4;      // 16 waves x 4 rows x 4 rounds = 256 rows; 64 x 6 columns cover in <= 383
    const int n_col_sets = (n_in + 63) >> 6;
    const int my_row = lane / (64 / kRows);
    // ---- phase 1: this layer's weights and biases, every load in flight before anything is waited for ----
    float v[kRounds][kRows][kCols], bias[kRounds];
#pragma unroll
    for (int rr = 0; rr < kRounds; ++rr) {
        const int j0 = (wave + n_waves * rr) * kRows;
#pragma unroll
        for (int t = 0; t < kCols; ++t)
            if (t < n_col_sets) {
#pragma unroll
                for (int a = 0; a < kRows; ++a) {
                    const int j = j0 + a, c = lane + 64 * t;
                    const int jc = j < n_out ? j : n_out - 1, cc = c < n_in ? c : n_in - 1;
                    v[rr][a][t] = params[r.w_off[k] + (size_t)jc * n_in + cc];
                }
            }
        const int jb = j0 + my_row < n_out ? j0 + my_row : n_out - 1;
        bias[rr] = params[r.b_off[k] + jb];
        if (j0 + my_row >= n_out) bias[rr] = 0.0f;
    }
    for (int q = threadIdx.x; q < kExpBins; q += blockDim.x) hist[q] = 0;
    if (threadIdx.x == 0) n_valid = 0;
    // ---- phase 2: the producer's exponents ----
    if (src >= 0 && threadIdx.x < 256) {
        unsigned w;
        while (((w = __hip_atomic_load(&flags[src * 256 + threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 8) !=
               (batch.epoch & 0xffffffu))
            __builtin_amdgcn_s_sleep(1);
        expo_src[threadIdx.x] = (int)(w & 0xffu) - 128;
    }
    __syncthreads();
    int nec[kCols];                            // this lane's columns: minus the producer's exponent
#pragma unroll
    for (int t = 0; t < kCols; ++t) {
        const int c = lane + 64 * t;
        nec[t] = (src >= 0 && c >= c0 && c < c1) ? -expo_src[c - c0] : 0;
    }
    // ---- phase 3: row norms (row_exponents_kernel's arithmetic, round by round) ----
#pragma unroll
    for (int rr = 0; rr < kRounds; ++rr) {
        const int j0 = (wave + n_waves * rr) * kRows;
        if (j0 < n_out) {
            double m2[kRows];
#pragma unroll
            for (int a = 0; a < kRows; ++a) m2[a] = 0.0;
#pragma unroll
            for (int t = 0; t < kCols; ++t)
                if (t < n_col_sets) {
#pragma unroll
                    for (int a = 0; a < kRows; ++a) {
                        const float w = (j0 + a >= n_out || lane + 64 * t >= n_in) ? 0.0f : v[rr][a][t];
                        const double x = (double)__builtin_ldexpf(w, nec[t]);
                        m2[a] = fma(x, x, m2[a]);
                    }
                }
            int bit = 32;
#pragma unroll
            for (int w = kRows / 2; w >= 1; w >>= 1, bit >>= 1) {
                const bool up = (lane & bit) != 0;
#pragma unroll
                for (int a = 0; a < w; ++a) {
                    const double give = up ? m2[a] : m2[a + w], keep = up ? m2[a + w] : m2[a];
                    m2[a] = keep + __shfl_xor(give, bit);
                }
            }
            double m1 = m2[0];
#pragma unroll
            for (; bit > 0; bit >>= 1) m1 += __shfl_xor(m1, bit);
            m1 = fma((double)bias[rr], (double)bias[rr], m1);
            int e2 = 0;
            (void)frexp(m1, &e2);
            const bool valid = m1 > 0.0 && m1 < (double)__builtin_inff() * (double)__builtin_inff();
            if ((lane & (64 / kRows - 1)) == 0 && j0 + my_row < n_out) row_exp[j0 + my_row] = valid ? (e2 + 1) >> 1 : -1000;
        }
    }
    __syncthreads();
    for (int j = threadIdx.x; j < n_out; j += blockDim.x)
        if (row_exp[j] > -1000) {
            const int q = row_exp[j] + kExpBins / 2;
            atomicAdd(&hist[q < 0 ? 0 : (q >= kExpBins ? kExpBins - 1 : q)], 1);
            atomicAdd(&n_valid, 1);
        }
    __syncthreads();
    if (wave == 0) {
        int own = 0;
        for (int q = 5 * lane; q < 5 * lane + 5; ++q) own += hist[q];
        int incl = own;
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(incl, o);
            if (lane >= o) incl += t;
        }
        const int half = n_valid / 2;
        const unsigned long long over = __ballot(incl > half);
        if (over != 0ull && lane == __builtin_ctzll(over)) {
            int seen = incl - own, q = 5 * lane;
            for (; q < 5 * lane + 5; ++q) {
                seen += hist[q];
                if (seen > half) break;
            }
            median_exp = q - kExpBins / 2;
        }
        if (over == 0ull && lane == 0) median_exp = 0;      // (no valid row)
    }
    __syncthreads();
    for (int j = threadIdx.x; j < 256; j += blockDim.x) {
        int e = (j < n_out && row_exp[j] > -1000) ? median_exp - row_exp[j] : 0;
        e = e > 30 ? 30 : (e < -30 ? -30 : e);
        put(j, e);
    }
}

// out[k][j][c] = params[k][j][c] * 2^(e_kj - e_src(k),c) (row_exponents_kernel's table; exact): the copy of the network the
// fp16-pair kernels evaluate. Grid (row block of 4, linear); a wavefront per row.
__global__ __launch_bounds__(256) void apply_row_exponents_kernel(const EqualiseBatch batch) {
    const float* params = batch.params[blockIdx.z];
    const int* row_exp = batch.row_exp[blockIdx.z];
    float* out = batch.out[blockIdx.z];
    const EqualiseRefs& r = batch.refs[blockIdx.z];
    const int k = blockIdx.y, lane = threadIdx.x & 63, j = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (k >= r.n || j >= r.out[k]) return;
    const int n_in = r.in[k], src = r.col_src[k], c0 = r.hid_col0[k], c1 = c0 + r.n_hid[k];
    const int e = row_exp[k * 256 + j];
    const float* w = params + r.w_off[k] + (size_t)j * n_in;
    float* o = out + r.w_off[k] + (size_t)j * n_in;
    for (int c = lane; c < n_in; c += 64) {
        const int ec = (src >= 0 && c >= c0 && c < c1) ? row_exp[src * 256 + c - c0] : 0;
        o[c] = __builtin_ldexpf(w[c], e - ec);      // (|e|, |ec| <= 30: one exact scaling)
    }
    if (lane == 0) out[r.b_off[k] + j] = __builtin_ldexpf(params[r.b_off[k] + j], e);
}

hipError_t launch_equalise_rows(int n, const float* const* params, const EqualiseRefs* refs, float* const* params_eq,
                                int* const* row_exp, hipStream_t s, unsigned* const* flags, unsigned epoch) {
    if (n < 1 || n > 2) return hipErrorInvalidValue;
    EqualiseBatch b{};
    b.n = n;
    int max_out = 1, max_n = 1;
    for (int i = 0; i < n; ++i) {
        if (refs[i].n <= 0 || refs[i].n > kMaxLinears || !row_exp[i] || !params[i] || !params_eq[i]) return hipErrorInvalidValue;
        for (int k = 0; k < refs[i].n; ++k) {
            if (refs[i].out[k] > 256 || refs[i].in[k] > 383) return hipErrorInvalidValue;
            max_out = refs[i].out[k] > max_out ? refs[i].out[k] : max_out;
        }
        max_n = refs[i].n > max_n ? refs[i].n : max_n;
        b.params[i] = params[i];
        b.row_exp[i] = row_exp[i];
        b.out[i] = params_eq[i];
        b.refs[i] = refs[i];
    }
    // one workgroup per linear with flags in memory (flags given, rows of <= 256 units in <= 4 rounds of 64), or the
    // one-workgroup-per-network walk (loading; NERF_TRAIN_GLUE=legacy)
    if (flags && flags[0] && (n < 2 || flags[1])) {
        for (int i = 0; i < n; ++i) b.flags[i] = flags[i];
        b.epoch = epoch;
        hipLaunchKernelGGL(row_exponents_layers_kernel, dim3(max_n, n), dim3(kRowExpThreads), 0, s, b);
    } else {
        hipLaunchKernelGGL(row_exponents_kernel, dim3(n), dim3(kRowExpThreads), 0, s, b);
    }
    hipLaunchKernelGGL(apply_row_exponents_kernel, dim3((max_out + 3) / 4, max_n, n), dim3(256), 0, s, b);
    return hipGetLastError();
}

hipError_t launch_layer_gains(const float* params, const GainRefs& refs, float* gain, hipStream_t s) {
    if (refs.n <= 0) return hipSuccess;
    hipError_t e = hipMemsetAsync(gain, 0, 2 * (size_t)refs.n * sizeof(float), s);
    if (e != hipSuccess) return e;
    int max_out = 1;
    for (int l = 0; l < refs.n; ++l) max_out = refs.out[l] > max_out ? refs.out[l] : max_out;
    hipLaunchKernelGGL(layer_gain_kernel, dim3(refs.n, (max_out + kGainRowsPerBlock - 1) / kGainRowsPerBlock), dim3(256), 0, s,
                       params, refs, gain);
    return hipGetLastError();
}

hipError_t launch_convert_stream_h2(const float* stream, const int* chunk_layer, int n_chunks, float* chunk_max,
                                    uint32_t* out, float* descale, hipStream_t s) {
    if (n_chunks <= 0) return hipSuccess;
    hipLaunchKernelGGL(chunk_absmax_kernel, dim3(n_chunks), dim3(256), 0, s, stream, chunk_max);
    hipLaunchKernelGGL(convert_stream_h2_kernel, dim3(n_chunks), dim3(256), 0, s, stream, chunk_layer, chunk_max,
                       n_chunks, out, descale);
    // the kernel's weight ring runs three chunks ahead, across tile boundaries: the stream is followed by a copy of its
    // head, so that a position only ever advances within a tile (kStreamTailChunks, PipeH)
    if (n_chunks < kStreamTailChunks) return hipErrorInvalidValue;
    return hipMemcpyAsync(out + (size_t)n_chunks * kChunkFloats, out, (size_t)kStreamTailChunks * kChunkBytes,
                          hipMemcpyDeviceToDevice, s);
}

}  // namespace nerf
