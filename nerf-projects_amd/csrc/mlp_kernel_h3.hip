// Fused positional-encoding + NeRF MLP kernel, "fp16-pair" arithmetic on v_mfma_f32_16x16x32_f16.
//
// Same job, same arithmetic and same pipeline as mlp_kernel_h2.hip (Embedder.embed nerf/embedder.py:72-80, the viewdir
// broadcast + concat of run_network nerf.ipynb:827-843, NeRF.forward nerf/nerf.py:57-111; every fp32 operand carried
// exactly as two fp16 halves, three MFMA products per term, per-layer / per-point power-of-two scaling, conversion of a
// layer's sums in the shadow of the next layer's MFMAs, weights streamed L2 -> LDS by LDS-DMA through a four-slot ring),
// but the products run on the 16x16x32 shape of the half-precision matrix pipe. This kernel is power-limited - the
// chip lowers its clock under the MFMA load - and the clock it holds depends on the shape: with the same work, the
// same LDS bytes and random operands a 16x16x32 loop ran 1.18x (bare) / 1.10x (with the step's LDS reads, LDS-DMA and
// vector work) as fast as the 32x32x16 loop (profiles/microbench/step_mix.hip, profiles/r01_mfma_microbench.txt).
//
// Layout. A wavefront still owns 32 points, as two column groups P0 (points 0..15) and P1 (16..31): lane = (column
// c = lane & 15, k-group g = lane >> 4) serves points c and 16 + c. A 32-row output tile is two 16-row tiles T0, T1, and
// one step = one 32-deep k-tile against one 32-row output tile = 12 MFMAs fed by four A-fragments [T0 hi, T0 lo,
// T1 hi, T1 lo] of 1 KiB (the same bytes per MAC as before: a fragment serves both point groups). The accumulator of
// a step is four 16x16 tiles ordered [P0 T0, P0 T1, P1 T0, P1 T1]; lane (c, g) holds rows 4 g + i of each, so the 8
// registers of a point group are, in order, exactly the 8 k-positions (k = 8 g + j) this lane feeds to the next layer's
// B operand: layers chain in registers as before (feature permutation hidden_col3, applied by pack_weights layout 1).
// gamma(xyz) / gamma(dir) need no cross-lane traffic at all here: a lane's positions are (sin, cos) pairs of one
// (frequency, component) each (pe3_col_xyz / pe3_col_dir), one sincosf per pair.
#include "mlp_pair_common.h"

namespace nerf {

namespace h3 {

// accumulator of one step: [2 P + T] 16x16 tiles; as 16 registers r = 4 (2 P + T) + i
struct Acc {
    f32x4 v[4];
};
__device__ __forceinline__ float acc_reg(const Acc& a, int r) { return a.v[r >> 2][r & 3]; }

__device__ __forceinline__ f32x4 mma16(const f32x4& a, const u32x4& b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h16x8, a), __builtin_bit_cast(h16x8, b), c, 0, 0, 0);
}

// MFMA pair K (0..5) of a step: K / 3 = the 16-row tile, K % 3 = the product (w_lo x_hi, w_hi x_lo, w_hi x_hi: small
// terms first), each against both point groups. FIRST: the accumulators start from zero. T0_ONLY: the weights of T1
// are all padding (one-row layers), its MFMAs are skipped.
template <int K, bool FIRST, bool T0_ONLY = false>
__device__ __forceinline__ void mma_pair(Acc& acc, const Frag4& f, const XT& x) {
    constexpr int T = K / 3, prod = K % 3;
    if constexpr (!(T0_ONLY && T == 1)) {
        const f32x4& w = f.q[2 * T + (prod == 0 ? 1 : 0)];
#pragma unroll
        for (int P = 0; P < 2; ++P) {
            const u32x4& b = prod == 1 ? x.lo[P] : x.hi[P];
            if constexpr (FIRST && prod == 0) {
                const f32x4 zero = {0, 0, 0, 0};
                acc.v[2 * P + T] = mma16(w, b, zero);
            } else {
                acc.v[2 * P + T] = mma16(w, b, acc.v[2 * P + T]);
            }
        }
    }
}

// max / sum over the four lanes (c, g = 0..3) that share a point pair
__device__ __forceinline__ float quad_max(float m) {
    m = fmaxf(m, __shfl_xor(m, 16));
    return fmaxf(m, __shfl_xor(m, 32));
}
__device__ __forceinline__ float quad_sum(float s) {
    s += __shfl_xor(s, 16);
    return s + __shfl_xor(s, 32);
}
// multiply point group P of a split tile by 2^d[P] (exact while nothing leaves the fp16 range)
__device__ __forceinline__ void rescale_tile(XT& x, const int (&d)[2]) {
#pragma unroll
    for (int P = 0; P < 2; ++P) {
        const _Float16 f = (_Float16)pow2f(d[P] < -30 ? -30 : (d[P] > 15 ? 15 : d[P]));
        const h16x2 ff = {f, f};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const unsigned wh = x.hi[P][q], wl = x.lo[P][q];
            const h16x2 ph = __builtin_bit_cast(h16x2, wh) * ff, pl = __builtin_bit_cast(h16x2, wl) * ff;
            x.hi[P][q] = __builtin_bit_cast(unsigned, ph);
            x.lo[P][q] = __builtin_bit_cast(unsigned, pl);
        }
    }
}

// ---- LDS reads outside hipcc's LDS-DMA guard -------------------------------------------------------------------
// a lane's 8 entries of one bias-block tile: [T][i]
struct Oct {
    f32x4 q[2];
};
__device__ __forceinline__ Oct lds_oct_issue(unsigned addr) {
    Oct t;
    asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:16" : "=&v"(t.q[0]), "=&v"(t.q[1]) : "v"(addr) : "memory");
    return t;
}
__device__ __forceinline__ void lds_oct_wait(Oct& t) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(t.q[0]), "+v"(t.q[1])::"memory");
}
// ---- the layer whose raw sums wait to become the next layer's operands (as mlp_kernel_h2.hip; a lane now carries two
// points, so the per-point numbers come in pairs) ----------------------------------------------------------------
struct Pending {
    float c[2];      // raw sum -> activation: descale * 2^-t_in (per point)
    float floor;     // 0 for ReLU, -inf for feature_linear
    float sc[2];     // activation -> operand: 2^t_out (per point)
    int t_out[2];
    unsigned bias_addr;   // LDS address of this k-group's bias entries of tile 0 (tile t: + 128 t)
    float m[2];      // running max |y|
};

// bias entries of accumulator registers 2 Pp, 2 Pp + 1: [T][i] with T = (Pp >> 1) & 1, i = 2 (Pp & 1)
__host__ __device__ constexpr int bias_pair_off(int Pp) { return 8 * (2 * ((Pp >> 1) & 1) + (Pp & 1)); }

template <int Pp>
__device__ __forceinline__ void conv_slice0(ConvTmp& t, const Acc& src, const Pending& pd, const f32x2& b) {
#ifdef NERF_ABLATE_CONV
    t.y0 = acc_reg(src, 2 * Pp) + b[0]; t.y1 = 0.0f;
    return;
#endif
    t.y0 = fmaxf(fmaf(acc_reg(src, 2 * Pp), pd.c[Pp >> 2], b[0]), pd.floor);
    t.y1 = fmaxf(fmaf(acc_reg(src, 2 * Pp + 1), pd.c[Pp >> 2], b[1]), pd.floor);
}
template <int Pp>
__device__ __forceinline__ void conv_slice1(ConvTmp& t, Pending& pd) {
#ifdef NERF_ABLATE_CONV
    return;
#endif
    pd.m[Pp >> 2] = fmaxf(fmaxf(pd.m[Pp >> 2], fabsf(t.y0)), fabsf(t.y1));
    t.a0 = t.y0 * pd.sc[Pp >> 2];
    t.a1 = t.y1 * pd.sc[Pp >> 2];
}
// a whole tile at once (not hidden: tile 0 at the start of a layer)
template <int Pp>
__device__ __forceinline__ void convert_pairs(XT& dst, const Acc& src, Pending& pd, const Oct& b) {
    if constexpr (Pp < 8) {
        constexpr int e = bias_pair_off(Pp) / 4;   // float index within the lane's 8
        ConvTmp t;
        conv_slice0<Pp>(t, src, pd, f32x2{b.q[e >> 2][e & 3], b.q[e >> 2][(e & 3) + 1]});
        conv_slice1<Pp>(t, pd);
        conv_slice2<Pp>(dst, t);
        convert_pairs<Pp + 1>(dst, src, pd, b);
    }
}
template <int T>
__device__ __forceinline__ void convert_tile(XT& dst, const Acc& src, Pending& pd) {
    Oct b = lds_oct_issue(pd.bias_addr + 128 * T);
    lds_oct_wait(b);
    convert_pairs<0>(dst, src, pd, b);
}

// ---- chunk kinds (group order: pack_weights.cpp layout 1; each unit of four groups re-cut into [T][hi|lo] by
// convert_stream_h2). CONV >= 0: while the chunk runs, step s converts register pair s of pending tile CONV; its two
// bias entries are requested one step earlier.
template <int CONV, bool FIRST>
__device__ __forceinline__ void chunk_ktile8(PipeH& p, Frag4& cur, Acc (&acc)[8], const XT& x, XT (&hid)[8],
                                             const Acc (&pend)[8], Pending& pd) {
    constexpr int C0 = CONV < 0 ? 0 : CONV;
    f32x2 r;
    ConvTmp t;
    if constexpr (CONV >= 0) r = lds_pair_issue<128 * C0 + bias_pair_off(0)>(pd.bias_addr);
    consume_chunk<8, (CONV >= 0 ? 1 : 0)>(p, cur, [&](auto tag, auto part, const Frag4& f) {
        constexpr int s = decltype(tag)::value, pt = decltype(part)::value;
        if constexpr (pt < 6) mma_pair<pt, FIRST>(acc[s], f, x);
        else if constexpr (CONV >= 0) {
            if constexpr (pt == 11) conv_slice0<s>(t, pend[C0], pd, r);
            else if constexpr (pt == 12) conv_slice1<s>(t, pd);
            else if constexpr (pt == 13) conv_slice2<s>(hid[C0], t);
            else if constexpr (pt == 14 && s < 7) r = lds_pair_issue<128 * C0 + bias_pair_off(s + 1)>(pd.bias_addr);
        }
    });
}
// one k-tile against 4 output tiles (direction part of the view layer)
__device__ __forceinline__ void chunk_ktile4(PipeH& p, Frag4& cur, Acc (&acc)[8], const XT& x) {
    consume_chunk<4, 0>(p, cur, [&](auto tag, auto part, const Frag4& f) {
        constexpr int s = decltype(tag)::value, pt = decltype(part)::value;
        if constexpr (pt < 6) mma_pair<pt, false>(acc[s], f, x);
    });
}
// two k-tiles against 4 output tiles (feature part of the view layer): steps 0-3 use x0, 4-7 use x1; converts
// pending tiles CONV and CONV + 1 meanwhile (two register pairs per step)
template <int CONV, bool FIRST>
__device__ __forceinline__ void chunk_pair4(PipeH& p, Frag4& cur, Acc (&acc)[8], const XT& x0, const XT& x1,
                                            XT (&hid)[8], const Acc (&pend)[8], Pending& pd) {
    constexpr int C0 = CONV < 0 ? 0 : CONV;
    f32x2 r0, r1;
    ConvTmp t0, t1;
    if constexpr (CONV >= 0) {
        r0 = lds_pair_issue<128 * C0 + bias_pair_off(0)>(pd.bias_addr);
        r1 = lds_pair_issue<128 * (C0 + 1) + bias_pair_off(0)>(pd.bias_addr);
    }
    consume_chunk<8, (CONV >= 0 ? 2 : 0)>(p, cur, [&](auto tag, auto part, const Frag4& f) {
        constexpr int s = decltype(tag)::value, pt = decltype(part)::value;
        if constexpr (pt < 6) {
            if constexpr (s < 4) mma_pair<pt, FIRST>(acc[s & 3], f, x0);
            else mma_pair<pt, false>(acc[s & 3], f, x1);
        } else if constexpr (CONV >= 0) {
            if constexpr (pt == 11) {
                conv_slice0<s>(t0, pend[C0], pd, r0);
                conv_slice0<s>(t1, pend[C0 + 1], pd, r1);
            } else if constexpr (pt == 12) {
                conv_slice1<s>(t0, pd);
                conv_slice1<s>(t1, pd);
            } else if constexpr (pt == 13) {
                conv_slice2<s>(hid[C0], t0);
                conv_slice2<s>(hid[C0 + 1], t1);
            } else if constexpr (pt == 14 && s < 7) {
                r0 = lds_pair_issue<128 * C0 + bias_pair_off(s + 1)>(pd.bias_addr);
                r1 = lds_pair_issue<128 * (C0 + 1) + bias_pair_off(s + 1)>(pd.bias_addr);
            }
        }
    });
}
// 8 k-tiles against ONE output tile: step s = k-tile s. T0_ONLY: at most 16 output rows exist
template <bool T0_ONLY>
__device__ __forceinline__ void chunk_row8(PipeH& p, Frag4& cur, Acc& acc, const XT (&x)[8]) {
    if constexpr (T0_ONLY) {
        const f32x4 zero = {0, 0, 0, 0};
        acc.v[1] = zero;
        acc.v[3] = zero;
    }
    consume_chunk<8, 0>(p, cur, [&](auto tag, auto part, const Frag4& f) {
        constexpr int s = decltype(tag)::value, pt = decltype(part)::value;
        if constexpr (pt < 6) {
            if constexpr (s == 0) mma_pair<pt, true, T0_ONLY>(acc, f, x[0]);
            else mma_pair<pt, false, T0_ONLY>(acc, f, x[s]);
        }
    });
}

// y = relu(acc * c + bias) for the view layer's 4 tiles (the last layer: nothing to overlap with)
__device__ __forceinline__ void finish_views(f32x16 (&y)[4], const Acc (&acc)[8], unsigned bias_addr, const float (&c)[2]) {
    Oct nxt = lds_oct_issue(bias_addr);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        Oct b = nxt;
        lds_oct_wait(b);
        if (t + 1 < 4) nxt = lds_oct_issue(bias_addr + 128 * (t + 1));
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int e = 4 * ((r >> 2) & 1) + (r & 3);   // [T][i]
            y[t][r] = fmaxf(fmaf(acc_reg(acc[t], r), c[r >> 3], b.q[e >> 2][e & 3]), 0.0f);
        }
    }
}

// One output row of a Linear over 4 fp32 activation tiles, for both point groups (weights per register in the bias
// block, [g][T][i] per tile)
__device__ __forceinline__ void row_dot4(const f32x16 (&x)[4], unsigned w_addr, float (&out)[2]) {
    float s0 = 0.0f, s1 = 0.0f;
    Oct nxt = lds_oct_issue(w_addr);
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
        Oct w = nxt;
        lds_oct_wait(w);
        if (kt + 1 < 4) nxt = lds_oct_issue(w_addr + 128 * (kt + 1));
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            s0 = fmaf(w.q[e >> 2][e & 3], x[kt][e], s0);
            s1 = fmaf(w.q[e >> 2][e & 3], x[kt][8 + e], s1);
        }
    }
    out[0] = quad_sum(s0);
    out[1] = quad_sum(s1);
}

// ---- inputs: gamma(xyz), gamma(dir) of one point for k-group g ----------------------------------------------------
// x0 / x1: the lane's 8 positions of gamma(xyz) k-tiles 0 / 1, dd: of the gamma(dir) tile (pe3_col_xyz, pe3_col_dir).
// x * 2^k is exact in fp32 (embedder.py:48,61); sincosf is the accurate ocml routine (arguments reach |x| * 512).
__device__ __forceinline__ float pick3(const float (&v)[3], int c) { return c == 0 ? v[0] : (c == 1 ? v[1] : v[2]); }

template <bool WANT_XYZ, bool WANT_DIR>
__device__ __forceinline__ void encode_point(const float (&p)[3], const float (&d)[3], int g, bool dirs, float (&x0)[8],
                                             float (&x1)[8], float (&dd)[8]) {
    if constexpr (WANT_XYZ) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int jp = 0; jp < 4; ++jp) {
                const int q = (4 * t + g) * 4 + jp;        // 0..31; 30, 31 only for t = 1, g = 3, jp = 2, 3
                const int qq = q < 30 ? q : 29;
                const int freq = qq / 3, comp = qq - 3 * freq;
                float sn, cs;
#ifdef NERF_ABLATE_PE
                sn = __builtin_ldexpf(pick3(p, comp), freq); cs = sn + 1.0f;
#else
                sincosf(__builtin_ldexpf(pick3(p, comp), freq), &sn, &cs);
#endif
                if (t == 1 && jp >= 2) {
                    if (q == 30) { sn = p[0]; cs = p[1]; }
                    if (q == 31) { sn = p[2]; cs = 0.0f; }
                }
                if (t == 0) { x0[2 * jp] = sn; x0[2 * jp + 1] = cs; }
                else { x1[2 * jp] = sn; x1[2 * jp + 1] = cs; }
            }
    }
    if constexpr (WANT_DIR) {
#pragma unroll
        for (int jp = 0; jp < 4; ++jp) {
            const int q = 4 * g + jp;                      // 0..15; 12.. only for g = 3
            const int qq = q < 12 ? q : 11;
            const int freq = qq / 3, comp = qq - 3 * freq;
            float sn, cs;
#ifdef NERF_ABLATE_PE
            sn = __builtin_ldexpf(pick3(d, comp), freq); cs = sn + 1.0f;
#else
            sincosf(__builtin_ldexpf(pick3(d, comp), freq), &sn, &cs);
#endif
            if (q == 12) { sn = d[0]; cs = d[1]; }
            if (q == 13) { sn = d[2]; cs = 0.0f; }
            if (q > 13) { sn = 0.0f; cs = 0.0f; }
            dd[2 * jp] = dirs ? sn : 0.0f;
            dd[2 * jp + 1] = dirs ? cs : 0.0f;
        }
    }
}

// dir_max: largest |component| of the direction (an upper bound of |gamma(dir)| together with 1), or of the encoded
// direction columns in embedded mode
template <int MODE, bool WANT_XYZ, bool WANT_DIR>
__device__ __forceinline__ void load_point(const MlpLaunch& a, int64_t pt, int g, float (&x0)[8], float (&x1)[8],
                                           float (&dd)[8], float* dir_max = nullptr) {
    if (MODE == kInputEmbedded) {
        const float* row = a.x + pt * a.x_ld;
        if constexpr (WANT_XYZ) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int c0 = pe3_col_xyz(0, g, j), c1 = pe3_col_xyz(1, g, j);
                x0[j] = (c0 >= 0 && c0 < a.in_ch) ? row[c0] : 0.0f;
                x1[j] = (c1 >= 0 && c1 < a.in_ch) ? row[c1] : 0.0f;
            }
        }
        float m = 0.0f;
        if (WANT_DIR || dir_max) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int c = pe3_col_dir(g, j);
                const float v = (a.use_viewdirs && c >= 0 && c < a.in_ch_views) ? row[a.in_ch + c] : 0.0f;
                if constexpr (WANT_DIR) dd[j] = v;
                m = fmaxf(m, fabsf(v));
            }
        }
        if (dir_max) *dir_max = quad_max(m);
        return;
    }
    float p[3], d[3] = {0.0f, 0.0f, 0.0f};
    const int64_t ray = pt / a.samples_per_ray;
    if (MODE == kInputPoints) {
        p[0] = a.pts[pt * 3 + 0];
        p[1] = a.pts[pt * 3 + 1];
        p[2] = a.pts[pt * 3 + 2];
        if (a.viewdirs) {
            d[0] = a.viewdirs[ray * 3 + 0];
            d[1] = a.viewdirs[ray * 3 + 1];
            d[2] = a.viewdirs[ray * 3 + 2];
        }
    } else {
        // pts = rays_o + rays_d * z (nerf.ipynb:447, :468): product and sum rounded separately
        const float* r = a.rays + ray * a.ray_ld;
        const float z = a.z_vals[pt];
#pragma unroll
        for (int c = 0; c < 3; ++c) p[c] = __fadd_rn(r[c], __fmul_rn(r[3 + c], z));
        if (a.ray_ld > 8) {
            d[0] = r[a.ray_ld - 3];
            d[1] = r[a.ray_ld - 2];
            d[2] = r[a.ray_ld - 1];
        }
    }
    if (dir_max) *dir_max = fmaxf(fmaxf(fabsf(d[0]), fabsf(d[1])), fmaxf(fabsf(d[2]), 1.0f));
    encode_point<WANT_XYZ, WANT_DIR>(p, d, g, a.use_viewdirs != 0, x0, x1, dd);
}

// both points of a lane: tiles in [point group][position] order
template <int MODE, bool WANT_XYZ, bool WANT_DIR>
__device__ __forceinline__ void load_inputs(const MlpLaunch& a, const int64_t (&pt)[2], int g, f32x16& x0, f32x16& x1,
                                            f32x16& dd, float* dir_max = nullptr) {
    float m = 0.0f;
#pragma unroll
    for (int P = 0; P < 2; ++P) {
        float a0[8], a1[8], ad[8], mp = 0.0f;
        load_point<MODE, WANT_XYZ, WANT_DIR>(a, pt[P], g, a0, a1, ad, dir_max ? &mp : nullptr);
        m = fmaxf(m, mp);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if constexpr (WANT_XYZ) {
                x0[8 * P + j] = a0[j];
                x1[8 * P + j] = a1[j];
            }
            if constexpr (WANT_DIR) dd[8 * P + j] = ad[j];
        }
    }
    if (dir_max) *dir_max = m;
}

}  // namespace h3

using namespace h3;

// ---- the kernel -------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void nerf_mlp_h3_kernel(const MlpLaunch a) {
    // The ring is the dynamic LDS allocation; the bias block and the small per-layer tables are static.
    extern __shared__ __attribute__((aligned(16))) char ring_lds[];
    __shared__ __attribute__((aligned(16))) float bias_lds[kBiasLdsBytes / 4];
    __shared__ __attribute__((aligned(16))) float layer_tab[4 * (kMaxDepth + 3)];   // per layer [descale, gain, max|b|, -]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4;

    PipeH pipe{(const char*)a.stream_h3, ring_lds, 0, 0, a.n_chunks, wave, lane, nullptr, nullptr, nullptr, nullptr};
    for (int k = 0; k < 2; ++k) {
        prefetch_pieces<0, 4>(piece_src(pipe, k), piece_dst(pipe, k));
        prefetch_pieces<0, 4>(piece_src(pipe, k) + 4096, piece_dst(pipe, k) + 4096);
    }
    prefetch_pieces<0, 4>(piece_src(pipe, 2), piece_dst(pipe, 2));   // chunk 0's first-half steps issue the other four
    for (int i = threadIdx.x; i < a.n_bias_tiles * kBiasTileFloats; i += 256) bias_lds[i] = a.bias3[i];
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

    const unsigned bias0 = lds_byte_addr(bias_lds) + 32 * g;   // this k-group's entries of bias-block tile 0
    const int n_layers = a.use_viewdirs ? a.D + 1 : a.D;
    const int64_t n_tiles = (a.n_points + kPointsPerGroup - 1) / kPointsPerGroup;
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        pipe.c = 0;
        const int64_t tile0 = tile * kPointsPerGroup + wave * kPointsPerWave;
        const int64_t pt_raw[2] = {tile0 + (lane & 15), tile0 + 16 + (lane & 15)};
        const int64_t pt[2] = {pt_raw[0] < a.n_points ? pt_raw[0] : a.n_points - 1,
                               pt_raw[1] < a.n_points ? pt_raw[1] : a.n_points - 1};

        XT xp0, xp1;
        float m_pe;
        int t_pe[2];   // exponent the encoded tiles are currently scaled by, per point group
        {
            f32x16 x0, x1, dd;
            load_inputs<MODE, true, false>(a, pt, g, x0, x1, dd);   // gamma(dir) waits for the view layer
            // range of the encoded inputs over the whole wavefront: wave-uniform, so it lives in an SGPR
            m_pe = wave_max(tile_absmax(x1, tile_absmax(x0, 0.0f)));
            t_pe[0] = t_pe[1] = pick_exponent(m_pe);
            split_tile(xp0, x0, pow2f(t_pe[0]));
            split_tile(xp1, x1, pow2f(t_pe[0]));
        }

        XT hid[8];
        Acc accA[8], accB[8];
        Pending pd;
        float sigma[2] = {0.0f, 0.0f};
        float m_prev[2];

        // what the raw sums of layer l become: called when its chunks are done. m_in = largest |input| of layer l
        // (true units), t_in = exponent its inputs were scaled by; both per point
        auto make_pending = [&](int l, const float (&m_in)[2], const int (&t_in)[2]) __attribute__((always_inline)) {
            const bool is_feature = a.use_viewdirs && l == a.D;
            const f32x4 tab = lds_vec4(layer_tab + 4 * l);
            pd.floor = is_feature ? -__builtin_inff() : 0.0f;
            float extra = 0.0f;
            // the next layer may concatenate these outputs with inputs that must fit the same scale
            if (is_feature) {
                f32x16 x0, x1, dd;
                float m_dd;
                load_inputs<MODE, false, false>(a, pt, g, x0, x1, dd, &m_dd);
                extra = wave_max(m_dd);
            } else if ((a.skip_in_mask >> (l + 1)) & 1) {
                extra = m_pe;
            }
#pragma unroll
            for (int P = 0; P < 2; ++P) {
                pd.c[P] = tab[0] * pow2f(-t_in[P]);
                const float bound = fmaxf(fmaf(tab[1], m_in[P], tab[2]) * 1.001f, extra);
                pd.t_out[P] = pick_exponent(bound);
                pd.sc[P] = pow2f(pd.t_out[P]);
                pd.m[P] = 0.0f;
            }
            pd.bias_addr = bias0 + 128 * (is_feature ? 8 * a.D + 1 : 8 * l);
        };
        // all 8 tiles of the pending layer are converted: its true output range
        auto close_pending = [&]() __attribute__((always_inline)) {
            bool loose = false;
#pragma unroll
            for (int P = 0; P < 2; ++P) {
                m_prev[P] = quad_max(pd.m[P]);
                // the scale was chosen for a bound of 2^(10 - t_out); outputs 2^12 and more below it have begun to lose
                // low-half bits. Counted, never silent: nerf_precision_status.
                const int slack = 10 - pd.t_out[P] - __builtin_amdgcn_frexp_expf(m_prev[P]);
                loose = loose || (m_prev[P] > 0.0f && slack >= 12 && pd.t_out[P] > -60);
            }
            if (loose && a.loose) atomicAdd(a.loose, 1u);
        };

        // layer 0: gamma(xyz) -> W (nerf.py:70-73)
        chunk_ktile8<-1, true>(pipe, cur, accA, xp0, hid, accB, pd);
        chunk_ktile8<-1, false>(pipe, cur, accA, xp1, hid, accB, pd);
        {
            const float m0[2] = {m_pe, m_pe};
            make_pending(0, m0, t_pe);
        }

        // trunk layers 1..D-1, then (with viewdirs) feature_linear as layer D without ReLU. Layer l accumulates
        // into `out` while the pending layer l-1 is converted out of `pend`.
        auto layer_pass = [&](Acc (&pend)[8], Acc (&out)[8], int l) __attribute__((always_inline)) {
            convert_tile<0>(hid[0], pend[0], pd);
            chunk_ktile8<1, true>(pipe, cur, out, hid[0], hid, pend, pd);
            chunk_ktile8<2, false>(pipe, cur, out, hid[1], hid, pend, pd);
            chunk_ktile8<3, false>(pipe, cur, out, hid[2], hid, pend, pd);
            chunk_ktile8<4, false>(pipe, cur, out, hid[3], hid, pend, pd);
            chunk_ktile8<5, false>(pipe, cur, out, hid[4], hid, pend, pd);
            chunk_ktile8<6, false>(pipe, cur, out, hid[5], hid, pend, pd);
            chunk_ktile8<7, false>(pipe, cur, out, hid[6], hid, pend, pd);
            chunk_ktile8<-1, false>(pipe, cur, out, hid[7], hid, pend, pd);
            close_pending();
            const int t_in[2] = {pd.t_out[0], pd.t_out[1]};
            float m_in[2] = {m_prev[0], m_prev[1]};
            if (a.use_viewdirs && l == a.D) {
                // alpha_linear reads the post-ReLU trunk output (nerf.py:86), i.e. this layer's input: one more
                // chunk, a single-row tile accumulated into a pending tile that is no longer needed
                chunk_row8<true>(pipe, cur, pend[0], hid);
                const float da = lds_scalar(layer_tab + 4 * (a.D + 2)), ba = lds_scalar(bias_lds + (8 * a.D) * 32);
                sigma[0] = fmaf(pend[0].v[0][0], da * pow2f(-t_in[0]), ba);
                sigma[1] = fmaf(pend[0].v[2][0], da * pow2f(-t_in[1]), ba);
            }
            if (!(a.use_viewdirs && l == a.D) && ((a.skip_in_mask >> l) & 1)) {
                // h = cat[input_pts, h] (nerf.py:79-80): bring the encoded inputs to this layer's scale, point by point
                const int dt[2] = {t_in[0] - t_pe[0], t_in[1] - t_pe[1]};
                rescale_tile(xp0, dt);
                rescale_tile(xp1, dt);
                t_pe[0] = t_in[0];
                t_pe[1] = t_in[1];
                chunk_ktile8<-1, false>(pipe, cur, out, xp0, hid, pend, pd);
                chunk_ktile8<-1, false>(pipe, cur, out, xp1, hid, pend, pd);
                m_in[0] = fmaxf(m_in[0], m_pe);
                m_in[1] = fmaxf(m_in[1], m_pe);
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

        const bool live[2] = {pt_raw[0] < a.n_points, pt_raw[1] < a.n_points};
        if (a.use_viewdirs) {
            // views_linears[0] on cat[feature, gamma(dir)] (nerf.py:93-98): 4 output tiles; the pending layer is
            // feature_linear
            convert_tile<0>(hid[0], accA[0], pd);
            convert_tile<1>(hid[1], accA[1], pd);
            chunk_pair4<2, true>(pipe, cur, accB, hid[0], hid[1], hid, accA, pd);
            chunk_pair4<4, false>(pipe, cur, accB, hid[2], hid[3], hid, accA, pd);
            chunk_pair4<6, false>(pipe, cur, accB, hid[4], hid[5], hid, accA, pd);
            chunk_pair4<-1, false>(pipe, cur, accB, hid[6], hid[7], hid, accA, pd);
            close_pending();
            XT xd;
            {
                f32x16 x0, x1, dd;
                load_inputs<MODE, false, true>(a, pt, g, x0, x1, dd);
                // gamma(dir) joins the feature tiles at each point's own scale
#pragma unroll
                for (int r = 0; r < 16; ++r) dd[r] *= pd.sc[r >> 3];
                split_tile(xd, dd, 1.0f);
            }
            chunk_ktile4(pipe, cur, accB, xd);
            f32x16 y[4];
            const float dv = lds_scalar(layer_tab + 4 * (a.D + 1));
            const float cv[2] = {dv * pow2f(-pd.t_out[0]), dv * pow2f(-pd.t_out[1])};
            finish_views(y, accB, bias0 + 128 * (8 * a.D + 9), cv);
            // rgb_linear (nerf.py:101): three rows over the 128-wide view layer
            const float* rb = bias_lds + (8 * a.D + 13) * 32;
            float r0[2], r1[2], r2[2];
            row_dot4(y, bias0 + 128 * (8 * a.D + 22), r0);
            row_dot4(y, bias0 + 128 * (8 * a.D + 26), r1);
            row_dot4(y, bias0 + 128 * (8 * a.D + 30), r2);
            const float b0 = lds_scalar(rb), b1 = lds_scalar(rb + 1), b2 = lds_scalar(rb + 2);
            if (g == 0) {
#pragma unroll
                for (int P = 0; P < 2; ++P)
                    if (live[P]) {
                        f32x4 o = {r0[P] + b0, r1[P] + b1, r2[P] + b2, sigma[P]};   // cat[rgb, alpha] (nerf.py:106)
                        *(f32x4*)(a.out + pt[P] * 4) = o;
                    }
            }
        } else {
            // output_linear (nerf.py:109): rows 0..out_ch-1 of one tile; the pending layer is trunk layer D-1
            convert_tile<0>(hid[0], accA[0], pd);
            convert_tile<1>(hid[1], accA[1], pd);
            convert_tile<2>(hid[2], accA[2], pd);
            convert_tile<3>(hid[3], accA[3], pd);
            convert_tile<4>(hid[4], accA[4], pd);
            convert_tile<5>(hid[5], accA[5], pd);
            convert_tile<6>(hid[6], accA[6], pd);
            convert_tile<7>(hid[7], accA[7], pd);
            Acc o;
            chunk_row8<false>(pipe, cur, o, hid);
            Oct b = lds_oct_issue(bias0 + 128 * (8 * a.D));
            lds_oct_wait(b);
            const float dz = lds_scalar(layer_tab + 4 * a.D);
#pragma unroll
            for (int P = 0; P < 2; ++P) {
                const float c = dz * pow2f(-pd.t_out[P]);
                if (live[P]) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const int row = 16 * (e >> 2) + 4 * g + (e & 3);
                        if (row < a.out_ch) a.out[pt[P] * a.out_ch + row] = fmaf(acc_reg(o, 8 * P + e), c, b.q[e >> 2][e & 3]);
                    }
                }
            }
        }
    }   // tile loop
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

hipError_t launch_mlp_h3(const MlpLaunch& a, int mode, hipStream_t s) {
    if (a.n_points <= 0) return hipSuccess;
    if (!a.stream_h3 || !a.bias3 || !a.descale || !a.gain) return hipErrorInvalidValue;
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
        const void* fn = mode == kInputEmbedded ? (const void*)nerf_mlp_h3_kernel<kInputEmbedded>
                         : mode == kInputPoints ? (const void*)nerf_mlp_h3_kernel<kInputPoints>
                                                : (const void*)nerf_mlp_h3_kernel<kInputRays>;
        e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        raised[dev][mode] = true;
    }
    switch (mode) {
        case kInputEmbedded:
            hipLaunchKernelGGL(nerf_mlp_h3_kernel<kInputEmbedded>, grid, block, lds, s, a);
            break;
        case kInputPoints:
            hipLaunchKernelGGL(nerf_mlp_h3_kernel<kInputPoints>, grid, block, lds, s, a);
            break;
        default:
            hipLaunchKernelGGL(nerf_mlp_h3_kernel<kInputRays>, grid, block, lds, s, a);
            break;
    }
    return hipGetLastError();
}

}  // namespace nerf
