// Fused backward-data pass of the training step in "fp16-pair" arithmetic (loss.backward() through
// NeRF.forward, nerf/nerf.py:57-111, as the training loop runs it at nerf.ipynb:1263-1275).
//
// The same chain as nerf_mlp_bwd_kernel (mlp_kernel.hip) - d raw -> the gradient at every pre-activation, transposed weights
// streamed L2 -> LDS by LDS-DMA, the running gradient chained in registers from layer to layer - on the machinery of the
// forward fp16-pair kernel (mlp_pair_common.h: four-slot weight ring, hand-placed steps of six v_mfma_f32_32x32x16_f16,
// every fp32 operand as an exact (hi, lo) pair of halves, one power-of-two scale per layer for the weights and per POINT
// for the gradient vector, chosen before the layer's outputs exist from a bound and counted when the bound was loose):
//
//   d(view pre) = (d rgb . W_rgb) * [hv > 0]                in the open: vector dot products, W_rgb rows from LDS
//   d feature   = W_views[:, :W]^T d(view pre)               4 chunks
//   d h_{D-1}   = W_feature^T d feature + w_alpha d sigma    8 chunks + 1: d sigma rides as one more operand value
//   d z_i       = d h_i * [h_i > 0];  d h_{i-1} = W_i[:, hidden]^T d z_i       8 chunks per trunk layer
//
// A layer's raw sums become the next layer's operands one tile per chunk of that next layer, in the shadow of its MFMAs
// (scale, ReLU mask, running maximum, scale, split); each masked gradient also goes to memory from there, four consecutive
// features per 16-byte store, for the weight-gradient kernel (dW = dZ^T X, train_dw_kernel.hip).
//
// Units. The pass runs on the ROW-EQUALISED network the fp16-pair forward kernel evaluated (PackedNet::d_params_eq: unit j
// scaled by 2^e_j, the columns that read it by 2^-e_j - the same function), i.e. on its transposed weights, and writes
// that network's gradients d z' = 2^-e d z; the weight gradients are brought back to the plain parameters' by exact powers
// of two where their slices are added up (GradJob::ex). Equalised rows are what keeps one huge unit from costing the
// others their low halves - in this direction as in the forward one.
//
// ReLU masks. The reference's autograd keeps the post-ReLU activations; the forward kernel keeps them too (the weight
// gradients need them) but the mask is one BIT of each: it writes 16 bytes per point, half-wave and layer (MlpStore::mask)
// and this kernel reads those - by LDS-DMA a whole layer ahead, no registers in flight - instead of a kilobyte.
#define NERF_FRAG_VGPR      // this kernel has the vector registers the forward kernel gives to the encoded inputs: A fragments there
#include "mlp_pair_common.h"

namespace nerf {

struct PendingB {
    float c;        // raw sum -> gradient, in the equalised network's units: descale * 2^-t_in (per point)
    float sc;       // gradient -> operand: 2^t_out (per point)
    int t_out;
    float m;        // running max |gradient|
    float* keep_base;      // where this layer's masked gradient goes ...
    unsigned keep_off;     // ... and this lane's BYTE offset of (its point, feature 4 h) in it
    u32x4 mask;     // this layer's ReLU mask words (all ones for d feature: feature_linear has no ReLU)
};

// register pair S of pending tile T: raw sums -> masked gradient (y0, y1)
template <int T, int S>
__device__ __forceinline__ void bconv0(ConvTmp& t, const f32x16& src, const PendingB& pd) {
    const unsigned w = pd.mask[T >> 1];
    const float g0 = src[2 * S] * pd.c, g1 = src[2 * S + 1] * pd.c;
    t.y0 = mask_apply<T, S, 0>(w, g0);
    t.y1 = mask_apply<T, S, 1>(w, g1);
}
__device__ __forceinline__ void bconv1(ConvTmp& t, PendingB& pd) {
    pd.m = fmaxf(fmaxf(pd.m, fabsf(t.y0)), fabsf(t.y1));
    t.a0 = t.y0 * pd.sc;
    t.a1 = t.y1 * pd.sc;
}
// BLK: the layout blocked by 32 points (MlpStore::blocked; mlp_kernel_h2.hip keep_pairs): a store instruction writes one
// contiguous KiB - piece (T, Q) of the point group - with the nt bit
template <int T, int Q, bool BLK>
__device__ __forceinline__ void bkeep(const PendingB& pd, const f32x2& even, float y0, float y1) {
    // (keep_base: the start of the ODD tile of the pair being converted, moved on in place by next_tile_pair - one scalar base
    // and the signed immediate reach two tiles; a base per tile cost the forward kernel its last scalar registers)
    if constexpr (BLK) keep_quad_nt<1024 * Q - ((T & 1) ? 0 : 4096)>(pd.keep_base, pd.keep_off, f32x4{even[0], even[1], y0, y1});
    else keep_quad<(32 * T + 8 * Q) * 4>(pd.keep_base, pd.keep_off, f32x4{even[0], even[1], y0, y1});
}

// One k-tile against 8 output tiles; CONV >= 0: while the chunk runs, step s converts register pair s of pending tile CONV.
template <bool BLK>
__device__ __forceinline__ void next_tile_pair(PendingB& pd) {
    if constexpr (BLK) pd.keep_base += 2048;
}
template <int CONV, bool FIRST, bool BLK>
__device__ __forceinline__ void chunk_bwd(PipeH& p, Frag4& cur, f32x16 (&acc)[8], const XT& x, XT (&hid)[8],
                                          const f32x16 (&pend)[8], PendingB& pd) {
    constexpr int C0 = CONV < 0 ? 0 : CONV;
    ConvTmp t;
    f32x2 even;
    consume_chunk<8, 0>(p, cur, [&](auto tag, auto part, const Frag4& f) {
        constexpr int s = decltype(tag)::value, pt = decltype(part)::value;
        if constexpr (pt < 6) mma_one<pt, FIRST>(acc[s], f, x);
        else if constexpr (CONV >= 0) {
            if constexpr (pt == 11) {
                bconv0<C0, s>(t, pend[C0], pd);
                if constexpr ((s & 1) == 0) even = f32x2{t.y0, t.y1};
                else bkeep<C0, (s >> 1), BLK>(pd, even, t.y0, t.y1);
            } else if constexpr (pt == 12) bconv1(t, pd);
            else if constexpr (pt == 13) conv_slice2<s>(hid[C0], t);
        }
    });
}

// a whole tile in the open (tile 0 at the start of a layer; every tile of d z_0, which nothing follows)
template <int T, int S, bool SPLIT, bool BLK>
__device__ __forceinline__ void bconvert_pairs(XT& dst, const f32x16& src, PendingB& pd, f32x2& even) {
    if constexpr (S < 8) {
        ConvTmp t;
        bconv0<T, S>(t, src, pd);
        if constexpr ((S & 1) == 0) even = f32x2{t.y0, t.y1};
        else bkeep<T, (S >> 1), BLK>(pd, even, t.y0, t.y1);
        if constexpr (SPLIT) {
            bconv1(t, pd);
            conv_slice2<S>(dst, t);
        } else {
            pd.m = fmaxf(fmaxf(pd.m, fabsf(t.y0)), fabsf(t.y1));      // (d z_0: nothing follows, but its weight gradients scale by its size)
        }
        bconvert_pairs<T, S + 1, SPLIT, BLK>(dst, src, pd, even);
    }
}
template <int T, bool SPLIT, bool BLK>
__device__ __forceinline__ void bconvert_tile(XT& dst, const f32x16& src, PendingB& pd) {
    f32x2 even;
    bconvert_pairs<T, 0, SPLIT, BLK>(dst, src, pd, even);
}

constexpr int kPreSlotBytes = 1024;      // one 16-byte record per lane
constexpr int kRgbRowFloats = 12 * kBiasTileFloats;

// VIEWS = false: networks without view directions (output_linear on the trunk, nerf.py:109; round 4). The chain starts at
// d h_{D-1} = W_output^T d raw: ONE chunk - the head's C <= 8 rows as columns 0..C-1 of a k-tile, d raw as the operand (each
// half-wave holds four channels: d raw is [P, 8], zero-padded) - where the view-dependent chain spends thirteen on the view
// layer, feature_linear and the alpha column. Slots of the per-wave records: [0] d raw and [1] the mask of trunk layer D-1, both
// a tile ahead.
template <bool BLK, bool VIEWS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void nerf_mlp_bwd_h2_kernel(const MlpBwdLaunch b) {
    extern __shared__ __attribute__((aligned(16))) char ring_lds[];
    __shared__ __attribute__((aligned(16))) float rgb_lds[kRgbRowFloats];             // rgb_linear's three rows, per register
    __shared__ __attribute__((aligned(16))) float layer_tab[4 * (kMaxDepth + 2)];     // per layer [descale, gain, alpha gain, -]
    // per wave: the records this kernel reads per point, fetched by LDS-DMA ahead of their use:
    //   [0] d raw of the tile (one tile ahead)   [1] the view layer's mask (one tile ahead)   [2] a trunk layer's mask (requested
    //   when the layer's chunks begin, read when they end)
    __shared__ __attribute__((aligned(16))) char pre_lds[kWavesPerGroup][3][kPreSlotBytes];
    __shared__ unsigned max_record[kBwdMaxSlots];                      // enter_max: this workgroup's maxima, flushed at the end
    __shared__ unsigned loose_hist[kLooseRecord];                       // this workgroup's loose-bound events, flushed at the end
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane >> 5;
    const int D = __builtin_amdgcn_readfirstlane(b.D);      // (explicitly scalar: it indexes the launch record)

    PipeH pipe{(const char*)b.stream_h2, ring_lds, 0, 0, b.n_chunks, wave, lane, nullptr, nullptr, nullptr, nullptr, 0, nullptr, nullptr, 0};
    pipe_start(pipe);
    for (int k = 0; k < 2; ++k) {
        prefetch_pieces<0, 4>(piece_src(pipe, k), piece_dst(pipe, k));
        prefetch_pieces<0, 4>(piece_src(pipe, k) + 4096, piece_dst(pipe, k) + 4096);
    }
    prefetch_pieces<0, 4>(piece_src(pipe, 2), piece_dst(pipe, 2));   // chunk 0's first-half steps issue the other four
    // rgb_linear's rows: bias-block tiles 8D+22 .. 8D+33 (pack_weights.cpp row_tiles)
    if constexpr (VIEWS)
        for (int i = threadIdx.x; i < kRgbRowFloats; i += 256) rgb_lds[i] = b.bias[(8 * D + 22) * kBiasTileFloats + i];
    if (threadIdx.x < kBwdMaxSlots) max_record[threadIdx.x] = 0u;
    if (threadIdx.x < kLooseRecord) loose_hist[threadIdx.x] = 0u;
    if ((int)threadIdx.x <= D) {
        const int l = threadIdx.x;
        layer_tab[4 * l] = b.descale[l];
        layer_tab[4 * l + 1] = b.gain[2 * l];
        layer_tab[4 * l + 2] = b.gain[2 * l + 1];
        layer_tab[4 * l + 3] = 0.0f;
    }

    const int64_t n_tiles = (b.n_points + kPointsPerGroup - 1) / kPointsPerGroup;
    auto point_of = [&](int64_t tile) {
        const int64_t raw = tile * kPointsPerGroup + wave * kPointsPerWave + (lane & 31);
        return raw < b.n_points ? raw : b.n_points - 1;      // padded lanes recompute the last point
    };
    char* const my_pre = &pre_lds[0][0][0] + wave * 3 * kPreSlotBytes;
    auto fetch = [&](const void* g, int slot) {
        __builtin_amdgcn_global_load_lds(GLB_PTR(g), LDS_PTR(my_pre + slot * kPreSlotBytes), 16, 0, 0);
    };
    auto fetched = [&](int slot) { return (const float*)(my_pre + slot * kPreSlotBytes + lane * 16); };
    auto mask_rec = [&](const unsigned* base, int64_t pt) { return base + 4 * (2 * pt + h); };
    // this lane's 16 bytes of d raw: the point's four channels (VIEWS), or the four of this half-wave out of eight
    auto d_raw_rec = [&](int64_t pt) { return VIEWS ? b.d_raw + pt * 4 : b.d_raw + pt * 8 + 4 * h; };
    auto first_mask = [&](int64_t pt) { return VIEWS ? mask_rec(b.fwd.mask_hv, pt) : mask_rec(b.fwd.mask[D - 1], pt); };
    if ((int64_t)blockIdx.x < n_tiles) {
        const int64_t pt0 = point_of(blockIdx.x);
        fetch(d_raw_rec(pt0), 0);
        fetch(first_mask(pt0), 1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();   // chunks 0, 1, the rgb rows, the layer table and the first tile's records are in LDS
    Frag4 cur;
    {
        const unsigned fr0 = lds_byte_addr(ring_lds) + lane * 16;
        frag_issue<0>(cur.q[0], fr0);
        frag_issue<1024>(cur.q[1], fr0);
        frag_issue<2048>(cur.q[2], fr0);
        frag_issue<3072>(cur.q[3], fr0);
    }
    const unsigned rgb0 = lds_addr(rgb_lds) + 64 * h;   // this half-wave's entries of tile 0 of row 0 (row c, tile t: + 128 (4 c + t))

#ifdef NERF_EXP_STAGGER      // timing experiment (profiles/r04_ab_notes.txt): workgroups out of phase with each other, so that the chip's
    // thousand waves do not issue their stores (and their weight-stream loads) in the same instants
    for (int k = 0; k < (int)((blockIdx.x >> 3) & 7); ++k) __builtin_amdgcn_s_sleep(NERF_EXP_STAGGER);
#endif
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        pipe_tile_start(pipe);
        const int64_t pt = point_of(tile);
        const int64_t pt_next = point_of(tile + gridDim.x < n_tiles ? tile + gridDim.x : tile);

        XT hid[8];
        f32x16 accA[8], accB[8];
        PendingB pd;
        float m_prev, dsig_abs, dsig_h0;

        f32x4 dr_head = {0.0f, 0.0f, 0.0f, 0.0f};      // (!VIEWS) this half-wave's four channels of d raw
        int t_head = 0;
        if constexpr (VIEWS) {
        // ---- d(view pre-activation) = (d rgb . W_rgb) * [hv > 0], in the open (nerf.py:101, :96-98) ----
            {
                const f32x4 dr = lds_vec4(fetched(0));
                const f32x4 mh = lds_vec4(fetched(1));
                const unsigned mw[2] = {__float_as_uint(mh[0]), __float_as_uint(mh[1])};
                dsig_abs = fabsf(dr[3]);
                dsig_h0 = h == 0 ? dr[3] : 0.0f;
                f32x16 g[4];
                float m = 0.0f;
                // (blocked: 16 KiB per point group of this 128-wide buffer)
                const unsigned off = BLK ? ((unsigned)pt >> 5) * 16384u + ((unsigned)pt & 31u) * 32u + (unsigned)h * 16u
                                         : 4u * ((unsigned)pt * (unsigned)b.out.hv_ld + 4u * (unsigned)h);
                Tile16 w0 = lds_tile_issue(rgb0), w1 = lds_tile_issue(rgb0 + 128 * 4), w2 = lds_tile_issue(rgb0 + 128 * 8);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    lds_tile_wait(w0);
                    lds_tile_wait(w1);
                    lds_tile_wait(w2);
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float v = fmaf(dr[2], w2.q[r >> 2][r & 3], fmaf(dr[1], w1.q[r >> 2][r & 3], dr[0] * w0.q[r >> 2][r & 3]));
                        // value r = 2s (+1) of tile t: bit 15 (31) - 8 (t & 1) - s of word t / 2
                        const int bit = ((r & 1) ? 31 : 15) - 8 * (t & 1) - (r >> 1);
                        g[t][r] = ((mw[t >> 1] >> bit) & 1u) ? v : 0.0f;
                        m = fmaxf(m, fabsf(g[t][r]));
                    }
                    if (t + 1 < 4) {
                        w0 = lds_tile_issue(rgb0 + 128 * (t + 1));
                        w1 = lds_tile_issue(rgb0 + 128 * (4 + t + 1));
                        w2 = lds_tile_issue(rgb0 + 128 * (8 + t + 1));
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x4 v = {g[t][4 * q], g[t][4 * q + 1], g[t][4 * q + 2], g[t][4 * q + 3]};
                        if constexpr (BLK)
                            asm volatile("global_store_dwordx4 %0, %1, %2 nt\n\ts_nop 1"
                                         :
                                         : "v"(off + (unsigned)((4 * t + q) * 1024)), "v"(v), "s"(b.out.hv)
                                         : "memory");
                        else
                            asm volatile("global_store_dwordx4 %0, %1, %2\n\ts_nop 1"
                                         :
                                         : "v"(off + (unsigned)((32 * t + 8 * q) * 4)), "v"(v), "s"(b.out.hv)
                                         : "memory");
                    }
                }
                m_prev = half_max(m);
                enter_max(&max_record[kBwdMaxViews], m_prev);
                const int t_v = pick_exponent(m_prev);
                const float sc = pow2f(t_v);
#pragma unroll
                for (int t = 0; t < 4; ++t) split_tile(hid[t], g[t], sc);
                pd.t_out = t_v;
            }

        } else {
            // ---- the operand of the head's chunk: d raw at its own per-point scale ----
            dr_head = lds_vec4(fetched(0));
            dsig_abs = 0.0f;
            dsig_h0 = 0.0f;
            m_prev = half_max(fmaxf(fmaxf(fabsf(dr_head[0]), fabsf(dr_head[1])), fmaxf(fabsf(dr_head[2]), fabsf(dr_head[3]))));
            t_head = pick_exponent(m_prev);
            pd.t_out = t_head;
        }

        // what the raw sums of backward layer bl become: called when its chunks are done. m_in = largest |input| of the layer
        // (this point), t_in = exponent its inputs were scaled by
        auto make_pending = [&](int bl_, float m_in, int t_in) {
            const int bl = __builtin_amdgcn_readfirstlane(bl_);
            f32x4 tab, mk;
            // (the mask of the layer these sums belong to: requested when its chunks began - slot 2 - or, for the no-views chain's
            // first layer, a tile ahead - slot 1)
            const float* mask_at = (!VIEWS && bl == 1) ? fetched(1) : fetched(2);
            asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %3\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(tab), "=&v"(mk)
                         : "v"(lds_byte_addr(layer_tab + 4 * bl)), "v"(lds_byte_addr(mask_at))
                         : "memory");
            pd.c = tab[0] * pow2f(-t_in);
            float bound = tab[1] * m_in;
            if (bl == 0) bound = fmaxf(bound, dsig_abs);          // the alpha column's operand shares this layer's scale
            if (bl == 1) bound = fmaf(tab[2], dsig_abs, bound);   // d h_{D-1} = W_feature^T d feature + w_alpha d sigma
            pd.t_out = pick_exponent(bound * 1.001f);
            pd.sc = pow2f(pd.t_out);
            pd.m = 0.0f;
            if (bl == 0) {
                pd.mask = u32x4{~0u, ~0u, ~0u, ~0u};
                pd.keep_base = wave_uniform(b.out.feat) + (BLK ? 1024 : 0);
                pd.keep_off = 4u * ((unsigned)pt * (unsigned)b.out.feat_ld + 4u * (unsigned)h);
                if constexpr (BLK) pd.keep_off = ((unsigned)pt >> 5) * 32768u + ((unsigned)pt & 31u) * 32u + (unsigned)h * 16u;
            } else {
                pd.mask = u32x4{__float_as_uint(mk[0]), __float_as_uint(mk[1]), __float_as_uint(mk[2]), __float_as_uint(mk[3])};
                pd.keep_base = wave_uniform(b.out.h[D - bl]) + (BLK ? 1024 : 0);
                pd.keep_off = 4u * ((unsigned)pt * (unsigned)b.out.h_ld[D - bl] + 4u * (unsigned)h);
                if constexpr (BLK) pd.keep_off = ((unsigned)pt >> 5) * 32768u + ((unsigned)pt & 31u) * 32u + (unsigned)h * 16u;
            }
        };
        auto close_pending = [&](int slot) {
            m_prev = half_max(pd.m);
            // how far the bound overshot (d sigma shares d feature's scale group: it counts as a member). Counted by size
            // (nerf_precision_detail); from 2^kLooseBwdGuard on also where the precision guard looks.
            const float m_grp = slot == kBwdMaxFeat ? fmaxf(m_prev, dsig_abs) : m_prev;
            const int slack = 10 - pd.t_out - __builtin_amdgcn_frexp_expf(m_grp);
            // Into LDS (a per-lane atomic on eight global addresses, a quarter of all lanes firing, queued in the memory pipe
            // the weight ring's counted waits look at: it doubled the kernel's time); the workgroup adds its sums at the end.
            if (m_grp > 0.0f && slack >= 12 && pd.t_out > -60) {
#ifdef NERF_LOOSE_BY_LAYER      // diagnostic build (tools/gpu/loose_probe.py): events of >= 2^24 by the slot they close, others in bucket 1
                const int bucket = slack >= 24 ? 2 + (slot == kBwdMaxFeat ? 0 : (slot >= 6 ? 1 : (slot >= 3 ? 2 : (slot >= 1 ? 3 : 4)))) : 1;
#else
                const int bucket = 1 + (slack >= 24 ? 6 : (slack - 12) >> 1);
#endif
                const unsigned one = 1u;
                asm volatile("ds_add_u32 %0, %1" : : "v"(lds_byte_addr(loose_hist + bucket)), "v"(one) : "memory");
                if (slack >= kLooseBwdGuard) asm volatile("ds_add_u32 %0, %1" : : "v"(lds_byte_addr(loose_hist)), "v"(one) : "memory");
            }
            if (slot >= 0) enter_max(&max_record[slot], m_prev);
        };

        XT xs;      // (filled where it is first needed: sixteen registers the four chunks before it cannot spare)
        auto clear_xs = [&]() {
            xs.hi[0] = u32x4{0u, 0u, 0u, 0u};
            xs.hi[1] = xs.hi[0];
            xs.lo[0] = xs.hi[0];
            xs.lo[1] = xs.hi[0];
        };
        if constexpr (VIEWS) {
            // ---- d feature = W_views[:, :W]^T d(view pre-activation): four k-tiles, nothing pending yet ----
            chunk_bwd<-1, true, BLK>(pipe, cur, accA, hid[0], hid, accB, pd);
            chunk_bwd<-1, false, BLK>(pipe, cur, accA, hid[1], hid, accB, pd);
            chunk_bwd<-1, false, BLK>(pipe, cur, accA, hid[2], hid, accB, pd);
            chunk_bwd<-1, false, BLK>(pipe, cur, accA, hid[3], hid, accB, pd);
            make_pending(0, m_prev, pd.t_out);
            // the alpha column's operand: d sigma as value 0 of half-wave 0 (the column k = 0 of its k-tile), at d feature's scale
            clear_xs();
            split_pair<0>(xs, dsig_h0 * pd.sc, 0.0f);
        } else {
            // ---- d h_{D-1} = W_output^T d raw: one k-tile whose columns 0..3 (half-wave 0) and 4..7 (half-wave 1) are the channels ----
            const float sc = pow2f(t_head);
            clear_xs();
            split_pair<0>(xs, dr_head[0] * sc, dr_head[1] * sc);
            split_pair<1>(xs, dr_head[2] * sc, dr_head[3] * sc);
            chunk_bwd<-1, true, BLK>(pipe, cur, accA, xs, hid, accB, pd);
            make_pending(1, m_prev, t_head);
        }

        // ---- backward layers 1 .. D: layer bl accumulates into `out` while the pending layer bl - 1 is converted ----
        auto layer_pass = [&](f32x16 (&pend)[8], f32x16 (&out)[8], int bl_) {
            const int bl = __builtin_amdgcn_readfirstlane(bl_);
            bconvert_tile<0, true, BLK>(hid[0], pend[0], pd);
            // this layer's outputs will want their ReLU mask at its end: trunk layer D - bl's, requested now
            fetch(mask_rec(b.fwd.mask[D - bl], pt), 2);
            if (bl == D) {      // the last layer: the next tile's records (their slots were read in this tile's prologue)
                fetch(d_raw_rec(pt_next), 0);
                fetch(first_mask(pt_next), 1);
            }
            chunk_bwd<1, true, BLK>(pipe, cur, out, hid[0], hid, pend, pd);
            next_tile_pair<BLK>(pd);
            chunk_bwd<2, false, BLK>(pipe, cur, out, hid[1], hid, pend, pd);
            chunk_bwd<3, false, BLK>(pipe, cur, out, hid[2], hid, pend, pd);
            next_tile_pair<BLK>(pd);
            chunk_bwd<4, false, BLK>(pipe, cur, out, hid[3], hid, pend, pd);
            chunk_bwd<5, false, BLK>(pipe, cur, out, hid[4], hid, pend, pd);
            next_tile_pair<BLK>(pd);
            chunk_bwd<6, false, BLK>(pipe, cur, out, hid[5], hid, pend, pd);
            chunk_bwd<7, false, BLK>(pipe, cur, out, hid[6], hid, pend, pd);
            chunk_bwd<-1, false, BLK>(pipe, cur, out, hid[7], hid, pend, pd);
            close_pending(bl == 1 ? kBwdMaxFeat : D - bl + 1);      // (pending: d feature, then d z_{D - bl + 1})
            if (VIEWS && bl == 1) chunk_bwd<-1, false, BLK>(pipe, cur, out, xs, hid, pend, pd);      // + w_alpha d sigma (nerf.py:86)
            make_pending(bl, m_prev, pd.t_out);
        };
        // (the pending sums are in accA either way: d feature with backward layer 1 next, or - without view directions -
        // d h_{D-1} with layer 2 next; the two accumulator sets alternate statically)
        int bl = VIEWS ? 1 : 2;
        bool pend_in_a = true;
        while (bl <= D) {
            layer_pass(accA, accB, bl);
            ++bl;
            pend_in_a = false;
            if (bl > D) break;
            layer_pass(accB, accA, bl);
            ++bl;
            pend_in_a = true;
        }
        if (!pend_in_a) {
#pragma unroll
            for (int t = 0; t < 8; ++t) accA[t] = accB[t];
        }
        // ---- d z_0: nothing follows to hide behind ----
        pd.keep_base = wave_uniform(b.out.h[0]) + (BLK ? 1024 : 0);      // (what make_pending(D) chose, as a value hipcc keeps in scalar registers)
        pd.m = 0.0f;
        bconvert_tile<0, false, BLK>(hid[0], accA[0], pd);
        bconvert_tile<1, false, BLK>(hid[0], accA[1], pd);
        next_tile_pair<BLK>(pd);
        bconvert_tile<2, false, BLK>(hid[0], accA[2], pd);
        bconvert_tile<3, false, BLK>(hid[0], accA[3], pd);
        next_tile_pair<BLK>(pd);
        bconvert_tile<4, false, BLK>(hid[0], accA[4], pd);
        bconvert_tile<5, false, BLK>(hid[0], accA[5], pd);
        next_tile_pair<BLK>(pd);
        bconvert_tile<6, false, BLK>(hid[0], accA[6], pd);
        bconvert_tile<7, false, BLK>(hid[0], accA[7], pd);
        enter_max(&max_record[0], half_max(pd.m));
    }   // tile loop
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    if (b.loose && threadIdx.x < kLooseRecord && loose_hist[threadIdx.x]) atomicAdd(b.loose + threadIdx.x, loose_hist[threadIdx.x]);
    flush_maxes(b.maxes, max_record, kBwdMaxSlots);
}

hipError_t launch_mlp_bwd_h2(const MlpBwdLaunch& b, hipStream_t s) {
    if (b.n_points <= 0) return hipSuccess;
    if (!b.stream_h2 || !b.descale || !b.gain || !b.bias || !b.d_raw) return hipErrorInvalidValue;
    const bool views = b.use_viewdirs != 0;
    // view-dependent: 4 + 8 + 1 chunks ahead of the trunk's, d raw [P, 4]; otherwise the head's one chunk, d raw [P, 8] (zero-padded)
    if (b.D < (views ? 1 : 2) || b.D > kMaxDepth || b.n_chunks != (views ? 13 : 1) + 8 * (b.D - 1)) return hipErrorInvalidValue;
    if (views ? b.C != 4 : (b.C < 1 || b.C > 8 || b.d_raw_ld != 8)) return hipErrorInvalidValue;
    if ((reinterpret_cast<uintptr_t>(b.d_raw) & 15) != 0) return hipErrorInvalidValue;
    // one-instruction 16-byte stores: aligned rows, byte offsets below 2^32; 16-byte mask records
    auto ok = [&](const float* p, int ld) {
        return p != nullptr && (ld & 3) == 0 && (reinterpret_cast<uintptr_t>(p) & 15) == 0 &&
               (uint64_t)b.n_points * (uint64_t)ld * 4u < ((uint64_t)1 << 32);
    };
    bool rows_ok = !views || (ok(b.out.hv, b.out.hv_ld) && ok(b.out.feat, b.out.feat_ld) && b.fwd.mask_hv &&
                              (reinterpret_cast<uintptr_t>(b.fwd.mask_hv) & 15) == 0);
    for (int i = 0; i < b.D; ++i)
        rows_ok = rows_ok && ok(b.out.h[i], b.out.h_ld[i]) && b.fwd.mask[i] && (reinterpret_cast<uintptr_t>(b.fwd.mask[i]) & 15) == 0;
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
    const size_t lds = kRingH * kChunkBytes;
    static bool raised[64][4] = {};
    const int blk = b.out.blocked ? 1 : 0, which = 2 * (views ? 1 : 0) + blk;
    const void* fn = views ? (blk ? (const void*)nerf_mlp_bwd_h2_kernel<true, true> : (const void*)nerf_mlp_bwd_h2_kernel<false, true>)
                           : (blk ? (const void*)nerf_mlp_bwd_h2_kernel<true, false> : (const void*)nerf_mlp_bwd_h2_kernel<false, false>);
    if (!raised[dev][which]) {
        e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        raised[dev][which] = true;
    }
    if (views) {
        if (blk) hipLaunchKernelGGL((nerf_mlp_bwd_h2_kernel<true, true>), grid, block, lds, s, b);
        else hipLaunchKernelGGL((nerf_mlp_bwd_h2_kernel<false, true>), grid, block, lds, s, b);
    } else {
        if (blk) hipLaunchKernelGGL((nerf_mlp_bwd_h2_kernel<true, false>), grid, block, lds, s, b);
        else hipLaunchKernelGGL((nerf_mlp_bwd_h2_kernel<false, false>), grid, block, lds, s, b);
    }
    return hipGetLastError();
}

// gain[2b] = largest row sum of |W^T| of backward layer b (= largest column sum of |W| over the block the layer
// contracts), gain[2b + 1] = largest |alpha weight| for b = 1. One workgroup per layer, a thread per column (coalesced
// along the rows of W).
__global__ __launch_bounds__(1024) void layer_gain_bwd_kernel(const float* params, const BwdGainRefs refs, float* gain) {
    __shared__ float part[4][256];
    __shared__ float red[2][4];
    const int l = blockIdx.x, r = threadIdx.x & 255, q = threadIdx.x >> 8;      // column r, row quarter q
    const float* w = params + refs.w_off[l] + refs.col0[l] + r;
    const int n_rows = refs.rows[l], per = (n_rows + 3) / 4;
    float sum = 0.0f;
    for (int c0 = q * per; c0 < (q + 1) * per; c0 += 8) {      // eight loads in flight
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {      // (unconditional, at a clamped row: behind a condition each load is a branch with its own wait)
            const int c = c0 + u;
            v[u] = w[(size_t)(c < n_rows ? c : n_rows - 1) * refs.ld[l]];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) sum += (c0 + u < (q + 1) * per && c0 + u < n_rows) ? fabsf(v[u]) : 0.0f;
    }
    part[q][r] = sum;
    __syncthreads();
    if (threadIdx.x < 256) {
        sum = part[0][r] + part[1][r] + part[2][r] + part[3][r];
        float am = (l == 1 && refs.alpha_off != 0xffffffffu) ? fabsf(params[refs.alpha_off + r]) : 0.0f;
        for (int o = 32; o > 0; o >>= 1) {
            sum = fmaxf(sum, __shfl_xor(sum, o));
            am = fmaxf(am, __shfl_xor(am, o));
        }
        if ((threadIdx.x & 63) == 0) {
            red[0][threadIdx.x >> 6] = sum;
            red[1][threadIdx.x >> 6] = am;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        gain[2 * l] = fmaxf(fmaxf(red[0][0], red[0][1]), fmaxf(red[0][2], red[0][3]));
        gain[2 * l + 1] = fmaxf(fmaxf(red[1][0], red[1][1]), fmaxf(red[1][2], red[1][3]));
    }
}

BwdGainRefs bwd_gain_refs(const nerf_arch& a, const std::vector<LinearDesc>& linears, uint32_t skip_in_mask) {
    BwdGainRefs r{};
    r.n = a.D + 1;
    auto set = [&](int b, const LinearDesc& d, int col0) {
        r.w_off[b] = (unsigned)d.w_off;
        r.ld[b] = d.in;
        r.rows[b] = d.out;
        r.col0[b] = col0;
    };
    if (a.use_viewdirs) {
        const LinearDesc &views = linears[a.D], &feat = linears[a.D + 1], &alpha = linears[a.D + 2];
        set(0, views, 0);
        set(1, feat, 0);
        r.alpha_off = (unsigned)alpha.w_off;
    } else {
        // the chain starts at backward layer 1 = W_output^T (linears[D + 1]); layer 0 does not exist (its gain is unused)
        set(0, linears[a.D + 1], 0);
        set(1, linears[a.D + 1], 0);
        r.alpha_off = 0xffffffffu;      // (no alpha column: layer_gain_bwd_kernel leaves gain[3] at 0)
    }
    for (int b = 2; b <= a.D; ++b) {
        const int i = a.D - b + 1;
        set(b, linears[i], ((skip_in_mask >> i) & 1) ? a.input_ch : 0);
    }
    return r;
}

hipError_t launch_layer_gains_bwd(const float* params, const BwdGainRefs& refs, float* gain, hipStream_t s) {
    if (refs.n <= 0) return hipSuccess;
    hipLaunchKernelGGL(layer_gain_bwd_kernel, dim3(refs.n), dim3(1024), 0, s, params, refs, gain);
    return hipGetLastError();
}

}  // namespace nerf
