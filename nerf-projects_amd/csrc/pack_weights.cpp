// Host-side repacking of a NeRF state_dict (nerf/nerf.py:32-55) into the MFMA fragment
// stream consumed by the fused encode+MLP kernel (mlp_kernel.hip).
//
// Orientation. The kernel computes H_out^T [features x points] = W [out x in] * H_in^T, so
// a weight matrix is the MFMA *A* operand and activations never leave the accumulator
// layout: for v_mfma_f32_32x32x2_f32 the D tile holds C[row][col] with col = lane & 31
// (the point) and row = (reg & 3) + 8*(reg >> 2) + 4*(lane >> 5) (the feature), and the B
// operand of k-step t wants B[k = lane >> 5][col = lane & 31]. Feeding accumulator register
// t of an activation tile as the B operand therefore contracts over feature
//     f(tile, t, h) = 32*tile + (t & 3) + 8*(t >> 2) + 4*h,   h = lane >> 5,
// with no cross-lane movement, provided the A operand of that k-step holds
// W[out_row = 32*ot + (lane & 31)][f(tile, t, h)]. That permutation is applied here, once.
//
// Positional-encoding tiles use their own slot map (pe_col_* in nerf_internal.h): half-wave h = 0 holds
// the sines, h = 1 the cosines; the two half-waves split the frequencies between them.
#include <cstdlib>
#include <cstring>
#include <vector>

#include "nerf_internal.h"

namespace nerf {

namespace {

inline int hidden_col(int tile, int t, int h) { return 32 * tile + (t & 3) + 8 * (t >> 2) + 4 * h; }

struct Stream {
    std::vector<float> data;
    float* new_chunk() {
        data.resize(data.size() + kChunkFloats, 0.0f);
        return data.data() + data.size() - kChunkFloats;
    }
};

struct Linear {
    const float* w;   // [out, in] row-major
    const float* b;   // [out]
    int out, in;
    float at(int r, int c) const { return (r < out && c >= 0 && c < in) ? w[(size_t)r * in + c] : 0.0f; }
    float bias(int r) const { return r < out ? b[r] : 0.0f; }
};

// group g of a chunk: 64 lanes x 4 consecutive k-steps: lane = (row lane & 31, half h = lane >> 5), group t4 of a unit
// holds k-steps 4 t4 .. 4 t4 + 3 -> col(t, h).

// a row vector as a one-column matrix: at(r, 0) = w[r] (alpha_linear in the backward pass: d h[r] += w_alpha[r] * d sigma)
struct ColumnOf {
    const float* w;
    int rows;
    float at(int r, int c) const { return (r < rows && c == 0) ? w[r] : 0.0f; }
};

// the transpose of the hidden part of a Linear, as the backward-data pass contracts it: d in[r] = sum_c W[c][col0 + r] * d out[c]
struct LinearT {
    const float* w;   // the Linear's [out, ld] row-major weight
    int rows, k, ld, col0;
    float at(int r, int c) const { return (r < rows && c >= 0 && c < k) ? w[(size_t)c * ld + col0 + r] : 0.0f; }
};

template <class Lin, class ColFn>
void fill_group(float* chunk, int g, const Lin& L, int ot, int t4, ColFn col) {
    float* dst = chunk + (size_t)g * kGroupFloats;
    for (int lane = 0; lane < 64; ++lane) {
        const int row = 32 * ot + (lane & 31), h = lane >> 5;
        for (int j = 0; j < 4; ++j) dst[lane * 4 + j] = L.at(row, col(4 * t4 + j, h));
    }
}

// one k-tile x `n_ot` (<= 8) output tiles: group = ot*4 + t4
template <class Lin, class ColFn>
void chunk_ktile(Stream& s, const Lin& L, int n_ot, ColFn col) {
    float* c = s.new_chunk();
    for (int ot = 0; ot < n_ot; ++ot)
        for (int t4 = 0; t4 < 4; ++t4) fill_group(c, ot * 4 + t4, L, ot, t4, col);
}

// feature held by accumulator register r of 32-row tile `tile` in half-wave h
inline int acc_feature(int tile, int r, int h) { return hidden_col(tile, r, h); }

void bias_tiles(std::vector<float>& out, const Linear& L, int n_ot) {
    for (int ot = 0; ot < n_ot; ++ot)
        for (int h = 0; h < 2; ++h)
            for (int r = 0; r < 16; ++r) out.push_back(L.bias(acc_feature(ot, r, h)));
}

// row `row` of a weight matrix over its first 32*n_kt input columns, in accumulator-register order
void row_tiles(std::vector<float>& out, const Linear& L, int row, int n_kt) {
    for (int kt = 0; kt < n_kt; ++kt)
        for (int h = 0; h < 2; ++h)
            for (int r = 0; r < 16; ++r) out.push_back(L.at(row, acc_feature(kt, r, h)));
}

// input column of hidden k-tile kt at (t, h)
inline int hid_col(int kt, int t, int h) { return hidden_col(kt, t, h); }

}  // namespace

// Mirrors the chunk order pack_weights emits below: trunk layer i (hidden chunks, then its encoding chunks when it
// reads cat[input_pts, h]), then feature_linear (id D), the alpha_linear tile (id D+2) and views_linears.0 (id D+1), or output_linear (id D).
std::vector<int> chunk_layers(const nerf_arch& a, uint32_t mask) {
    std::vector<int> ids;
    for (int i = 0; i < a.D; ++i) {
        const bool pe_in = (i == 0) || (mask >> i & 1);
        ids.insert(ids.end(), (pe_in ? 2 : 0) + (i > 0 ? 8 : 0), i);
    }
    if (a.use_viewdirs) {
        ids.insert(ids.end(), 8, a.D);
        ids.insert(ids.end(), 1, a.D + 2);   // the alpha_linear tile
        ids.insert(ids.end(), 5, a.D + 1);
    } else {
        ids.insert(ids.end(), 1, a.D);
    }
    return ids;
}

// The weight stream of the fused backward-data kernel (mlp_kernel.hip, nerf_mlp_bwd_kernel; view-dependent networks):
// the same fragment format as the forward stream, every chunk one k-tile against 8 output tiles, for the chain
//   d feature = W_views[:, :W]^T d(view pre-activation)     4 chunks  (k = the W/2 view units)
//   d h_{D-1} = W_feature^T d feature                        8 chunks
//             + w_alpha d sigma                              1 chunk: the alpha row as the column k = 0 of a k-tile, for the
//                                                            fp16-pair kernel (d sigma rides as one more operand value);
//                                                            the fp32 kernel adds the rank-1 term itself and passes over it
//   (without view directions: d h_{D-1} = W_output^T d raw  1 chunk instead of those thirteen)
//   d h_{i-1} = W_i[:, hidden columns]^T d z_i, i = D-1..1   8 chunks each
// `tensors` in state_dict order as for pack_weights; validated there.
int pack_backward_stream(const nerf_arch& a, const float* const* tensors, uint32_t mask, float** stream_out,
                         int* n_chunks) {
    if (a.W != kWidth || (!a.use_viewdirs && a.output_ch > kBwdMaxOutRows)) {
        set_error("pack_backward_stream: networks of width %d (without view directions: at most %d output channels) only", kWidth,
                  kBwdMaxOutRows);
        return NERF_E_INVALID;
    }
    Stream st;
    auto layer = [&](const LinearT& T, int n_kt) {
        for (int kt = 0; kt < n_kt; ++kt) chunk_ktile(st, T, 8, [kt](int t, int h) { return hid_col(kt, t, h); });
    };
    const float* const* head = tensors + 2 * a.D + 2;
    if (a.use_viewdirs) {
        layer(LinearT{tensors[2 * a.D], a.W, a.W / 2, a.W + a.input_ch_views, 0}, 4);
        layer(LinearT{head[0], a.W, a.W, a.W, 0}, 8);
        chunk_ktile(st, ColumnOf{head[2], a.W}, 8, [](int t, int h) { return hid_col(0, t, h); });
    } else {
        // without view directions the chain starts at d h_{D-1} = W_output^T d raw (nerf.py:109): the head's C <= 8 rows as the
        // columns 0..C-1 of one k-tile for the fp16-pair kernel (d raw is its operand: channels 0-3 in half-wave 0, 4-7 in
        // half-wave 1 - the places hid_col gives columns 0-7); the fp32 kernel forms the product from the rows in the bias block
        // and passes over the chunk
        chunk_ktile(st, LinearT{head[0], a.W, a.output_ch, a.W, 0}, 8, [](int t, int h) { return hid_col(0, t, h); });
    }
    for (int i = a.D - 1; i >= 1; --i) {
        const bool pe_in = (mask >> i) & 1;
        layer(LinearT{tensors[2 * i], a.W, a.W, pe_in ? a.W + a.input_ch : a.W, pe_in ? a.input_ch : 0}, 8);
    }
    float* out = (float*)malloc(st.data.size() * sizeof(float));
    if (!out) {
        set_error("out of host memory packing weights");
        return NERF_E_NOMEM;
    }
    memcpy(out, st.data.data(), st.data.size() * sizeof(float));
    *stream_out = out;
    *n_chunks = (int)(st.data.size() / kChunkFloats);
    return NERF_OK;
}

int pack_weights(const nerf_arch& a, const float* const* tensors, int n_tensors, float** stream_out,
                 int* n_chunks, float** bias_out, int* n_bias_tiles, uint32_t* skip_in_mask,
                 int* out_ch) {
    // The kernels' register tiling is kWidth = 256 wide (8 accumulator tiles per layer, 4 for the view layer). A narrower
    // network (nerf/nerf.py:9: any W) is packed into it with ZERO rows and columns for the units it does not have: their
    // pre-activations are exactly 0, relu(0) = 0 feeds exact zeros on, and x + 0 * y = x in IEEE arithmetic - the same
    // function bit for bit, at the 256-wide network's cost. Every column map below is bounded by the real width.
    if (a.W < 2 || a.W > kWidth) {
        set_error("unsupported netwidth W=%d: 2..%d (narrower networks run zero-padded to %d)", a.W, kWidth, kWidth);
        return NERF_E_INVALID;
    }
    if (a.D < 1 || a.D > kMaxDepth) {
        set_error("unsupported netdepth D=%d (1..%d)", a.D, kMaxDepth);
        return NERF_E_INVALID;
    }
    if (a.input_ch < 3 || a.input_ch > 63 || (a.input_ch - 3) % 6 != 0) {
        set_error("unsupported input_ch=%d (3 + 6*multires, multires <= 10)", a.input_ch);
        return NERF_E_INVALID;
    }
    // without viewdirs views_linears exists but is never evaluated (create_nerf passes input_ch_views = 0)
    if (a.use_viewdirs && (a.input_ch_views < 3 || a.input_ch_views > 27 || (a.input_ch_views - 3) % 6 != 0)) {
        set_error("unsupported input_ch_views=%d (3 + 6*multires_views, multires_views <= 4)", a.input_ch_views);
        return NERF_E_INVALID;
    }
    if (!a.use_viewdirs && (a.input_ch_views < 0 || a.input_ch_views > 64)) {
        set_error("unsupported input_ch_views=%d", a.input_ch_views);
        return NERF_E_INVALID;
    }
    if (a.n_skips < 0 || a.n_skips > NERF_MAX_SKIPS) {
        set_error("n_skips=%d out of range", a.n_skips);
        return NERF_E_INVALID;
    }
    if (!a.use_viewdirs && (a.output_ch < 1 || a.output_ch > 32)) {
        set_error("unsupported output_ch=%d (1..32)", a.output_ch);
        return NERF_E_INVALID;
    }
    const int want = nerf_num_weight_tensors(&a);
    if (n_tensors != want) {
        set_error("expected %d state_dict tensors for this architecture, got %d", want, n_tensors);
        return NERF_E_INVALID;
    }
    for (int i = 0; i < n_tensors; ++i)
        if (!tensors[i]) {
            set_error("state_dict tensor %d is NULL", i);
            return NERF_E_INVALID;
        }

    uint32_t mask = 0;   // bit i: layer i's input is cat[input_pts, h] (nerf/nerf.py:79-80)
    for (int k = 0; k < a.n_skips; ++k) {
        const int s = a.skips[k];
        if (s < 0 || s >= a.D) continue;   // `i in self.skips` never true: ignored by the reference too
        if (s == a.D - 1) {
            // the reference would feed a (W+input_ch)-wide h to its W-wide heads and raise
            set_error("skip at the last trunk layer (%d) is not a valid reference configuration", s);
            return NERF_E_INVALID;
        }
        mask |= 1u << (s + 1);
    }

    Stream st;
    std::vector<float> bias;
    // hidden unit f(kt, t, h) as a column of a Linear whose hidden inputs start at `off`: units >= n_units do not exist
    // (zero padding; in the view layer they would alias the gamma(dir) columns that follow the feature vector)
    auto hid = [](int kt, int off, int n_units) {
        return [kt, off, n_units](int t, int h) {
            const int c = hid_col(kt, t, h);
            return c < n_units ? off + c : -1;
        };
    };
    // gamma(xyz) columns >= input_ch (multires < 10) do not exist; in a skip layer they would
    // alias the hidden columns that follow input_pts, so bound them here.
    auto xyz_col = [&](int tile) {
        const int nx = a.input_ch;
        return [tile, nx](int t, int h) {
            const int c = pe_col_xyz(16 * tile + t, h);
            return (c >= 0 && c < nx) ? c : -1;
        };
    };

    for (int i = 0; i < a.D; ++i) {
        const bool pe_in = (i == 0) || (mask >> i & 1);
        const int in = (i == 0) ? a.input_ch : (pe_in ? a.W + a.input_ch : a.W);
        Linear L{tensors[2 * i], tensors[2 * i + 1], a.W, in};
        bias_tiles(bias, L, 8);
        if (i > 0) {
            const int off = pe_in ? a.input_ch : 0;
            for (int kt = 0; kt < 8; ++kt) chunk_ktile(st, L, 8, hid(kt, off, a.W));
        }
        // the encoding chunks of a skip layer FOLLOW its hidden chunks: the fp16-pair kernel converts the previous
        // layer's outputs tile by tile while the hidden chunks run and has nothing left to hide behind these two
        if (pe_in) {
            chunk_ktile(st, L, 8, xyz_col(0));
            chunk_ktile(st, L, 8, xyz_col(1));
        }
    }
    // views_linears.0 is tensors[2D], [2D+1] in both variants (nerf/nerf.py:43 builds it always)
    const float* const* head = tensors + 2 * a.D + 2;
    if (a.use_viewdirs) {
        Linear views{tensors[2 * a.D], tensors[2 * a.D + 1], a.W / 2, a.W + a.input_ch_views};
        Linear feature{head[0], head[1], a.W, a.W};
        Linear alpha{head[2], head[3], 1, a.W};
        Linear rgb{head[4], head[5], 3, a.W / 2};
        // alpha_linear (nerf.py:86) and rgb_linear (nerf.py:101) have 1 and 3 output rows: as MFMA tiles they
        // would be 97 % / 91 % padding (128 + 64 MFMAs per 32 points), so the kernel evaluates them as
        // per-lane dot products over the activation registers it already holds. Their weights travel in the
        // bias block, arranged per accumulator register exactly like a bias (row_tiles below).
        bias_tiles(bias, alpha, 1);
        // feature_linear (nerf.py:89): a trunk-shaped layer without ReLU
        bias_tiles(bias, feature, 8);
        for (int kt = 0; kt < 8; ++kt) chunk_ktile(st, feature, 8, hid(kt, 0, a.W));
        // alpha_linear once more as a one-row MFMA tile over the 8 k-tiles (group = kt*4 + t4), for the fp16-pair
        // kernel, whose vector pipe is busy converting activations; the fp32 kernel passes over this chunk
        {
            float* c = st.new_chunk();
            for (int kt = 0; kt < 8; ++kt)
                for (int t4 = 0; t4 < 4; ++t4)
                    fill_group(c, kt * 4 + t4, alpha, 0, t4, hid(kt, 0, a.W));
        }
        // views_linears.0 (nerf.py:93-98): input cat[feature(W), gamma(dir)], 4 output tiles.
        // two feature k-tiles per chunk: group = (ktl*4 + ot)*4 + t4
        bias_tiles(bias, views, 4);
        for (int kp = 0; kp < 4; ++kp) {
            float* c = st.new_chunk();
            for (int ktl = 0; ktl < 2; ++ktl)
                for (int ot = 0; ot < 4; ++ot)
                    for (int t4 = 0; t4 < 4; ++t4) {
                        const int kt = 2 * kp + ktl;
                        fill_group(c, (ktl * 4 + ot) * 4 + t4, views, ot, t4, hid(kt, 0, a.W));
                    }
        }
        {
            const int off = a.W;
            const int nv = a.input_ch_views;
            chunk_ktile(st, views, 4, [off, nv](int t, int h) {
                const int c = pe_col_dir(t, h);
                return (c >= 0 && c < nv) ? off + c : -1;
            });
        }
        bias_tiles(bias, rgb, 1);
        row_tiles(bias, alpha, 0, 8);                       // tiles 8D+14 .. 8D+21
        for (int c = 0; c < 3; ++c) row_tiles(bias, rgb, c, 4);   // tiles 8D+22+4c ..
        *out_ch = 4;   // cat[rgb, alpha] (nerf.py:106)
    } else {
        Linear outl{head[0], head[1], a.output_ch, a.W};
        float* c = st.new_chunk();
        for (int kt = 0; kt < 8; ++kt)
            for (int t4 = 0; t4 < 4; ++t4)
                fill_group(c, kt * 4 + t4, outl, 0, t4, hid(kt, 0, a.W));
        bias_tiles(bias, outl, 1);
        // output_linear's rows once more per accumulator register (tiles 8D+1+8c .. +8 for row c): the backward-data pass of
        // the training step turns d raw into d h_{D-1} with them (nerf_mlp_bwd_kernel); only for the channel counts the
        // reference builds (4, or 5 with N_importance > 0, nerf.ipynb:885) - a 32-channel head would not fit the bias block
        // and only when they fit it: a deep trunk with a wide head (D = 12, 8 channels: 161 tiles) loads without them for
        // inference and trains on the layer-by-layer chain (nerf_load_weights leaves the backward stream out)
        if (a.output_ch <= kBwdMaxOutRows &&
            (bias.size() + (size_t)8 * a.output_ch * kBiasTileFloats) * sizeof(float) <= (size_t)kBiasLdsBytes)
            for (int c = 0; c < a.output_ch; ++c) row_tiles(bias, outl, c, 8);
        *out_ch = a.output_ch;
    }
    const size_t ns = st.data.size(), nb = bias.size();
    if (nb * sizeof(float) > (size_t)kBiasLdsBytes) {
        set_error("bias block of %zu bytes exceeds the %d-byte LDS reservation", nb * sizeof(float), kBiasLdsBytes);
        return NERF_E_INVALID;
    }
    float* s_out = (float*)malloc(ns * sizeof(float));
    float* b_out = (float*)malloc(nb * sizeof(float));
    if (!s_out || !b_out) {
        free(s_out);
        free(b_out);
        set_error("out of host memory packing weights");
        return NERF_E_NOMEM;
    }
    memcpy(s_out, st.data.data(), ns * sizeof(float));
    memcpy(b_out, bias.data(), nb * sizeof(float));
    *stream_out = s_out;
    *bias_out = b_out;
    *n_chunks = (int)(ns / kChunkFloats);
    *n_bias_tiles = (int)(nb / kBiasTileFloats);
    *skip_in_mask = mask;
    return NERF_OK;
}

}  // namespace nerf
