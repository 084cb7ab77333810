// Internal declarations shared by the HIP translation units of libnerf_mi355x.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

#include "nerf_mi355x.h"

namespace nerf {

// ---- packed weight stream ---------------------------------------------------------------
// The fused encode+MLP kernel consumes weights as a linear stream of 32 KiB chunks, each
// chunk being 32 groups x 64 lanes x 4 floats: exactly the bytes one workgroup copies
// HBM/L2 -> LDS with 32 `global_load_lds_dwordx4` wave-instructions and then reads back
// as `ds_read_b128` MFMA A-fragments (4 consecutive k-steps per lane per read).
constexpr int kChunkFloats = 8192;
constexpr int kChunkBytes = kChunkFloats * 4;
constexpr int kStreamTailChunks = 3;      // the fp16-pair stream is followed by a copy of its first chunks (mlp_pair_common.h)
constexpr int kGroupFloats = 256;            // 64 lanes x 4 k-steps
constexpr int kBiasTileFloats = 32;          // [h(2)][16 accumulator registers]
constexpr int kBiasLdsBytes = 20480;         // up to 160 bias / row-vector tiles
constexpr int kWidth = 256;                  // trunk width this build is specialised for
constexpr int kMaxDepth = 12;
constexpr int kBwdMaxOutRows = 8;           // output_linear rows the fused backward-data kernel takes per register (no viewdirs)
constexpr int kPointsPerWave = 32;
constexpr int kWavesPerGroup = 4;
constexpr int kPointsPerGroup = kPointsPerWave * kWavesPerGroup;

// ---- slot maps of the encoded tiles (kernel input side: mlp_inputs.h; weight side: pack_weights.cpp) ----------
// Column of gamma(xyz) (nerf/embedder.py:28-65: [x y z | sin f0 xyz | cos f0 xyz | ...], 63 wide) held by slot
// s = 16*tile + t of half-wave h, or -1 for padding. Half-wave 0 holds sines, half-wave 1 cosines; slots 0..14 are
// the five frequencies the half-wave evaluates itself (0-4 for h = 0, 5-9 for h = 1), slots 15..29 the five its
// partner evaluates; slots 30, 31 the raw coordinates.
__host__ __device__ constexpr int pe_col_xyz(int s, int h) {
    return s < 15 ? 3 + 6 * (s / 3 + 5 * h) + 3 * h + (s % 3)
         : s < 30 ? 3 + 6 * ((s - 15) / 3 + 5 * (1 - h)) + 3 * h + (s % 3)
         : s == 30 ? (h ? 2 : 0) : (h ? -1 : 1);
}
// Column of gamma(dir) (27 wide) held by slot t of the single direction tile: slots 0..5 own frequencies
// (0-1 / 2-3), 6..11 the partner's, 12, 13 the raw components.
__host__ __device__ constexpr int pe_col_dir(int t, int h) {
    return t < 6 ? 3 + 6 * (t / 3 + 2 * h) + 3 * h + (t % 3)
         : t < 12 ? 3 + 6 * ((t - 6) / 3 + 2 * (1 - h)) + 3 * h + (t % 3)
         : t == 12 ? (h ? 2 : 0) : (t == 13 ? (h ? -1 : 1) : -1);
}

// one nn.Linear inside the flat parameter buffer (state_dict order: weight [out,in] then bias [out])
struct LinearDesc {
    int out = 0, in = 0;
    size_t w_off = 0, b_off = 0;
};

// optimiser state, created on the first training step of a slot
struct TrainState {
    bool ready = false;
    float* d_grad = nullptr;     // n_params
    float* d_m = nullptr;        // Adam exp_avg
    float* d_v = nullptr;        // Adam exp_avg_sq
    float* d_wt = nullptr;       // every weight transposed ([in,out]) at the same w_off: B operand of the forward GEMMs
    int* d_stream_table = nullptr;   // fused-stream element -> index into params (or -1)
    int* d_bias_table = nullptr;
    int* d_bwd_table = nullptr;      // backward-stream element -> index into params (or -1)
    // transposed-weight stream of the fused backward-data kernels, cut from the plain parameters or - when the pass runs in
    // the equalised network's units (TrainUnits) - from PackedNet::d_params_eq
    float* d_stream_bwd = nullptr;
    int n_chunks_bwd = 0;
    bool bwd_dirty = true;           // the master parameters changed since d_stream_bwd (and its fp16-pair twin) were made
    bool bwd_is_eq = false;          // what d_stream_bwd currently holds
    bool bwd_is_pair = false;        // ... and whether its fp16-pair twin and gains were made with it
    // fp16-pair twin of d_stream_bwd (mlp_bwd_kernel_h2.hip): one scale group per transposed matrix, the gains bound a
    // layer's input gradient from its output gradient: [largest row sum of |W^T|, largest |alpha weight| (layer 1 only)]
    uint32_t* d_stream_bwd_h2 = nullptr;
    float* d_descale_bwd = nullptr;
    float* d_gain_bwd = nullptr;
    int* d_chunk_layer_bwd = nullptr;
    float* d_chunk_max_bwd = nullptr;
    bool grads_valid = false;
};

struct PackedNet {
    nerf_arch arch{};
    bool loaded = false;
    float* d_params = nullptr;   // flat master copy of the state dict (fp32)
    size_t n_params = 0;
    std::vector<LinearDesc> linears;   // pts_linears[0..D-1], views, then feature, alpha, rgb | output
    std::vector<int> stream_table, bias_table;
    std::vector<int> bwd_table;        // empty when the fused backward-data kernel does not cover the architecture
    TrainState train;
    float* d_stream = nullptr;   // n_chunks * kChunkFloats
    float* d_bias = nullptr;     // n_bias_tiles * kBiasTileFloats
    // fp16-pair twin of d_stream (mlp_kernel_h2.hip): same chunks, every weight as (hi, lo) halves
    // scaled by its layer's power of two; d_descale[layer] undoes the scale
    uint32_t* d_stream_h2 = nullptr;
    float* d_descale = nullptr;
    int* d_chunk_layer = nullptr;
    float* d_chunk_max = nullptr;
    float* d_gain = nullptr;     // per layer [max row sum of |W|, max |b|]: bounds a layer's outputs from its inputs
    // What the fp16-pair kernel evaluates is a ROW-EQUALISED copy of the parameters (launch_equalise_rows): the same function,
    // every hidden unit scaled by a power of two that brings its weight row (and bias) to the layer's median norm, the factor undone in
    // the columns of the layers that read the unit. The stream, the bias block and the gains of that kernel come from it.
    float* d_params_eq = nullptr;
    float* d_stream_eq = nullptr;   // fp32 stream of the equalised parameters (input of the fp16-pair conversion)
    float* d_bias_h2 = nullptr;     // bias block of the equalised parameters
    bool h2_dirty = false;          // the master parameters changed (optimiser step): refresh before the next fp16-pair launch
    bool f32_dirty = false;         // likewise d_stream / d_bias (the fp32 kernels' inputs)
    // e_j of every hidden unit ([kMaxLinears][256], 0 where rows are not scaled), chosen by every refresh_h2: the training step
    // turns the equalised network's gradients into the plain parameters' with them (GradJob::ex)
    int* d_row_exp = nullptr;
    // row_exponents_layers_kernel (one workgroup per linear): flag k = the launch's epoch once linear k's exponents are written
    unsigned* d_eq_flags = nullptr;
    unsigned eq_epoch = 0;
    int n_chunks = 0;
    int n_bias_tiles = 0;
    uint32_t skip_in_mask = 0;   // bit i: trunk layer i reads [input_pts, h]
    int out_ch = 4;              // channels NeRF.forward returns
};

// nerf_ctx::d_loose: two records of 8 words, the rendering calls' at 0 and the training step's at kLooseTrain (the kernels
// get the record's address). Word 0 of a record is the counter the precision guard watches (nerf_precision_status); words
// 1..7 a histogram of the backward-data kernel's (wavefront, layer) events by how far the a-priori bound overshot: 2^12-13,
// 2^14-15, ..., >= 2^24 (nerf_precision_detail; only the training record's fill)
constexpr int kLooseRecord = 8;
constexpr int kLooseWords = 2 * kLooseRecord;
constexpr int kLooseTrain = kLooseRecord;
constexpr int kLooseBwdGuard = 1000;    // overshoot (in binades) from which a backward event would also count for the guard:
                                        // never (nerf_mi355x.h, nerf_precision_detail, says why)

enum MlpInputMode { kInputEmbedded = 0, kInputPoints = 1, kInputRays = 2 };

// Training forward pass through the fused fp32 kernel: where the activations autograd would keep are written
// (row-major [points, channels], any row stride; nullptr = not kept). See train_api.cpp forward_pass.
struct MlpStore {
    float* h[kMaxDepth];     // post-ReLU output of trunk layer i
    int h_ld[kMaxDepth];
    float* feat;             // feature_linear output (no ReLU), 256 wide
    int feat_ld;
    float* hv;               // views_linears[0] output (post ReLU), 128 wide
    int hv_ld;
    // fp16-pair kernels only (the forward pass writes, the backward pass reads; nullptr = not wanted):
    // ReLU masks, one bit per unit - per point and half-wave h a 16-byte record [4 words] at ((point * 2 + h) * 4): word w
    // covers the tiles 2w, 2w + 1 of that half-wave's 128 features; value 2s of tile T is bit 15 - 8 (T & 1) - s, value
    // 2s + 1 bit 31 - 8 (T & 1) - s (the order the conversion hooks meet them in). mask_hv: the view layer's, words 0, 1.
    unsigned* mask[kMaxDepth];
    unsigned* mask_hv;
    // fp16-pair kernels only: h[i] / feat (forward) and h[i] / feat / hv (backward-data) are BLOCKED by 32 points instead of
    // row-major: [point / 32][feature / 32][(feature / 8) % 4][point % 32][(feature / 4) % 2][feature % 4] - the 16 bytes a
    // lane of those kernels holds per store instruction (its point, four consecutive features) laid out so that the
    // instruction's 64 lanes write one contiguous KiB. A buffer has ceil(P / 32) groups of 32 x width floats; the forward
    // pass's hv stays row-major. Consumers: grad_batch_pair_dma_kernel and grad_batch_kernel<1> (GradJob::blocked).
    int blocked;
    // [kBwdMaxSlots] float bits of running maxima (see MlpBwdLaunch::maxes): the forward pass enters the kept activations'
    // (kBwdMaxKept + i) and the feature vector's (kBwdMaxFeatValue)
    unsigned* maxes;
};

struct MlpLaunch {
    const float* stream;
    const uint32_t* stream_h2;   // NERF_PRECISION_F16X2 only
    const float* descale;
    const float* gain;
    unsigned* loose;             // counter of (wave, layer) events where the a-priori output bound was >= 2^12 x too wide
    unsigned long long* stamps;   // -DNERF_STAMPS builds only: s_memtime samples of one wave (profiles/microbench/stamps.py)
    const float* bias;
    int n_chunks;
    int n_bias_tiles;
    int D;
    uint32_t skip_in_mask;
    int use_viewdirs;
    int out_ch;
    int in_ch;            // encoded xyz width (columns of x in embedded mode)
    int in_ch_views;
    int64_t n_points;
    int64_t samples_per_ray;
    // kInputEmbedded
    const float* x;
    int x_ld;
    // kInputPoints
    const float* pts;
    const float* viewdirs;   // [n_rays,3] or nullptr
    // kInputRays
    const float* rays;
    int ray_ld;
    const float* z_vals;
    float* out;
    int store;               // 1: also write the activations named in `st` (fp32 kernel only)
    MlpStore st;
};

// Fused backward-data pass (nerf_mlp_bwd_kernel): from d raw to the gradient at every pre-activation, one launch.
struct MlpBwdLaunch {
    const float* stream;     // pack_backward_stream
    const uint32_t* stream_h2;   // its fp16-pair twin, scales and gains (mlp_bwd_kernel_h2.hip only)
    const float* descale;
    const float* gain;
    unsigned* loose;
    int n_chunks;
    const float* bias;       // the forward bias block (carries the rgb / alpha rows per accumulator register)
    int n_bias_tiles;
    int D;
    int64_t n_points;
    const float* d_raw;      // [P, C]: d rgb (0..2), d sigma (3)
    int C;
    int d_raw_ld;            // row stride of d_raw in floats (fp16-pair kernel without view directions: 8, zero-padded; else C)
    int use_viewdirs;        // 0: output_linear head (C <= kBwdMaxOutRows rows, nerf.py:109): fp32 kernel only
    MlpStore fwd;            // the activations the forward pass kept (ReLU masks): h[i], hv
    MlpStore out;            // h[i] = d(pre-activation of trunk layer i), feat = d feature, hv = d(view pre-activation)
    // optional [kBwdMaxSlots], zeroed by the caller: the largest |value| of what this pass writes and reads, as float bits
    // (atomicMax on non-negative floats) - slot i: d(pre-activation) of trunk layer i; kBwdMaxFeat: d feature; kBwdMaxViews;
    // kBwdMaxKept + i: the kept output of trunk layer i. The fp16-pair weight-gradient kernel takes its scales from them.
    unsigned* maxes;
};
constexpr int kBwdMaxFeat = kMaxDepth, kBwdMaxViews = kMaxDepth + 1, kBwdMaxKept = 16, kBwdMaxFeatValue = 30, kBwdMaxSlots = 32;
// entered by the fp16-pair forward pass for the gamma columns' weight gradients: the largest |gamma(x)| of the pass (the range
// the kernel measures on its encoded inputs anyway) and 1.0 = the bound of gamma(d) (a unit vector, sines and cosines)
constexpr int kBwdMaxGammaX = 28, kBwdMaxGammaD = 29;
// kBwdMaxViews: d(view pre-activation); kBwdMaxFeatValue: the largest |feature| (entered by the fp16-pair forward pass)

// host-side packer (pack_weights.cpp)
int pack_weights(const nerf_arch& arch, const float* const* tensors, int n_tensors,
                 float** stream_out, int* n_chunks, float** bias_out, int* n_bias_tiles,
                 uint32_t* skip_in_mask, int* out_ch);

// stream of the fused backward-data kernel (view-dependent networks; see pack_weights.cpp)
int pack_backward_stream(const nerf_arch& arch, const float* const* tensors, uint32_t skip_in_mask, float** stream_out,
                         int* n_chunks);

// scale group ("layer") of every chunk of the stream, in stream order
std::vector<int> chunk_layers(const nerf_arch& arch, uint32_t skip_in_mask);

// kernel launchers (mlp_kernel.hip, mlp_kernel_h2.hip, ray_kernels.hip)
hipError_t launch_mlp(const MlpLaunch& a, int mode, hipStream_t s);
hipError_t launch_mlp_h2(const MlpLaunch& a, int mode, hipStream_t s);
hipError_t launch_mlp_bwd(const MlpBwdLaunch& b, hipStream_t s);
hipError_t launch_mlp_bwd_h2(const MlpBwdLaunch& b, hipStream_t s);
// the transposed matrices of the backward chain, in stream order (0: W_views[:, :W]^T, 1: W_feature^T, b >= 2: trunk layer
// D - b + 1's hidden columns): gain[2b] = largest row sum of |W^T|, gain[2b + 1] = largest |alpha weight| for b = 1, else 0
struct BwdGainRefs {
    int n;
    unsigned w_off[kMaxDepth + 2];
    int ld[kMaxDepth + 2], rows[kMaxDepth + 2], col0[kMaxDepth + 2];
    unsigned alpha_off;
};
BwdGainRefs bwd_gain_refs(const nerf_arch& arch, const std::vector<LinearDesc>& linears, uint32_t skip_in_mask);
hipError_t launch_layer_gains_bwd(const float* params, const BwdGainRefs& refs, float* gain, hipStream_t s);
// the Linear whose outputs are re-quantised after layer l (trunk 0..D-1, then feature_linear), for launch_layer_gains
struct GainRefs {
    int n;
    unsigned long long w_off[kMaxDepth + 1], b_off[kMaxDepth + 1];
    int out[kMaxDepth + 1], in[kMaxDepth + 1];
};
GainRefs gain_refs(const nerf_arch& arch, const std::vector<LinearDesc>& linears);
hipError_t launch_layer_gains(const float* params, const GainRefs& refs, float* gain, hipStream_t s);
// Row equalisation of the hidden layers (see PackedNet::d_params_eq). Linears are processed in `order`; linear k has its
// rows scaled if scale_rows[k]; the `n_hid[k]` columns from `hid_col0[k]` on are divided by the row scales of linear
// col_src[k] (-1: none).
constexpr int kMaxLinears = 16;
struct EqualiseRefs {
    int n;
    int order[kMaxLinears];
    int out[kMaxLinears], in[kMaxLinears];
    unsigned w_off[kMaxLinears], b_off[kMaxLinears];
    int scale_rows[kMaxLinears], col_src[kMaxLinears], hid_col0[kMaxLinears], n_hid[kMaxLinears];
};
// row_exp [kMaxLinears][256]: e_j of every linear (0 where rows are not scaled): chosen, written out and applied - for one
// network or two in the same pair of launches (the training step equalises the coarse and the fine network together)
hipError_t launch_equalise_rows(int n, const float* const* params, const EqualiseRefs* refs, float* const* params_eq,
                                int* const* row_exp, hipStream_t s, unsigned* const* flags = nullptr, unsigned epoch = 0);
EqualiseRefs equalise_refs(const nerf_arch& arch, const std::vector<LinearDesc>& linears);
struct PackedNet;
// everything the fp16-pair kernel reads, rebuilt from the master parameters (api.cpp; at load and, lazily, after training steps)
int refresh_h2(PackedNet& net, hipStream_t s);
int refresh_h2_many(PackedNet* const* nets, int n, hipStream_t s);
// the fp32 kernels' stream and bias block, likewise
int refresh_f32(PackedNet& net, hipStream_t s);
hipError_t launch_convert_stream_h2(const float* stream, const int* chunk_layer, int n_chunks, float* chunk_max,
                                    uint32_t* out, float* descale, hipStream_t s);
// Everything the fp16-pair kernels read, rebuilt after an optimiser step in two launches (refresh_kernels.hip): up to four
// streams (two networks x forward / backward-data), their bias blocks and gain tables
struct RefreshStream {
    const float* params;      // the (row-equalised) parameters the stream is cut from
    const int* table;         // stream element -> parameter index, or -1
    float* stream;            // fp32 stream (n_chunks x kChunkFloats)
    float* chunk_max;         // [n_chunks]
    const int* chunk_layer;   // scale group of every chunk
    uint32_t* out_h2;         // fp16-pair stream ((n_chunks + kStreamTailChunks) x kChunkFloats words)
    float* descale;           // per scale group
    int n_chunks;
};
struct RefreshBias {
    const float* params;
    const int* table;
    float* out;
    int n;
};
struct RefreshBatch {
    int n_streams, n_bias, n_gain, n_bgain;
    RefreshStream st[4];
    RefreshBias bias[2];
    const float* gain_params[2];
    GainRefs gain[2];
    float* gain_out[2];
    const float* bgain_params[2];
    BwdGainRefs bgain[2];
    float* bgain_out[2];
    const unsigned* mirror_src;   // nerf_ctx::d_loose -> its host mirror (device-visible address), or nullptr
    unsigned* mirror_dst;
};
hipError_t launch_refresh(const RefreshBatch& b, hipStream_t s);
hipError_t launch_embed(const float* x, int64_t n, int multires, float* out, hipStream_t s);
hipError_t launch_stratified(const float* rays, int ray_ld, int64_t N, int S, int lindisp,
                             const float* t_rand, float* z_vals, hipStream_t s);
hipError_t launch_composite(const float* raw, int C, const float* z, const float* rays_d,
                            int d_ld, const float* noise, int white_bkgd, int64_t N, int S,
                            float* rgb, float* disp, float* acc, float* weights, float* depth,
                            hipStream_t s);
hipError_t launch_sample_pdf(const float* bins, const float* weights, int w_ld, int w_off,
                             const float* z_coarse, const float* u, int64_t N, int M,
                             int n_samples, float* samples, float* z_merged, float* z_std,
                             hipStream_t s);

hipError_t launch_raygen(const nerf_camera& cam, int64_t first, int64_t n, float* rays, hipStream_t s);
hipError_t launch_pack_rays(const nerf_camera& cam, const float* rays_o, int o_ld, const float* rays_d, int d_ld, int64_t n,
                            float* rays, hipStream_t s);
hipError_t launch_image_metrics(const float* a, const float* b, int H, int W, float max_val, float* tmp,
                                double* partial, float* out, hipStream_t s);

// ---- training (train_kernels.hip) -------------------------------------------------------------------
struct GemmRows {      // C[M,N] = A[M,K] B[K,N]  (+bias) (ReLU) (C *= mask > 0) (C += old C)
    const float* A; int lda;
    const float* B; int ldb;
    float* C; int ldc;
    int64_t M; int N, K;
    const float* bias;
    int relu;
    const float* mask; int ldm;
    int accumulate;
};
struct GradExps {      // see GradJob
    const int* row;
    const int* col;
    int col_lo, col_hi;
    __host__ __device__ int of(int m, int n) const {
        return (row ? row[m] : 0) - ((col && n >= col_lo && n < col_hi) ? col[n] : 0);
    }
    __host__ __device__ int of_row(int m) const { return row ? row[m] : 0; }
};
struct GemmTN {        // part[slice][Mo, No] = sum_p A[p,Mo]^T B[p,No];  dbp[slice][Mo] = sum_p A[p,Mo]
    const float* A; int lda;
    const float* B; int ldb;
    int64_t P; int Mo, No;
    int64_t pts_per_slice;
    float* part;
    float* dbp;
    int narrow_first;   // the columns beyond a multiple of 256 come first (cat[gamma(x), h]) rather than last
    GradExps ex;        // see GradJob
    int b_blocked;      // gemm_tn_small4_kernel only: B (X, 256 wide) is blocked by 32 points (MlpStore::blocked); ldb unused
};
hipError_t launch_gemm_rows(const GemmRows& g, hipStream_t s);
hipError_t launch_gemm_tn(const GemmTN& g, int n_slices, float* dW, int ldw, float* db, int accumulate, hipStream_t s);
int gemm_tn_col_blocks(int Mo, int No);
bool gemm_tn_is_small(int Mo, int No);
bool gemm_tn_is_direct(int Mo);
// Several weight gradients in ONE launch (train_dw_kernel.hip): a job = the columns [n_begin, n_end) of one Linear's
// dW = dY^T X (+ db), Mo a multiple of 128. All jobs of a batch share the point range and its split into slices, so
// the grid is slices x jobs and a layer needs only 256 / jobs slices to fill the chip: its partials (and the pass that
// adds them up, in slice order) shrink by the number of jobs.
struct GradJob {
    const float* A; int lda;      // dY [P, Mo]
    const float* B; int ldb;      // X  [P, >= n_end]
    int Mo, n_begin, n_end;
    float* dW; int ldw;           // dW[m * ldw + n], n in [n_begin, n_end)
    float* db;                    // nullptr: this job leaves the bias gradient to another one
    float* part;                  // [n_slices][Mo][n_end - n_begin]   (set by launch_grad_batch)
    float* dbp;                   // [n_slices][Mo]
    const unsigned* a_max;        // fp16-pair kernel only: float bits of the largest |dY| and |X| (MlpBwdLaunch::maxes)
    const unsigned* b_max;
    // A pass that ran on the row-equalised network (PackedNet::d_params_eq: W'[m][n] = 2^(e_m - e_n) W[m][n]) hands over
    // dY and X in that network's units; by the chain rule dL/dW[m][n] = 2^(e_m - e_n) dL/dW'[m][n] and dL/db[m] =
    // 2^e_m dL/db'[m] - exact powers of two, applied where the slices are added up. row_exp[m], col_exp[n] (n the column
    // index inside the Linear's input, scaled only inside [col_lo, col_hi): the hidden part of a concatenated input);
    // nullptr = 0.
    GradExps ex;
    // bit 0: A (dY) is blocked by 32 points (MlpStore::blocked), bit 1: B (X) is, and its feature 0 is column b_first of the
    // Linear's input (the hidden part of a concatenated input lives in a buffer of its own then)
    int blocked, b_first;
    // (last, so that the positional initialisers of the other jobs leave them zero) grad_batch_pair_dma_kernel only: a ONE-ROW
    // Linear on the same input X - alpha_linear beside feature_linear (nerf.py:86,89) - rides along: y [P] (stride ldy) is its
    // dY, and its weight gradient sum_p y[p] X[p, :] is formed in fp32 from the X values the job has in registers anyway (a
    // kernel of its own read X a second time). y_part [n_slices][n_end - n_begin], y_dbp [n_slices].
    const float* y; int ldy;
    float* y_part;
    float* y_dbp;
};
constexpr int kMaxGradJobs = 12;
// points per slice are a multiple of this: whole 32-point tiles for the staged kernel, whole groups of k-steps (two points each,
// eight in flight; 12 or 16 measured no faster) for the direct one - only the last slice of a pass ends in the predicated tail loop
constexpr int kSlicePointQuantum = 96;
struct GradBatch {
    int n, n_slices, accumulate;
    int64_t P, pts_per_slice;
    GradJob job[kMaxGradJobs];
};
// wide: every job is 256 columns (NT = 4); otherwise at most 64 columns each (NT = 1). scratch: part_floats / dbp_floats available
// pair: the wide jobs on the fp16 matrix pipe (every job needs a_max / b_max)
hipError_t launch_grad_batch(GradBatch& b, bool wide, float* part, size_t part_floats, float* dbp, size_t dbp_floats,
                             hipStream_t s, bool pair = false);
hipError_t launch_grad_batch_narrow_pair(GradBatch& b, float* part, size_t part_floats, float* dbp, size_t dbp_floats, hipStream_t s);
// a rider row (GradJob::y) needs the LDS-prefetch kernel; its sums are added up into (dW_row [n_end - n_begin], db_row)
struct GradRider {
    int job;               // index in the batch
    float* dW; float* db;
    GradExps ex;
};
bool grad_pair_takes_riders();
hipError_t launch_grad_batch_with_rider(GradBatch& b, const GradRider& r, float* part, size_t part_floats, float* dbp,
                                        size_t dbp_floats, hipStream_t s);
hipError_t launch_embed_train(const float* rays, int ray_ld, const float* z, int64_t P, int S, int Lx, int Lv,
                              float* x0, int ld0, float* x1, int ld1, float* vcat, int ldv, int voff, hipStream_t s, int pad = 0);
hipError_t launch_mse(const float* x, const float* t, int64_t n, float* grad, double* part, float* loss, hipStream_t s);
// The fused small launches of the training step (train_kernels.hip): each evaluates the expressions of the stage kernels it
// replaces, through the same device functions (ray_device.h)
hipError_t launch_train_prologue(const float* rays, int ray_ld, int64_t N, int S, int lindisp, const float* t_rand, float* z,
                                 int Lx, int Lv, float* x0, int ld0, float* x1, int ld1, float* vcat, int ldv, int voff,
                                 unsigned* zero, int n_zero, hipStream_t s, int pad = 0);
hipError_t launch_train_mid(const float* raw, int C, const float* z_c, const float* rays_d, int d_ld, const float* noise,
                            int white_bkgd, int64_t N, int S, float* rgb_c, float* w_c, const float* u, int n_samples,
                            float* z_f, hipStream_t s);
struct TrainEpilogue {
    const float* rays_d; int d_ld;
    const float* target;
    int64_t N;
    int white_bkgd;
    // the last pass (the fine one, or the only one)
    const float* raw_l; int C_l; const float* z_l; const float* noise_l; int S_l; float* d_raw_l;
    int dC_l, dC_c;          // row strides of d_raw_l / d_raw_c (0 = C)
    // the coarse pass when a fine one follows it (raw_c = nullptr otherwise): its colours come from the mid launch
    const float* raw_c; int C_c; const float* z_c; const float* noise_c; int S_c; float* d_raw_c; const float* rgb_c;
    float* out_rgb;          // [N,3] the last pass's colours (the caller's buffer, or scratch)
    float* out_rgb0;         // [N,3] optional copy of the coarse colours for the caller
    double* part;            // [2][N] scratch
    unsigned* ticket;        // zeroed before the launch
    float* loss_dev;         // [2] scratch
    float* out_loss;         // [2] optional (nerf_train_args::loss)
    float* out_stats;        // [5] optional (nerf_train_args::stats)
};
hipError_t launch_train_epilogue(const TrainEpilogue& e, hipStream_t s);
hipError_t launch_train_stats(const float* loss, bool two, float* stats, hipStream_t s);
hipError_t launch_composite_bwd(const float* raw, int C, const float* z, const float* rays_d, int d_ld,
                                const float* noise, int white_bkgd, int64_t N, int S, const float* g_rgb,
                                float* d_raw, hipStream_t s, int dC = 0);      // dC: row stride of d_raw (0 = C)
hipError_t launch_adam(float* p, const float* g, float* m, float* v, int64_t n, float lr, float b1, float b2, float eps,
                       int step, hipStream_t s);
hipError_t launch_transpose(const float* src, int rows, int cols, float* dst, hipStream_t s);
hipError_t launch_gather(const float* params, const int* table, int64_t n, float* out, hipStream_t s);

void set_error(const char* fmt, ...);

}  // namespace nerf
