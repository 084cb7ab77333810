/*
 * nerf_mi355x.h - C ABI of the MI355X-native NeRF ray-chunk renderer.
 *
 * The reference (isaacchunn/nerf-projects, nerf/) has no FFI or plugin layer on this
 * path: its "operator API" is the set of Python signatures in notebook cells 8-12/15 of
 * nerf/nerf.ipynb plus nerf/nerf.py, nerf/embedder.py and nerf/nerf_helpers.py
 * (SURVEY.md section 8b). Each entry point below states the reference callable it
 * replaces; nerf-projects_amd/host.py binds them with ctypes under the reference's own
 * names and signatures, and INTEGRATION.md shows the stub a maintainer would add.
 *
 * Conventions
 *   - Every function returns 0 on success, a negative NERF_E_* code on failure, or a positive
 *     NERF_W_* warning (work done, outputs valid); nerf_last_error() returns a thread-local
 *     message for the last failure or warning.
 *   - Pointers marked [dev] are device (HBM) addresses on the context's GPU; [host]
 *     are ordinary host addresses. All arrays are dense row-major fp32.
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream). Calls
 *     enqueue work and return; they do not synchronise. Outputs are complete when the
 *     stream reaches the point after the call (the reference is synchronous only
 *     because eager PyTorch on one stream is).
 *   - One context per GPU. A context owns the packed weights and ONE scratch workspace (grown on
 *     demand) that nerf_render_rays / nerf_render_frame / nerf_render_shard / nerf_train_step /
 *     nerf_image_metrics all reuse. Those calls may come from several streams or host threads:
 *     the library orders them itself (a mutex around the enqueue; a call on another stream than
 *     the previous one first waits, on the device, for an event recorded after the previous
 *     call), so they never overlap on the scratch - and therefore never overlap each other; for
 *     concurrent renders on one GPU create one context per stream. Weight loading
 *     (nerf_load_weights, nerf_set_adam_state) is not ordered against rendering on other
 *     streams: finish it (or synchronise) first.
 *   - No CPU fallback exists: without a gfx950 device nerf_ctx_create fails.
 */
#ifndef NERF_MI355X_H
#define NERF_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NERF_OK 0
#define NERF_E_INVALID (-1)     /* bad argument / unsupported configuration */
#define NERF_E_HIP (-2)         /* a HIP runtime call failed                */
#define NERF_E_STATE (-3)       /* e.g. weights of a slot not loaded        */
#define NERF_E_NOMEM (-4)
/* Positive codes are warnings: the call did its work and the outputs are valid (see "Precision guard" below). */
#define NERF_W_PRECISION 1            /* the fp16-pair kernel's scale bound was loose in this call; outputs are its own   */
#define NERF_W_PRECISION_FALLBACK 2   /* ... and the work was redone (frame) / is being done from now on (training) in fp32 */

#define NERF_MAX_SKIPS 8
#define NERF_SLOT_COARSE 0      /* network_fn   (nerf.ipynb:887-889) */
#define NERF_SLOT_FINE 1        /* network_fine (nerf.ipynb:892-896) */
#define NERF_NUM_SLOTS 16

typedef struct nerf_ctx nerf_ctx;

/* Constructor arguments of the reference NeRF module (nerf/nerf.py:9). */
typedef struct nerf_arch {
    int32_t D;                  /* trunk depth, netdepth                          */
    int32_t W;                  /* trunk width, netwidth (this build: 256 only)   */
    int32_t input_ch;           /* 3 + 6*multires (63), or 3 for i_embed == -1    */
    int32_t input_ch_views;     /* 3 + 6*multires_views (27), or 3                */
    int32_t output_ch;          /* 4, or 5 when N_importance > 0 (nerf.ipynb:885) */
    int32_t n_skips;
    int32_t skips[NERF_MAX_SKIPS];
    int32_t use_viewdirs;
} nerf_arch;

/* Library / device ------------------------------------------------------------------ */

const char* nerf_last_error(void);
const char* nerf_version(void);
/* Number of visible HIP devices, or a negative error. */
int nerf_device_count(void);

/* Create a context on HIP device `device` (must be gfx950). */
int nerf_ctx_create(int device, nerf_ctx** out);
void nerf_ctx_destroy(nerf_ctx* ctx);

/* Arithmetic of the fused encode+MLP kernel (NeRF.forward, nerf/nerf.py:57-111). Inputs, outputs, biases,
 * activations between layers and everything outside the MLP are fp32 in both modes (the reference sets
 * torch.float32, nerf.ipynb:76); the modes differ in how the 256-wide contractions are evaluated:
 *   NERF_PRECISION_F32    v_mfma_f32_32x32x2_f32: an fp32 fmaf chain (24-bit operands).
 *   NERF_PRECISION_F16X2  every operand v carried as two fp16 numbers, hi = rn16(v), lo = rn16(v - hi), scaled per
 *                         layer (weights) and per point (activations) by powers of two so that nothing leaves the
 *                         fp16 range; three v_mfma_f32_32x32x16_f16 products per term (W_lo x_hi + W_hi x_lo +
 *                         W_hi x_hi, each exact in the fp32 accumulator), fp32 accumulation. What is proven:
 *                         |v - hi - lo| <= 2^-23 |v| (the remainder v - hi has up to 12 significant bits, lo keeps
 *                         11: up to one fp32 ulp is lost, none when the remainder fits) and the dropped W_lo x_lo is
 *                         <= 2^-22 |W x| per product. That is a per-product bound about 4x the fp32 rounding unit;
 *                         what is MEASURED is that through the whole network the error against an fp64 evaluation
 *                         is equal or smaller than the fp32 MFMA chain's (rms <= 1.25x, max <= 2x, asserted by
 *                         tests/test_hip_parity.py::test_mlp_precisions_vs_fp64 incl. adversarial scalings), because
 *                         these errors are unbiased and far below the accumulated rounding of a 256-term fp32 sum.
 *                         Scale groups: a weight more than 2^12 below the largest of its LAYER, or an activation
 *                         more than 2^12 below the largest of its POINT, has a low half in the fp16 subnormal
 *                         range (absolute resolution 2^-24 of the scaled unit) and keeps fewer than 24 bits. So
 *                         that hidden units of very different size do not meet in one group, the kernel evaluates
 *                         a ROW-EQUALISED copy of the network, made at load time and after optimiser steps: unit j
 *                         is scaled by 2^e_j to the median row norm (weights and bias) of its layer and column j of every layer that
 *                         reads it by 2^-e_j - the same function exactly (ReLU commutes with positive factors, the
 *                         factors are powers of two); nerf_get_weights returns the plain parameters. With it one
 *                         row of a layer 2^20 larger than the others costs nothing, whether its output is used or
 *                         not (tests/test_hip_parity.py::test_fp16_pair_rows_of_unequal_size). What remains is
 *                         WITHIN a row (a weight 2^12 below the layer's largest matters only when the inputs of
 *                         the large ones vanish) and within a point (units equal in norm but 2^12 apart on that
 *                         point); nerf_precision_status counts the points whose scale bound was that loose.
 * A context starts in NERF_PRECISION_F16X2 (about 3x the frame rate of the fp32 chain).
 * Takes effect for the following calls on this context; weights loaded earlier stay valid. */
#define NERF_PRECISION_F32 0
#define NERF_PRECISION_F16X2 1
int nerf_set_precision(nerf_ctx* ctx, int precision);
int nerf_get_precision(nerf_ctx* ctx);
/* The same for the RENDERING entry points only (nerf_render_rays / _frame / _shard and the stage calls): the training
 * step keeps the arithmetic nerf_set_precision chose and any fp32 fallback it is in. This is what a caller that re-renders a
 * range in fp32 after a precision warning uses (the Python mirror's batchify_rays): a validation render inside a training
 * loop must not put the loop back on the fp16-pair kernels. nerf_get_precision returns the rendering arithmetic. */
int nerf_set_render_precision(nerf_ctx* ctx, int precision);
/* NERF_PRECISION_F16X2 chooses a layer's per-point output scale from an a-priori bound (largest row sum of |W| x
 * largest |input| + largest |bias|). A bound 2^12 or more above a point's real outputs starts to cost low-order
 * bits; each such (wavefront, layer) occurrence is counted, never silent. Synchronises the device; `reset` zeroes the
 * counters. The figure is the sum of two counters: the rendering calls' and the training step's (kept apart so that each
 * side's guard sees only its own events). 0 on every network trained or initialised like a NeRF; non-zero means: compare
 * with NERF_PRECISION_F32 on these weights (rows of large weights that cancel). */
int nerf_precision_status(nerf_ctx* ctx, int64_t* loose_bound_events, int reset);
/* Precision guard: how that counter reaches the caller without being asked for. The reference evaluates the network in
 * fp32 (nerf/nerf.py:57-111 under torch.float32, nerf.ipynb:76), so a loose bound must not pass silently:
 *   nerf_render_frame / nerf_render_shard   args->precision_guard: NERF_GUARD_OFF enqueue and return as before;
 *                        NERF_GUARD_REPORT synchronise the stream at the end and return NERF_W_PRECISION if events were
 *                        counted during the frame; NERF_GUARD_FALLBACK additionally render the range again with the fp32
 *                        kernel and return NERF_W_PRECISION_FALLBACK (the Python mirror's render() does this).
 *   nerf_render_rays     stays asynchronous; it enqueues a 4-byte copy of the counter to a pinned host mirror behind its
 *                        kernels. nerf_precision_peek reads the mirror WITHOUT synchronising (events of completed work that
 *                        have not been reported yet); nerf_precision_check synchronises `stream` first. Both mark what
 *                        they return as reported. The Python mirror peeks on entry of render_rays() and checks at the end
 *                        of batchify_rays(), which re-renders the chunks in fp32 when the check is positive.
 *   nerf_train_step      peeks on entry at ITS OWN counter (a frame rendered in between neither consumes a step's events nor
 *                        adds to them): if events of an earlier step have become visible the context's TRAINING switches
 *                        to the fp32 kernels from this step on (until nerf_set_precision is called again) and the call
 *                        returns NERF_W_PRECISION_FALLBACK once. nerf_precision_peek / _check report rendering's events only. */
#define NERF_GUARD_OFF 0
#define NERF_GUARD_REPORT 1
#define NERF_GUARD_FALLBACK 2
/* counts[0] = the counter above; counts[1..7] = the training step's backward-data kernel (gradients, scaled per point like
 * the activations): its (point, layer) events by the bound's overshoot, 2^12-13, 2^14-15, ..., 2^22-23, >= 2^24. These are
 * reported, not guarded: a ReLU-masked gradient vector is sparse, so its largest entry often lies far below the bound of
 * the product it came from - one event in four on the test networks - without any loss that matters: the error of a
 * layer's gradient stays 2^-22 of |W^T| max|dz| for that point (the norm-wise bound of any fp32 product), entries far below
 * the point's largest are what is coarser, and those are negligible in the sums over points the weight gradients are. The
 * gradient tests hold the fp16-pair path to the fp32 path's bars (tests/test_hip_parity.py: autograd parity, six decades of
 * ray errors, rows 2^20 apart). Synchronises the device. */
int nerf_precision_detail(nerf_ctx* ctx, int64_t* counts /*[host] [8]*/, int reset);
int nerf_precision_peek(nerf_ctx* ctx, int64_t* new_events);
int nerf_precision_check(nerf_ctx* ctx, void* stream, int64_t* new_events);

/* Weights ---------------------------------------------------------------------------
 * Replaces NeRF.__init__ + load_state_dict (nerf/nerf.py:9-55; checkpoint reload at
 * nerf.ipynb:927-935). `tensors` are host pointers to the state_dict entries in this
 * order, each exactly as PyTorch stores it (weight [out,in] row-major, bias [out]):
 *   pts_linears.0.weight, pts_linears.0.bias, ..., pts_linears.{D-1}.weight, .bias,
 *   views_linears.0.weight, views_linears.0.bias,
 *   then if use_viewdirs: feature_linear.{weight,bias}, alpha_linear.{weight,bias},
 *                         rgb_linear.{weight,bias}
 *        else:            output_linear.{weight,bias}
 * The weights are repacked once into the MFMA fragment stream the kernel consumes.
 */
int nerf_load_weights(nerf_ctx* ctx, int slot, const nerf_arch* arch,
                      const float* const* tensors /*[host]*/, int n_tensors);
/* Expected tensor count for an architecture (2*D + 2 + (use_viewdirs ? 6 : 2)). */
int nerf_num_weight_tensors(const nerf_arch* arch);

/* Stage entry points (each backs one reference callable) --------------------------- */

/* Embedder.embed / get_embedder (nerf/embedder.py:72-80, 82-116).
 * x [n,3] -> out [n, 3+6*multires]; multires == 0 is the i_embed == -1 identity. */
int nerf_embed(nerf_ctx* ctx, const float* x /*[dev]*/, int64_t n, int multires,
               float* out /*[dev]*/, void* stream);

/* NeRF.forward (nerf/nerf.py:57-111) on already-encoded rows
 * x [B, input_ch + input_ch_views] -> out [B, out_ch] where out_ch is 4 with viewdirs
 * and arch.output_ch without. */
int nerf_mlp_forward(nerf_ctx* ctx, int slot, const float* x /*[dev]*/, int64_t B,
                     float* out /*[dev]*/, void* stream);

/* run_network (nerf.ipynb:790-855) with embed_fn/embeddirs_fn = get_embedder(multires /
 * multires_views): pts [n_rays*n_samples,3], viewdirs [n_rays,3] (NULL when the model
 * does not use them) -> out [n_rays*n_samples, out_ch]. Encoding, the per-sample
 * broadcast of viewdirs and the MLP are fused; netchunk does not exist (results are
 * independent of it). */
int nerf_run_network(nerf_ctx* ctx, int slot, const float* pts /*[dev]*/,
                     const float* viewdirs /*[dev]*/, int64_t n_rays, int64_t n_samples,
                     float* out /*[dev]*/, void* stream);

/* raw2outputs (nerf.ipynb:254-349). raw [N,S,C] (C >= 4; channels 0-2 rgb logits,
 * 3 sigma), z_vals [N,S], rays_d [N,3], noise [N,S] or NULL (already scaled by
 * raw_noise_std - the caller owns the RNG). Any output pointer may be NULL. */
int nerf_raw2outputs(nerf_ctx* ctx, const float* raw /*[dev]*/, int C,
                     const float* z_vals /*[dev]*/, const float* rays_d /*[dev]*/,
                     const float* noise /*[dev]*/, int white_bkgd, int64_t N, int S,
                     float* rgb_map /*[dev] [N,3]*/, float* disp_map /*[dev] [N]*/,
                     float* acc_map /*[dev] [N]*/, float* weights /*[dev] [N,S]*/,
                     float* depth_map /*[dev] [N]*/, void* stream);

/* sample_pdf (nerf/nerf_helpers.py:372-439). bins [N,M], weights [N,M-1],
 * u [N,n_samples] or NULL for det=True (u = linspace(0,1,n_samples)) -> out [N,n_samples]. */
int nerf_sample_pdf(nerf_ctx* ctx, const float* bins /*[dev]*/, const float* weights /*[dev]*/,
                    const float* u /*[dev]*/, int64_t N, int M, int n_samples,
                    float* out /*[dev]*/, void* stream);

/* Stratified depths of render_rays (nerf.ipynb:418-444): z_vals [N, N_samples] from the near/far columns of
 * the ray record; lindisp samples uniformly in disparity; t_rand [N, N_samples] (or NULL) jitters every
 * sample inside its stratum (perturb > 0). */
int nerf_stratified_z(nerf_ctx* ctx, const float* rays /*[dev]*/, int ray_stride, int64_t N, int N_samples,
                      int lindisp, const float* t_rand /*[dev]*/, float* z_vals /*[dev]*/, void* stream);

/* The resampling stage of render_rays (nerf.ipynb:458-467, 486): z_vals_mid, sample_pdf on
 * weights[...,1:-1], then sort(cat[z_vals, z_samples]) and std(z_samples). u [N,n_samples] or NULL (det).
 * z_samples, z_merged [N, S+n_samples] and z_std [N] may each be NULL. */
int nerf_resample(nerf_ctx* ctx, const float* z_vals /*[dev] [N,S]*/, const float* weights /*[dev] [N,S]*/,
                  const float* u /*[dev]*/, int64_t N, int S, int n_samples, float* z_samples /*[dev]*/,
                  float* z_merged /*[dev]*/, float* z_std /*[dev]*/, void* stream);

/* The ray-chunk renderer ------------------------------------------------------------
 * render_rays (nerf.ipynb:359-492) for one chunk of rays, all stages on the device with
 * no host synchronisation: stratified depths -> encode+MLP (coarse) -> composite ->
 * sample_pdf -> merge/sort -> encode+MLP (fine) -> composite -> z_std.
 */
typedef struct nerf_render_args {
    const float* rays;          /* [dev] [N, ray_stride]: o(3) d(3) near far [viewdir(3)]
                                   exactly as render() packs it (nerf.ipynb:622-629)  */
    int64_t n_rays;
    int32_t ray_stride;         /* 8 or 11 (floats per ray)                            */
    int32_t N_samples;          /* S_c                                                 */
    int32_t N_importance;       /* S_i, 0 disables the fine pass                       */
    int32_t slot_coarse;        /* network_fn                                          */
    int32_t slot_fine;          /* network_fine, or -1 to reuse network_fn (:471)      */
    int32_t lindisp;
    int32_t white_bkgd;
    int32_t perturb;            /* perturb > 0: t_rand must be given; u_rand replaces
                                   the deterministic linspace (det = perturb == 0)     */
    const float* t_rand;        /* [dev] [N,S_c] uniforms for stratified jitter or NULL */
    const float* u_rand;        /* [dev] [N,S_i] uniforms for sample_pdf or NULL        */
    const float* noise0;        /* [dev] [N,S_c] sigma noise (scaled) coarse, or NULL   */
    const float* noise;         /* [dev] [N,S_c+S_i] sigma noise fine pass, or NULL     */
    /* outputs, any may be NULL */
    float* rgb_map;             /* [dev] [N,3] last pass                               */
    float* disp_map;            /* [dev] [N]                                           */
    float* acc_map;             /* [dev] [N]                                           */
    float* raw;                 /* [dev] [N,S_last,out_ch] (retraw)                    */
    float* rgb0;                /* [dev] [N,3] coarse pass (N_importance > 0)          */
    float* disp0;               /* [dev] [N]                                           */
    float* acc0;                /* [dev] [N]                                           */
    float* z_std;               /* [dev] [N]                                           */
    /* optional intermediates for stage-wise parity */
    float* z_vals_coarse;       /* [dev] [N,S_c]                                       */
    float* weights_coarse;      /* [dev] [N,S_c]                                       */
    float* z_samples;           /* [dev] [N,S_i]                                       */
    float* z_vals_fine;         /* [dev] [N,S_c+S_i]                                   */
    float* weights_fine;        /* [dev] [N,S_c+S_i]                                   */
    float* depth_map;           /* [dev] [N] last pass                                 */
    const float* z_vals_fine_in;/* [dev] [N,S_c+S_i] inject fine depths (skips sampling) */
    void* stream;
} nerf_render_args;

int nerf_render_rays(nerf_ctx* ctx, const nerf_render_args* args);

/* Ray generation ---------------------------------------------------
 * get_rays (nerf/nerf_helpers.py:222-296) + the packing done by render() (nerf.ipynb:596-629):
 * viewdir normalisation before the NDC warp and before the c2w_staticcam override, optional
 * ndc_rays (nerf_helpers.py:311-369, called with near = 1.0), near/far columns. Writes the
 * [n_pixels, 8|11] ray record for flat pixel indices [first_pixel, first_pixel + n_pixels) of the
 * H x W image (row-major, index = row*W + col), so a rank can generate only its own shard.
 */
typedef struct nerf_camera {
    int32_t H, W;
    float fx, fy, cx, cy;       /* K[0][0], K[1][1], K[0][2], K[1][2] as fp32 (what torch casts to) */
    float c2w[12];              /* camera-to-world [3,4] row-major                               */
    float c2w_static[12];       /* c2w_staticcam, used when has_static != 0                       */
    int32_t has_static;
    int32_t ndc;
    double ndc_focal;           /* K[0][0] as the Python float that ndc_rays receives             */
    float near, far;            /* columns 6 and 7                                                */
    int32_t use_viewdirs;       /* 11 columns instead of 8                                        */
} nerf_camera;

int nerf_generate_rays(nerf_ctx* ctx, const nerf_camera* cam, int64_t first_pixel, int64_t n_pixels,
                       float* rays /*[dev] [n_pixels, 8|11]*/, void* stream);

/* render(rays=(rays_o, rays_d), ...) - the form the training loop calls (nerf.ipynb:1258) - packs a caller-supplied batch:
 * viewdirs = rays_d / |rays_d| taken before the NDC warp (nerf.ipynb:600-614), optional ndc_rays(H, W, K[0][0], 1., ...)
 * (:616-619), near / far columns (:622-629) -> the [n, 8|11] ray record. Of `cam` only H, W, ndc, ndc_focal, near, far and
 * use_viewdirs are read. rays_o / rays_d are [n, >= 3] with row strides of o_stride / d_stride floats (so that the two
 * halves of a stacked record can be passed in place). One kernel instead of the six tensor operations of the reference. */
int nerf_pack_rays(nerf_ctx* ctx, const nerf_camera* cam, const float* rays_o /*[dev]*/, int o_stride,
                   const float* rays_d /*[dev]*/, int d_stride, int64_t n, float* rays /*[dev] [n, 8|11]*/, void* stream);

/* render() for one camera (nerf.ipynb:558-640) in a single call: ray generation for the flat pixel range
 * [first_pixel, first_pixel + n_pixels), the batchify_rays chunk loop and render_rays per chunk, all enqueued
 * on `stream` with no host synchronisation. Deterministic rendering only (perturb = 0, raw_noise_std = 0:
 * render_kwargs_test); outputs are [n_pixels, ...] in pixel order, optional ones may be NULL. */
typedef struct nerf_frame_args {
    nerf_camera cam;
    int64_t first_pixel, n_pixels;
    int64_t chunk;              /* rays per render_rays call; <= 0 means 32768 (the reference default)   */
    int32_t N_samples, N_importance;
    int32_t slot_coarse, slot_fine;
    int32_t lindisp, white_bkgd;
    float* rgb_map;             /* [dev] [n_pixels,3] */
    float* disp_map;            /* [dev] [n_pixels]   */
    float* acc_map;             /* [dev] [n_pixels]   */
    float* rgb0;                /* [dev] optional, N_importance > 0 */
    float* disp0;
    float* acc0;
    float* z_std;
    void* stream;
    int32_t precision_guard;    /* NERF_GUARD_* (see "Precision guard"); 0 = none: the call only enqueues  */
} nerf_frame_args;

int nerf_render_frame(nerf_ctx* ctx, const nerf_frame_args* args);

/* Multi-GPU frame rendering (the `render_sharded` entry of section 8b's export list).
 * The reference's nerf/ path is single-device; rays are independent and cost the same, so the flat [H*W] pixel index is
 * cut into `world` contiguous shards (the first n_total % world ranks get one pixel more) and every rank - one process
 * and one nerf_ctx per GPU - renders its shard with no data-path collective:
 *   nerf_shard_bounds   the partition rule: rank owns [*first_pixel, *first_pixel + *n_pixels)
 *   nerf_render_shard   nerf_render_frame for that range: `args->first_pixel / n_pixels` are ignored and the range of
 *                       (world, rank) is used; outputs are [n_pixels of the shard, ...]; the range is returned through
 *                       first_pixel / n_pixels (may be NULL). An empty shard (n_total < world) renders nothing.
 * The one exchange per frame - gathering rgb|disp|acc (20 B/ray) to rank 0 - belongs to the host's collective library
 * (RCCL ncclGather / torch.distributed.gather on the same stream; INTEGRATION.md shows both): this library links
 * libamdhip64 only. nerf_render_frame(first_pixel, n_pixels) itself is the C-level shard call for any other partition. */
int nerf_shard_bounds(int64_t n_total, int world, int rank, int64_t* first_pixel, int64_t* n_pixels);
int nerf_render_shard(nerf_ctx* ctx, const nerf_frame_args* args, int world, int rank, int64_t* first_pixel,
                      int64_t* n_pixels);

/* Image metrics -----------------------------------------------------
 * calculate_ssim (nerf/nerf_helpers.py:21-111): separable 11-tap Gaussian (sigma 1.5), zero padded,
 * on [H,W,3] images clamped to [0,max_val]; img2mse (nerf_helpers.py:8). Results are written to
 * device scalars: out[0] = mean SSIM, out[1] = MSE of the clamped images. */
int nerf_image_metrics(nerf_ctx* ctx, const float* img1 /*[dev] [H,W,3]*/, const float* img2 /*[dev]*/,
                       int H, int W, float max_val, float* out /*[dev] [2]*/, void* stream);

/* Training step -----------------------------------------------------
 * One iteration of the reference's training loop body (nerf.ipynb:1258-1282) for a batch of rays:
 *   render(rays, retraw=True, **render_kwargs_train) -> loss = img2mse(rgb, target) [+ img2mse(rgb0,
 *   target) when N_importance > 0] -> loss.backward() -> torch.optim.Adam step,
 * on the master fp32 copy of the weights that nerf_load_weights keeps on the device. Afterwards the
 * inference entry points see the updated weights. The caller owns the RNG (t_rand, u_rand, noise*, as in
 * nerf_render_args), the ray batching and the learning-rate schedule (nerf.ipynb:1278-1282).
 * Arithmetic: the forward pass and the hidden-width weight gradients follow nerf_set_precision (NERF_PRECISION_F16X2:
 * fp16-pair arithmetic, whose error is the fp32 kernels'; NERF_TRAIN_FORWARD=f32 / NERF_TRAIN_DW=f32 in the environment
 * keep fp32); backward-data, the other weight gradients, Adam and the master weights are fp32 always.
 */
typedef struct nerf_train_args {
    const float* rays;          /* [dev] [N, 8|11] as render() packs them                       */
    const float* target;        /* [dev] [N,3] target_s                                         */
    int64_t n_rays;
    int32_t ray_stride;
    int32_t N_samples, N_importance;
    int32_t slot_coarse, slot_fine;     /* a distinct fine network is required when N_importance > 0 */
    int32_t lindisp, white_bkgd, perturb;
    const float* t_rand;        /* [dev] [N,S_c]      */
    const float* u_rand;        /* [dev] [N,S_i]      */
    const float* noise0;        /* [dev] [N,S_c]      */
    const float* noise;         /* [dev] [N,S_c+S_i]  */
    float lr, beta1, beta2, eps;        /* Adam; the reference uses betas (0.9, 0.999), eps 1e-8     */
    int32_t step;               /* 1-based optimizer step count (bias correction)               */
    int32_t apply_update;       /* 0: compute loss and gradients only                           */
    float* loss;                /* [dev] [2]: img_loss of the last pass, img_loss0 of the coarse pass */
    float* rgb_map;             /* [dev] [N,3] optional                                         */
    float* rgb0;                /* [dev] [N,3] optional                                         */
    void* stream;
    const float* z_vals_fine_in;/* [dev] [N,S_c+S_i] optional: the fine pass at THESE depths instead of the resampled ones (the
                                   resampling still runs; parity tests inject the reference's fine depths, as
                                   nerf_render_args::z_vals_fine_in does for rendering: sample_pdf is ill-conditioned where a
                                   bin's mass is tiny, and a depth that moves by 5e-4 turns the top-frequency columns of
                                   gamma(x) - hence layer 0's weight gradient - by a third of a radian)            */
    float* stats;               /* [dev] [5] optional: img_loss, img_loss0, loss = their sum, psnr = mse2psnr(img_loss), psnr0
                                   (nerf.ipynb:1262-1272, nerf_helpers.py:14) - what the loop body prints, without a tensor
                                   operation per number (entries 1 and 4 are 0 when N_importance = 0)            */
} nerf_train_args;

int nerf_train_step(nerf_ctx* ctx, const nerf_train_args* args);
/* Current master weights / last gradients of a slot, copied to host tensors in nerf_load_weights order. */
int nerf_get_weights(nerf_ctx* ctx, int slot, float* const* tensors /*[host]*/, int n_tensors);
int nerf_get_gradients(nerf_ctx* ctx, int slot, float* const* tensors /*[host]*/, int n_tensors);
/* torch.optim.Adam's per-parameter state (exp_avg, exp_avg_sq) of a slot, host tensors in nerf_load_weights order:
 * the 'optimizer_state_dict' of the reference's checkpoints (nerf.ipynb:1290-1299; reloaded at :925-932). */
int nerf_get_adam_state(nerf_ctx* ctx, int slot, float* const* exp_avg /*[host]*/, float* const* exp_avg_sq /*[host]*/,
                        int n_tensors);
int nerf_set_adam_state(nerf_ctx* ctx, int slot, const float* const* exp_avg /*[host]*/,
                        const float* const* exp_avg_sq /*[host]*/, int n_tensors);

/* Measurement hooks ------------------------------------------------------------------
 * Accumulated device time of the dominant kernel (the fused encode+MLP kernel),
 * measured with HIP events recorded on the launch stream around every launch while
 * profiling is enabled. nerf_profile_read synchronises those events. */
int nerf_profile_enable(nerf_ctx* ctx, int on);
int nerf_profile_read(nerf_ctx* ctx, double* mlp_ms, int64_t* mlp_launches,
                      int64_t* mlp_points, int reset);

/* The same for the training step's kernels (HIP events on the step's stream while profiling is enabled), summed by kind:
 * [0] the forward passes' fused launches, [1] the backward-data launches, [2] the hidden-width weight-gradient launches
 * with their slice reductions, [3] the other weight gradients. ms / launches / points are [4] arrays (NULL = not wanted). */
int nerf_profile_read_train(nerf_ctx* ctx, double* ms /*[host] [4]*/, int64_t* launches /*[host] [4]*/,
                            int64_t* points /*[host] [4]*/, int reset);

/* Bytes of device workspace currently held by the context. */
int64_t nerf_workspace_bytes(nerf_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif /* NERF_MI355X_H */
