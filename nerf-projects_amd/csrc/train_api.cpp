// Training step behind the C ABI: the body of the reference's training
// iteration, nerf.ipynb:1258-1282 -
//     render(rays=batch_rays, retraw=True, **render_kwargs_train) -> img2mse(rgb, target) [+ img2mse(rgb0, target)]
//     -> loss.backward() -> Adam step
// for one batch of rays, entirely on the device. The caller owns the RNG (t_rand / u_rand / noise),
// the ray batching and the learning-rate schedule.
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "ctx_internal.h"

using namespace nerf;

namespace {

int ensure_train_state(nerf_ctx* c, PackedNet& net) {
    TrainState& t = net.train;
    if (t.ready) return NERF_OK;
    const size_t nb = net.n_params * sizeof(float);
    HIP_TRY(hipMalloc((void**)&t.d_grad, nb));
    HIP_TRY(hipMalloc((void**)&t.d_m, nb));
    HIP_TRY(hipMalloc((void**)&t.d_v, nb));
    HIP_TRY(hipMalloc((void**)&t.d_wt, nb));
    HIP_TRY(hipMemset(t.d_grad, 0, nb));
    HIP_TRY(hipMemset(t.d_m, 0, nb));
    HIP_TRY(hipMemset(t.d_v, 0, nb));
    // (d_stream_table / d_bias_table: uploaded by nerf_load_weights, which needs them for the fp16-pair stream)
    if (!net.bwd_table.empty()) {
        HIP_TRY(hipMalloc((void**)&t.d_bwd_table, net.bwd_table.size() * sizeof(int)));
        HIP_TRY(hipMemcpy(t.d_bwd_table, net.bwd_table.data(), net.bwd_table.size() * sizeof(int), hipMemcpyHostToDevice));
        HIP_TRY(hipMalloc((void**)&t.d_stream_bwd, net.bwd_table.size() * sizeof(float)));
        t.n_chunks_bwd = (int)(net.bwd_table.size() / kChunkFloats);
    }
    t.ready = true;
    return NERF_OK;
}

bool gemm_forward_requested();

// The master parameters changed (first training call on a slot, optimiser step): everything derived from them is stale
// and is rebuilt by whoever needs it next - the fp32 kernels' stream (refresh_f32), the fp16-pair kernel's equalised
// stream (refresh_h2), the backward-data streams (refresh_bwd). Only the layer-by-layer forward GEMMs' W^T copies are
// made here.
int mark_params_changed(PackedNet& net, hipStream_t s, bool stepped) {
    if (gemm_forward_requested() || net.arch.W != kWidth)      // W^T is the B operand of the layer-by-layer forward GEMMs only
        for (const LinearDesc& d : net.linears)
            HIP_TRY(launch_transpose(net.d_params + d.w_off, d.out, d.in, net.train.d_wt + d.w_off, s));
    if (stepped) {
        net.f32_dirty = true;
        net.h2_dirty = true;
    }
    net.train.bwd_dirty = true;
    return NERF_OK;
}

// scale group of every chunk of the backward stream (pack_backward_stream): one per transposed matrix
std::vector<int> bwd_chunk_layers(const nerf_arch& a) {
    std::vector<int> ids;
    if (a.use_viewdirs) {
        ids.assign(4, 0);
        ids.insert(ids.end(), 8, 1);
        ids.insert(ids.end(), 1, 1);       // the alpha column shares feature_linear's scale
    } else {
        ids.assign(1, 1);                  // W_output^T: backward layer 1 of the chain without view directions
    }
    for (int b = 2; b <= a.D; ++b) ids.insert(ids.end(), 8, b);
    return ids;
}

// The transposed-weight stream of the fused backward-data kernels, from the parameters whose units the pass runs in
// (`eq`: the row-equalised copy the fp16-pair forward pass evaluated), and - `pair` - its fp16-pair twin with the gains
// that bound a layer's input gradient from its output gradient.
int refresh_bwd(PackedNet& net, bool eq, bool pair, hipStream_t s) {
    TrainState& t = net.train;
    if (!t.d_stream_bwd) return NERF_OK;
    const bool have_pair = t.d_stream_bwd_h2 && t.d_descale_bwd && t.d_gain_bwd && t.d_chunk_layer_bwd && t.d_chunk_max_bwd;
    if (!t.bwd_dirty && t.bwd_is_eq == eq && (!pair || have_pair)) return NERF_OK;
    HIP_TRY(launch_gather(eq ? net.d_params_eq : net.d_params, t.d_bwd_table, (int64_t)net.bwd_table.size(), t.d_stream_bwd, s));
    t.bwd_is_eq = eq;
    t.bwd_is_pair = pair;
    if (pair) {
        const int nb = t.n_chunks_bwd;
        if (!have_pair) {
            const std::vector<int> ids = bwd_chunk_layers(net.arch);
            if ((int)ids.size() != nb) {
                set_error("internal: %zu scale groups for %d backward chunks", ids.size(), nb);
                return NERF_E_INVALID;
            }
            // all five or none: a step after a failed allocation must find the slot without a pair stream, not half of one
            void* buf[5] = {};
            const size_t bytes[5] = {(size_t)(nb + kStreamTailChunks) * kChunkBytes, (kMaxDepth + 3) * sizeof(float),
                                     2 * (kMaxDepth + 2) * sizeof(float), (size_t)nb * sizeof(int), (size_t)nb * sizeof(float)};
            hipError_t e = hipSuccess;
            for (int i = 0; i < 5 && e == hipSuccess; ++i) e = hipMalloc(&buf[i], bytes[i]);
            if (e == hipSuccess) e = hipMemcpyAsync(buf[3], ids.data(), bytes[3], hipMemcpyHostToDevice, s);
            if (e == hipSuccess) e = hipStreamSynchronize(s);      // (`ids` is a host temporary; once per slot)
            if (e != hipSuccess) {
                for (void* b : buf)
                    if (b) (void)hipFree(b);
                set_error("refresh_bwd: allocating the fp16-pair backward stream failed: %s", hipGetErrorString(e));
                return e == hipErrorOutOfMemory ? NERF_E_NOMEM : NERF_E_HIP;
            }
            for (void** old : {(void**)&t.d_stream_bwd_h2, (void**)&t.d_descale_bwd, (void**)&t.d_gain_bwd,
                               (void**)&t.d_chunk_layer_bwd, (void**)&t.d_chunk_max_bwd})
                if (*old) (void)hipFree(*old);
            t.d_stream_bwd_h2 = (uint32_t*)buf[0];
            t.d_descale_bwd = (float*)buf[1];
            t.d_gain_bwd = (float*)buf[2];
            t.d_chunk_layer_bwd = (int*)buf[3];
            t.d_chunk_max_bwd = (float*)buf[4];
        }
        HIP_TRY(launch_convert_stream_h2(t.d_stream_bwd, t.d_chunk_layer_bwd, nb, t.d_chunk_max_bwd, t.d_stream_bwd_h2,
                                         t.d_descale_bwd, s));
        HIP_TRY(launch_layer_gains_bwd(eq ? net.d_params_eq : net.d_params, bwd_gain_refs(net.arch, net.linears, net.skip_in_mask),
                                       t.d_gain_bwd, s));
    }
    t.bwd_dirty = false;
    return NERF_OK;
}

// After an optimiser step, in four launches instead of eleven per network and direction (refresh_kernels.hip): the
// row-equalised copies (row_exponents + apply), then every stream the fp16-pair kernels read - the forward ones and, where the
// backward-data kernel has run on them, the transposed ones - with their bias blocks and gains, and the precision guard's
// counters to their host mirror. What refresh_h2_many + refresh_bwd(eq, pair) + mirror_loose do stage by stage.
int refresh_after_step(nerf_ctx* c, PackedNet* const* nets, int n, hipStream_t s, bool* mirrored) {
    const float* params[2];
    float* out[2];
    int* rexp[2];
    EqualiseRefs refs[2];
    if (n < 1 || n > 2) return NERF_E_INVALID;
    for (int i = 0; i < n; ++i) {
        params[i] = nets[i]->d_params;
        out[i] = nets[i]->d_params_eq;
        rexp[i] = nets[i]->d_row_exp;
        refs[i] = equalise_refs(nets[i]->arch, nets[i]->linears);
    }
    unsigned* flags[2] = {nullptr, nullptr};
    for (int i = 0; i < n; ++i) {
        PackedNet& net = *nets[i];
        if (!net.d_eq_flags) {
            HIP_TRY(hipMalloc((void**)&net.d_eq_flags, kMaxLinears * 256 * sizeof(unsigned)));
            HIP_TRY(hipMemsetAsync(net.d_eq_flags, 0, kMaxLinears * 256 * sizeof(unsigned), s));
            net.eq_epoch = 0;
        }
        flags[i] = net.d_eq_flags;
    }
    // (one epoch for the launch: both networks' counters move together; 0 is the value of a fresh flag and is skipped)
    unsigned epoch = nets[0]->eq_epoch + 1;
    if (n > 1 && nets[1]->eq_epoch + 1 > epoch) epoch = nets[1]->eq_epoch + 1;
    if ((epoch & 0xffffffu) == 0) epoch += 1;      // (24 bits travel in a mailbox word; 0 is a fresh word's)
    for (int i = 0; i < n; ++i) nets[i]->eq_epoch = epoch;
    HIP_TRY(launch_equalise_rows(n, params, refs, out, rexp, s, flags, epoch));
    RefreshBatch b{};
    for (int i = 0; i < n; ++i) {
        PackedNet& net = *nets[i];
        TrainState& t = net.train;
        b.st[b.n_streams++] = RefreshStream{net.d_params_eq, t.d_stream_table, net.d_stream_eq, net.d_chunk_max, net.d_chunk_layer,
                                            net.d_stream_h2, net.d_descale, net.n_chunks};
        b.bias[b.n_bias++] = RefreshBias{net.d_params_eq, t.d_bias_table, net.d_bias_h2, (int)net.bias_table.size()};
        b.gain_params[b.n_gain] = net.d_params_eq;
        b.gain[b.n_gain] = gain_refs(net.arch, net.linears);
        b.gain_out[b.n_gain++] = net.d_gain;
        // the backward-data stream, when the fp16-pair backward kernel has been running on the equalised transposed weights
        const bool bwd = t.d_stream_bwd && t.d_stream_bwd_h2 && t.d_descale_bwd && t.d_gain_bwd && t.d_chunk_layer_bwd &&
                         t.d_chunk_max_bwd && t.bwd_is_eq && t.bwd_is_pair;
        if (bwd) {
            b.st[b.n_streams++] = RefreshStream{net.d_params_eq, t.d_bwd_table, t.d_stream_bwd, t.d_chunk_max_bwd,
                                                t.d_chunk_layer_bwd, t.d_stream_bwd_h2, t.d_descale_bwd, t.n_chunks_bwd};
            b.bgain_params[b.n_bgain] = net.d_params_eq;
            b.bgain[b.n_bgain] = bwd_gain_refs(net.arch, net.linears, net.skip_in_mask);
            b.bgain_out[b.n_bgain++] = t.d_gain_bwd;
            t.bwd_dirty = false;
        } else {
            t.bwd_dirty = true;      // (rebuilt by refresh_bwd when a backward pass next wants it)
        }
        net.h2_dirty = false;
    }
    if (c->h_loose_dev && (c->precision == NERF_PRECISION_F16X2 || c->train_precision == NERF_PRECISION_F16X2)) {
        b.mirror_src = c->d_loose;
        b.mirror_dst = c->h_loose_dev;
        *mirrored = true;
    }
    HIP_TRY(launch_refresh(b, s));
    return NERF_OK;
}

struct Pass {              // one network evaluated at P = N*S points with everything autograd would keep
    const PackedNet* net;
    int64_t N, P;
    int S;
    std::vector<float*> in;      // input of trunk layer i ([P, in_i], row stride in_ld[i])
    std::vector<int> in_ld;
    std::vector<float*> h;       // output of trunk layer i (post ReLU), row stride h_ld[i]
    std::vector<int> h_ld;
    int precision = NERF_PRECISION_F32;   // the context's arithmetic: F16X2 runs the forward pass on the fp16-pair kernel
    unsigned* loose = nullptr;            // the context's loose-bound counter (nerf_precision_status)
    unsigned* maxes = nullptr;            // [kBwdMaxSlots] largest |values| of the backward pass (MlpBwdLaunch::maxes)
    float *vcat = nullptr, *hv = nullptr, *raw = nullptr, *d_raw = nullptr;
    float *g_a = nullptr, *g_b = nullptr, *g_hv = nullptr;   // gradient scratch
    std::vector<float*> dz;      // fused backward: d(pre-activation) of trunk layer i, [P, W]
    bool fused_backward = false;
    nerf_ctx* ctx = nullptr;     // (for the profiling hooks)
    bool eq = false;             // the pass runs in the units of the row-equalised network (fp16-pair forward): see GradJob
    bool pair_backward = false;  // backward-data on the fp16 pipe (mlp_bwd_kernel_h2.hip); needs `eq`
    unsigned* mask[kMaxDepth] = {};   // ReLU masks of the trunk layers, one bit per unit (MlpStore::mask), and the view layer's
    unsigned* mask_hv = nullptr;
    int vcat_ld = 0, C = 4;
    int dC = 4;              // row stride of d_raw (8, zero-padded, where the fp16-pair backward kernel reads a head of C <= 8 channels)
    // The kept activations and the gradients at the pre-activations BLOCKED by 32 points (MlpStore::blocked) instead of row-major:
    // h[i], feat_blk (the feature vector; vcat then holds gamma(d) alone, voff = 0), dz[i], g_a (d feature), g_hv. Only when every
    // producer and consumer is one of the fp16-pair kernels that know the layout (set_units).
    bool blocked = false;
    float* feat_blk = nullptr;
    int voff = 0;            // column of gamma(d) in vcat
};

size_t pass_floats(const PackedNet& net, int64_t P) {
    const nerf_arch& a = net.arch;
    size_t f = (size_t)P * a.input_ch;
    f += (size_t)a.D * ((size_t)P * a.W + 64);      // Pass::dz (fused backward)
    for (int i = 0; i < a.D; ++i) f += (size_t)P * (a.W + a.input_ch + 3) + 4;
    f += (size_t)P * (a.W + a.input_ch_views + 3) + (size_t)P * (a.W / 2);
    f += (size_t)P * 8 * 2;                     // raw, d_raw (<= 8 channels budgeted... out_ch <= 32 handled below)
    f += (size_t)P * 2 * (net.out_ch > 8 ? net.out_ch : 0);
    f += (size_t)P * a.W * 2 + (size_t)P * (a.W / 2);
    f += (size_t)(a.D + 1) * ((size_t)P * 8 + 64);    // Pass::mask, mask_hv
    f += (size_t)(2 * a.D + 6) * 32 * (size_t)(a.W + 64) + (size_t)P * 72;      // blocked buffers are whole groups of 32 points; gamma(d) apart
    return f + 64 * 32;
}

void carve_pass(Arena& ar, Pass& ps) {
    const PackedNet& net = *ps.net;
    const nerf_arch& a = net.arch;
    const int64_t P = ps.P;
    const size_t Pg = (size_t)((P + 31) / 32 * 32);      // whole groups of 32 points (blocked buffers)
    ps.C = net.out_ch;
    // gamma(x) for layer 0 (blocked passes: rows of 64, zero-padded - what the fp16-pipe gradients of those columns fetch)
    const int e_ld = ps.blocked ? 64 : a.input_ch;
    float* E = ar.take((size_t)P * e_ld);
    ps.in.assign(a.D, nullptr);
    ps.in_ld.assign(a.D, 0);
    ps.h.assign(a.D, nullptr);
    ps.h_ld.assign(a.D, 0);
    ps.in[0] = E;
    ps.in_ld[0] = e_ld;
    for (int i = 0; i < a.D; ++i) {
        const bool next_cat = (i + 1 < a.D) && ((net.skip_in_mask >> (i + 1)) & 1);
        if (ps.blocked) {
            // h_i in a buffer of its own, blocked by 32 points; a layer that reads cat[gamma(x), h_i] (nerf.py:79-80) finds gamma(x)
            // in the narrow row-major buffer layer 0 reads (in[i + 1]: what the gamma(x) columns' weight gradients read)
            ps.h[i] = ar.take(Pg * a.W);
            ps.h_ld[i] = a.W;
            if (next_cat) {      // (the same gamma(x) layer 0 reads: no second copy)
                ps.in[i + 1] = E;
                ps.in_ld[i + 1] = e_ld;
            } else if (i + 1 < a.D) {
                ps.in[i + 1] = ps.h[i];
                ps.in_ld[i + 1] = a.W;
            }
        } else if (next_cat) {
            // [gamma(x) | h_i] (nerf.py:79-80), rows padded IN FRONT so that h_i starts on a 16-byte boundary and the row
            // stride is a multiple of four floats: the fused kernels then write and read h_i with 16-byte accesses (63
            // leading columns and a stride of 319 floats meant four dword stores for each of them)
            const int pad = (4 - a.input_ch % 4) % 4, ld = a.W + a.input_ch + pad;
            float* cat = ar.take((size_t)P * ld + 4);
            ps.in[i + 1] = cat + pad;
            ps.in_ld[i + 1] = ld;
            ps.h[i] = cat + pad + a.input_ch;
            ps.h_ld[i] = ld;
        } else {
            ps.h[i] = ar.take((size_t)P * a.W);
            ps.h_ld[i] = a.W;
            if (i + 1 < a.D) {
                ps.in[i + 1] = ps.h[i];
                ps.in_ld[i + 1] = a.W;
            }
        }
    }
    if (a.use_viewdirs) {
        if (ps.blocked) {
            ps.feat_blk = ar.take(Pg * a.W);                              // the feature vector, blocked
            ps.vcat_ld = 64;                                              // gamma(dir) alone (nerf.py:93 concatenates them), zero-padded rows
            ps.vcat = ar.take((size_t)P * ps.vcat_ld);
            ps.voff = 0;
        } else {
            ps.vcat_ld = (a.W + a.input_ch_views + 3) / 4 * 4;            // row stride a multiple of four floats, as above
            ps.vcat = ar.take((size_t)P * ps.vcat_ld);                    // [feature | gamma(dir)] (nerf.py:93)
            ps.voff = a.W;
        }
        ps.hv = ar.take((size_t)P * (a.W / 2));
    }
    ps.raw = ar.take((size_t)P * ps.C);
    ps.dC = (ps.pair_backward && !a.use_viewdirs) ? 8 : ps.C;
    ps.d_raw = ar.take((size_t)P * ps.dC);
    const size_t rows = ps.blocked ? Pg : (size_t)P;
    ps.g_a = ar.take(rows * a.W);
    ps.g_b = ar.take((size_t)P * a.W);
    ps.g_hv = ar.take(rows * (a.W / 2));
    ps.dz.assign(a.D, nullptr);
    if (ps.fused_backward)
        for (int i = 0; i < a.D; ++i) ps.dz[i] = ar.take(rows * a.W);     // d(pre-activation) of every trunk layer
    if (!ps.maxes) ps.maxes = (unsigned*)ar.take(kBwdMaxSlots);      // (the fused prologue zeroes one block holding both passes')
    if (ps.eq) {      // (the fp16-pair forward kernel always writes them)
        // ReLU masks, one bit per unit: [P][2 half-waves][4 words] per trunk layer, the view layer's in the same record size
        for (int i = 0; i < a.D; ++i) ps.mask[i] = (unsigned*)ar.take((size_t)P * 8);
        ps.mask_hv = (unsigned*)ar.take((size_t)P * 8);
    }
}

// The training forward pass. NERF_TRAIN_GEMM_FORWARD=1 in the environment (or a network the fused kernel's store
// path does not cover) selects the layer-by-layer GEMM chain below; the default is ONE launch of the fused fp32
// encode+MLP kernel (mlp_kernel.hip, STORE variant: weights streamed by LDS-DMA, activations chained in registers) that
// also writes what autograd would keep - every trunk layer's post-ReLU output, the feature vector, the view layer's
// output - into the same buffers the GEMM chain fills, so the backward pass below is unchanged. Same arithmetic class
// (v_mfma_f32_32x32x2_f32, fp32 accumulate), a different order of the 256 products of a sum.
// With the context in NERF_PRECISION_F16X2 (the default) that launch is the fp16-pair kernel's STORE variant
// (mlp_kernel_h2.hip: the contractions at 3 fp16 MFMAs per term, results as close to fp64 as the fp32 chain's), on a stream
// converted from the plain parameters; NERF_TRAIN_FORWARD=f32 keeps the fp32 kernel. The backward pass is fp32 either way.
bool pair_forward_allowed() {
    static const bool on = [] {
        const char* e = getenv("NERF_TRAIN_FORWARD");
        return !(e && (e[0] == 'f' || e[0] == 'F') && e[1] == '3');
    }();
    return on;
}

// The wide weight-gradient jobs follow the context's precision as well: NERF_PRECISION_F16X2 runs them on the fp16 pipe
// (train_dw_kernel.hip, grad_batch_pair_kernel); NERF_TRAIN_DW=f32 keeps the fp32 kernel.
bool pair_dw_allowed() {
    static const bool on = [] {
        const char* e = getenv("NERF_TRAIN_DW");
        return !(e && (e[0] == 'f' || e[0] == 'F') && e[1] == '3');
    }();
    return on;
}

// Backward-data follows too (mlp_bwd_kernel_h2.hip) when the forward pass ran on the fp16-pair kernel; NERF_TRAIN_BWD=f32
// keeps the fp32 kernel (on the same - equalised - transposed weights).
bool pair_bwd_allowed() {
    static const bool on = [] {
        const char* e = getenv("NERF_TRAIN_BWD");
        return !(e && (e[0] == 'f' || e[0] == 'F') && e[1] == '3');
    }();
    return on;
}

// NERF_TRAIN_GLUE=legacy keeps the step's small stages as launches of their own (stratified depths, encodings, compositing,
// resampling, losses, compositing backward: the stage kernels of ray_kernels.hip / train_kernels.hip) - the A/B switch of
// the fused launches, which call the same device functions and are bit-identical (tests: test_train_glue_is_bit_identical)
bool glue_legacy() {
    static const bool on = [] {
        const char* e = getenv("NERF_TRAIN_GLUE");
        return e && (e[0] == 'l' || e[0] == 'L');
    }();
    return on;
}

bool gemm_forward_requested() {
    static const bool on = [] {
        const char* e = getenv("NERF_TRAIN_GEMM_FORWARD");
        return e && *e && *e != '0';
    }();
    return on;
}

// gamma(x) into layer 0's input and into every concat buffer, gamma(dir) into the view concat buffer: one launch for the
// usual single skip connection (the kernel writes a second destination), one more per further skip layer
// The training step's first launch (train_kernels.hip, embed_train_kernel<true>): the coarse pass's depths are made where its
// encodings are, and the step's small accumulators zeroed by the same launch
struct Prologue {
    int lindisp;
    const float* t_rand;
    unsigned* zero;
    int n_zero;
};

hipError_t embed_inputs(Pass& ps, const float* rays, int ray_ld, float* z, int Lx, int Lv, hipStream_t s,
                        const Prologue* pro = nullptr) {
    const PackedNet& net = *ps.net;
    const nerf_arch& a = net.arch;
    int first = -1;
    for (int i = 1; i < a.D && first < 0; ++i)
        if ((net.skip_in_mask >> i) & 1) first = i;
    if (ps.blocked) first = -1;      // (blocked passes: the skip layers' gamma(x) IS layer 0's buffer, carve_pass)
    hipError_t e = pro ? launch_train_prologue(rays, ray_ld, ps.N, ps.S, pro->lindisp, pro->t_rand, z, Lx, Lv, ps.in[0],
                                               ps.in_ld[0], first > 0 ? ps.in[first] : nullptr,
                                               first > 0 ? ps.in_ld[first] : 0, ps.vcat, ps.vcat_ld, ps.voff, pro->zero,
                                               pro->n_zero, s, ps.blocked ? 1 : 0)
                       : launch_embed_train(rays, ray_ld, z, ps.P, ps.S, Lx, Lv, ps.in[0], ps.in_ld[0],
                                            first > 0 ? ps.in[first] : nullptr, first > 0 ? ps.in_ld[first] : 0, ps.vcat,
                                            ps.vcat_ld, ps.voff, s, ps.blocked ? 1 : 0);
    for (int i = first + 1; first > 0 && i < a.D && e == hipSuccess; ++i)
        if ((net.skip_in_mask >> i) & 1)
            e = launch_embed_train(rays, ray_ld, z, ps.P, ps.S, Lx, 0, ps.in[i], ps.in_ld[i], nullptr, 0, nullptr, 0, 0, s);
    return e;
}

int forward_pass_fused(Pass& ps, const float* rays, int ray_ld, float* z, hipStream_t s, const Prologue* pro) {
    const PackedNet& net = *ps.net;
    const nerf_arch& a = net.arch;
    const int Lx = (a.input_ch - 3) / 6, Lv = a.use_viewdirs ? (a.input_ch_views - 3) / 6 : 0;
    // the encodings are still written out: they are the X of dW = dY^T X for layer 0, the skip layer and the view layer
    HIP_TRY(embed_inputs(ps, rays, ray_ld, z, Lx, Lv, s, pro));
    MlpLaunch m{};
    m.stream = net.d_stream;
    m.bias = net.d_bias;
    m.n_chunks = net.n_chunks;
    m.n_bias_tiles = net.n_bias_tiles;
    m.D = a.D;
    m.skip_in_mask = net.skip_in_mask;
    m.use_viewdirs = a.use_viewdirs;
    m.out_ch = net.out_ch;
    m.in_ch = a.input_ch;
    m.in_ch_views = a.input_ch_views;
    m.n_points = ps.P;
    m.samples_per_ray = ps.S;
    m.rays = rays;
    m.ray_ld = ray_ld;
    m.z_vals = z;
    m.out = ps.raw;
    m.store = 1;
    for (int i = 0; i < a.D; ++i) {
        m.st.h[i] = ps.h[i];
        m.st.h_ld[i] = ps.h_ld[i];
    }
    if (a.use_viewdirs) {
        m.st.feat = ps.blocked ? ps.feat_blk : ps.vcat;
        m.st.feat_ld = ps.blocked ? a.W : ps.vcat_ld;
        m.st.hv = ps.hv;
        m.st.hv_ld = a.W / 2;
    }
    m.st.blocked = ps.blocked ? 1 : 0;
    PackedNet& w = const_cast<PackedNet&>(net);      // (the lazily refreshed streams are caches of the parameters)
    if (ps.eq) {
        // the fp16-pair kernel on the row-equalised network - the stream the renderer uses: what it keeps (activations, ReLU
        // masks, maxima) is in that network's units, and so is everything the backward pass derives from it
        if (net.h2_dirty) {
            const int rc = refresh_h2(w, s);
            if (rc != NERF_OK) return rc;
        }
        m.stream_h2 = net.d_stream_h2;
        m.descale = net.d_descale;
        m.gain = net.d_gain;
        m.bias = net.d_bias_h2;
        m.loose = ps.loose;
        m.st.maxes = ps.maxes;
        for (int i = 0; i < a.D; ++i) m.st.mask[i] = ps.mask[i];
        m.st.mask_hv = ps.mask_hv;
        TrainTimer timer(ps.ctx, s, 0, ps.P);
        HIP_TRY(launch_mlp_h2(m, kInputRays, s));
        return NERF_OK;
    }
    if (net.f32_dirty) {
        const int rc = refresh_f32(w, s);
        if (rc != NERF_OK) return rc;
    }
    TrainTimer timer(ps.ctx, s, 0, ps.P);
    HIP_TRY(launch_mlp(m, kInputRays, s));
    return NERF_OK;
}

int forward_pass(Pass& ps, const float* rays, int ray_ld, float* z, hipStream_t s, const Prologue* pro = nullptr) {
    const PackedNet& net = *ps.net;
    const nerf_arch& a = net.arch;
    // (the fused kernels keep 256-wide rows: a narrower network, which they evaluate zero-padded, trains on the chain below)
    if (!gemm_forward_requested() && ps.C == net.out_ch && a.D <= kMaxDepth && a.W == kWidth)
        return forward_pass_fused(ps, rays, ray_ld, z, s, pro);
    const float* wt = net.train.d_wt;
    const float* prm = net.d_params;
    const int Lx = (a.input_ch - 3) / 6, Lv = a.use_viewdirs ? (a.input_ch_views - 3) / 6 : 0;
    // gamma(x) into layer 0's input and into every concat buffer; gamma(dir) into the view concat buffer
    HIP_TRY(embed_inputs(ps, rays, ray_ld, z, Lx, Lv, s, pro));
    for (int i = 0; i < a.D; ++i) {
        const LinearDesc& d = net.linears[i];
        GemmRows g{ps.in[i], ps.in_ld[i], wt + d.w_off, d.out, ps.h[i], ps.h_ld[i], ps.P, d.out, d.in,
                   prm + d.b_off, 1, nullptr, 0, 0};
        HIP_TRY(launch_gemm_rows(g, s));
    }
    const float* hl = ps.h[a.D - 1];
    const int hl_ld = ps.h_ld[a.D - 1];
    if (a.use_viewdirs) {
        const LinearDesc &views = net.linears[a.D], &feat = net.linears[a.D + 1], &alpha = net.linears[a.D + 2],
                         &rgb = net.linears[a.D + 3];
        GemmRows ga{hl, hl_ld, wt + alpha.w_off, 1, ps.raw + 3, ps.C, ps.P, 1, a.W, prm + alpha.b_off, 0, nullptr, 0, 0};
        HIP_TRY(launch_gemm_rows(ga, s));                                                  // nerf.py:86
        GemmRows gf{hl, hl_ld, wt + feat.w_off, a.W, ps.vcat, ps.vcat_ld, ps.P, a.W, a.W, prm + feat.b_off, 0, nullptr, 0, 0};
        HIP_TRY(launch_gemm_rows(gf, s));                                                  // nerf.py:89
        GemmRows gv{ps.vcat, ps.vcat_ld, wt + views.w_off, views.out, ps.hv, views.out, ps.P, views.out, views.in,
                    prm + views.b_off, 1, nullptr, 0, 0};
        HIP_TRY(launch_gemm_rows(gv, s));                                                  // nerf.py:96-98
        GemmRows gr{ps.hv, views.out, wt + rgb.w_off, 3, ps.raw, ps.C, ps.P, 3, rgb.in, prm + rgb.b_off, 0, nullptr, 0, 0};
        HIP_TRY(launch_gemm_rows(gr, s));                                                  // nerf.py:101
    } else {
        const LinearDesc& out = net.linears[a.D + 1];
        GemmRows go{hl, hl_ld, wt + out.w_off, out.out, ps.raw, ps.C, ps.P, out.out, a.W, prm + out.b_off, 0, nullptr, 0, 0};
        HIP_TRY(launch_gemm_rows(go, s));                                                  // nerf.py:109
    }
    return NERF_OK;
}

struct TnScratch {
    float* part;      // [max_slices][256][No <= 320] partial dW
    float* dbp;       // [max_slices][256] partial db (twice that: a batch of layers keeps all its bias partials at once)
    int max_slices;
    int64_t P;        // points of the pass being differentiated
    int accumulate;   // add to the gradients already there (the second pass through a shared network)
    size_t part_floats, dbp_floats;
};

// Slices per GEMM: the grid is slices x column blocks (gemm_tn_col_blocks) workgroups at one per CU, so aim at a whole number of
// 256-workgroup rounds (a 1.5-round grid wastes a third of the machine) while keeping >= 256 points per slice.
inline int pick_slices(int64_t P, int Mo, int No, int max_slices) {
    if (gemm_tn_is_small(Mo, No)) {     // [slices][Mo <= 8][No] partials: 2048 slices fit the buffers sized for 256 x 256 rows
        const int64_t s = (P + 63) / 64;      // (measured with 128 / 256 / 512 points per slice: 48 / 55 / 88 us per launch against 42)
        return (int)(s < 1 ? 1 : (s > 2048 ? 2048 : s));
    }
    const int yb = gemm_tn_col_blocks(Mo, No);
    int s = 256 / yb;
    const int64_t cap = (P + 255) / 256;
    if (s > cap) s = (int)cap;
    if (s > max_slices) s = max_slices;
    return s < 1 ? 1 : s;
}

// How the gradient of Linear k is brought back from the equalised network's units (GradJob::ex); all zero otherwise
GradExps grad_exps(const PackedNet& net, int k, bool eq) {
    GradExps ex{nullptr, nullptr, 0, 0};
    if (!eq) return ex;
    const EqualiseRefs r = equalise_refs(net.arch, net.linears);
    ex.row = net.d_row_exp + 256 * k;                       // zeros where the rows are not scaled (alpha, rgb)
    if (r.col_src[k] >= 0) {
        ex.col = net.d_row_exp + 256 * r.col_src[k] - r.hid_col0[k];
        ex.col_lo = r.hid_col0[k];
        ex.col_hi = r.hid_col0[k] + r.n_hid[k];
    }
    return ex;
}

// dW (+db) of one Linear: dW = dY^T X, db = dY^T 1
int grad_linear(const PackedNet& net, const LinearDesc& d, const float* dY, int ldy, const float* X, int ldx, int64_t P,
                const TnScratch& sc, hipStream_t s, bool eq = false, bool x_blocked = false) {
    const int n_slices = pick_slices(P, d.out, d.in, sc.max_slices);
    int64_t pps = (P + n_slices - 1) / n_slices;
    pps = (pps + kSlicePointQuantum - 1) / kSlicePointQuantum * kSlicePointQuantum;
    // a trunk layer behind a skip reads cat[gamma(x), h] (nerf.py:79-80): its narrow columns come first; the view layer
    // reads cat[feature, gamma(d)] (nerf.py:93): last
    const bool trunk = &d >= &net.linears[0] && &d < &net.linears[0] + net.arch.D;
    GemmTN g{dY, ldy, X, ldx, P, d.out, d.in, pps, sc.part, sc.dbp, (trunk && d.in > net.arch.W) ? 1 : 0,
             grad_exps(net, (int)(&d - &net.linears[0]), eq), x_blocked ? 1 : 0};
    HIP_TRY(launch_gemm_tn(g, n_slices, net.train.d_grad + d.w_off, d.in, net.train.d_grad + d.b_off, sc.accumulate, s));
    return NERF_OK;
}

// The backward pass with the data gradients from ONE launch of nerf_mlp_bwd_kernel (mlp_kernel.hip): d raw -> the
// gradient at every pre-activation, transposed weights streamed, the running gradient chained in registers, masks read
// from the activations the forward pass kept. The weight gradients stay what they were: one gemm_tn per Linear
// (dW = dZ^T X, slices summed in a fixed order), now all issued after that launch. NERF_TRAIN_GEMM_BACKWARD=1 keeps the
// layer-by-layer chain below (also used for networks without view directions).
bool gemm_backward_requested() {
    static const bool on = [] {
        const char* e = getenv("NERF_TRAIN_GEMM_BACKWARD");
        return e && *e && *e != '0';
    }();
    return on;
}

int backward_pass_fused_noviews(Pass& ps, const TnScratch& sc, hipStream_t s);

// NERF_TRAIN_NARROW=f32: the gamma(x) / gamma(d) columns' weight gradients of a blocked pass on the fp32 kernel (A/B)
bool narrow_pair_wanted() {
    static const bool on = [] {
        const char* e = getenv("NERF_TRAIN_NARROW");
        return !(e && (e[0] == 'f' || e[0] == 'F') && e[1] == '3');
    }();
    return on;
}

// The gamma columns' jobs on the fp16 pipe (grad_batch_narrow_pair_kernel). A workgroup is a slice (its four waves share its
// points and add their sums up in LDS): as many as fill the chip's 256 workgroups once over all jobs
int narrow_pair_batch(Pass& ps, const TnScratch& sc, GradBatch& narrow_pair, hipStream_t s) {
    if (narrow_pair.n == 0) return NERF_OK;
    int n_slices = 256 / narrow_pair.n;
    const int64_t cap = (ps.P + 255) / 256;
    if (n_slices > cap) n_slices = (int)cap;
    if (n_slices < 1) n_slices = 1;
    int64_t pps = (ps.P + n_slices - 1) / n_slices;
    pps = (pps + 127) / 128 * 128;
    narrow_pair.n_slices = n_slices;
    narrow_pair.pts_per_slice = pps;
    narrow_pair.P = ps.P;
    narrow_pair.accumulate = sc.accumulate;
    TrainTimer timer(ps.ctx, s, 3, ps.P);
    HIP_TRY(launch_grad_batch_narrow_pair(narrow_pair, sc.part, sc.part_floats, sc.dbp, sc.dbp_floats, s));
    return NERF_OK;
}

int backward_pass_fused(Pass& ps, const TnScratch& sc, hipStream_t s) {
    const PackedNet& net = *ps.net;
    const nerf_arch& a = net.arch;
    if (!a.use_viewdirs) return backward_pass_fused_noviews(ps, sc, s);
    const LinearDesc &views = net.linears[a.D], &feat = net.linears[a.D + 1], &alpha = net.linears[a.D + 2],
                     &rgb = net.linears[a.D + 3];
    float* d_feat = ps.g_a;      // [P, W]
    const bool eq = ps.eq;
    {
        const int rc = refresh_bwd(const_cast<PackedNet&>(net), eq, ps.pair_backward, s);
        if (rc != NERF_OK) return rc;
    }
    MlpBwdLaunch b{};
    b.stream = net.train.d_stream_bwd;
    b.n_chunks = net.train.n_chunks_bwd;
    b.bias = eq ? net.d_bias_h2 : net.d_bias;      // (the rgb / alpha rows of the network the pass runs on)
    b.n_bias_tiles = net.n_bias_tiles;
    b.D = a.D;
    b.n_points = ps.P;
    b.d_raw = ps.d_raw;
    b.C = ps.C;
    b.use_viewdirs = 1;
    for (int i = 0; i < a.D; ++i) {
        b.fwd.h[i] = ps.h[i];
        b.fwd.h_ld[i] = ps.h_ld[i];
        b.out.h[i] = ps.dz[i];
        b.out.h_ld[i] = a.W;
    }
    b.fwd.hv = ps.hv;
    b.fwd.hv_ld = views.out;
    b.out.hv = ps.g_hv;
    b.out.hv_ld = views.out;
    b.out.feat = d_feat;
    b.out.feat_ld = a.W;
    b.out.blocked = ps.blocked ? 1 : 0;
    // (ps.maxes was zeroed before the forward pass, which - on the fp16-pair kernel - has entered the kept activations'
    // maxima and the feature vector's; the backward kernels add the gradients', the fp32 one the kept ones as well)
    const bool pair_dw = ps.precision == NERF_PRECISION_F16X2 && pair_dw_allowed() && ps.maxes != nullptr;
    if (pair_dw || ps.pair_backward) b.maxes = ps.maxes;
    if (ps.pair_backward) {
        b.stream_h2 = net.train.d_stream_bwd_h2;
        b.descale = net.train.d_descale_bwd;
        b.gain = net.train.d_gain_bwd;
        b.loose = ps.loose;
        for (int i = 0; i < a.D; ++i) b.fwd.mask[i] = ps.mask[i];
        b.fwd.mask_hv = ps.mask_hv;
        TrainTimer timer(ps.ctx, s, 1, ps.P);
        HIP_TRY(launch_mlp_bwd_h2(b, s));
    } else {
        TrainTimer timer(ps.ctx, s, 1, ps.P);
        HIP_TRY(launch_mlp_bwd(b, s));
    }
    // the view layer's job can join the fp16-pair batch when the feature vector's size is known: the fp16-pair forward pass
    // has tracked it (kBwdMaxFeatValue)
    const bool feat_known = pair_dw && eq;
    const float* hl = ps.h[a.D - 1];
    const int hl_ld = ps.h_ld[a.D - 1];
    int rc;
    if ((rc = grad_linear(net, rgb, ps.d_raw, ps.C, ps.hv, views.out, ps.P, sc, s, eq))) return rc;
    const bool batched = gemm_tn_is_direct(a.W) && gemm_tn_is_direct(views.out) && a.input_ch <= 64 &&
                         a.input_ch_views <= 64 && a.D + 1 <= kMaxGradJobs;
    // alpha_linear reads what feature_linear reads (h_{D-1}, nerf.py:86,89): its one-row gradient rides in that job of the
    // fp16-pair batch (GradJob::y) where the kernel that takes riders is in use; a launch of its own otherwise
    static const bool rider_wanted = [] {      // NERF_TRAIN_DW_RIDER=0: A/B
        const char* e = getenv("NERF_TRAIN_DW_RIDER");
        return !(e && *e == '0');
    }();
    const bool alpha_rides = rider_wanted && batched && pair_dw && grad_pair_takes_riders() && a.D + 2 <= kMaxGradJobs &&
                             hl_ld % 4 == 0;
    if (ps.blocked && !(alpha_rides && batched && pair_dw && feat_known)) {
        set_error("internal: the blocked activation layout was chosen for a pass whose weight gradients cannot read it");
        return NERF_E_INVALID;
    }
    if (!alpha_rides && (rc = grad_linear(net, alpha, ps.d_raw + 3, ps.C, hl, hl_ld, ps.P, sc, s, eq))) return rc;
    if (!batched) {
        if ((rc = grad_linear(net, views, ps.g_hv, views.out, ps.vcat, ps.vcat_ld, ps.P, sc, s, eq))) return rc;
        if ((rc = grad_linear(net, feat, d_feat, a.W, hl, hl_ld, ps.P, sc, s, eq))) return rc;
        for (int i = a.D - 1; i >= 0; --i)
            if ((rc = grad_linear(net, net.linears[i], ps.dz[i], a.W, ps.in[i], ps.in_ld[i], ps.P, sc, s, eq))) return rc;
        return NERF_OK;
    }
    // Every other weight gradient in two launches (+ their reductions): the 256-column blocks of all layers, then the
    // gamma(x) / gamma(d) columns. With J jobs in a launch a layer is cut into 256 / J slices instead of 256: J times
    // fewer partial sums to write and to add up (a layer's 256 partials were 67 MB, and the pass over them 12 % of the step).
    float* grad = net.train.d_grad;
    // (pairs: the jobs whose operands the backward kernel has measured - d z_i / d feature against the kept h_{i-1}, the
    // view layer's against a bound of the feature vector - go to the fp16-pair kernel when the context's arithmetic is F16X2)
    GradBatch wide{}, narrow{}, pairs{}, narrow_pair{};
    // blocked: 0, or bit 0 = dY blocked by 32 points, bit 1 = X blocked with its feature 0 at column b_first (GradJob)
    auto job = [&](GradBatch& b, const LinearDesc& d, const float* dY, int ldy, const float* X, int ldx, int n0, int n1,
                   bool with_db, const unsigned* a_max = nullptr, const unsigned* b_max = nullptr, int blocked = 0,
                   int b_first = 0) {
        GradJob& j = b.job[b.n++];
        j = GradJob{dY, ldy, X, ldx, d.out, n0, n1, grad + d.w_off, d.in, with_db ? grad + d.b_off : nullptr,
                    nullptr, nullptr, a_max, b_max, grad_exps(net, (int)(&d - &net.linears[0]), eq)};
        j.blocked = blocked;
        j.b_first = b_first;
    };
    GradBatch& hidden = pair_dw ? pairs : wide;
    auto mx = [&](int slot) -> const unsigned* { return pair_dw ? ps.maxes + slot : nullptr; };
    if (ps.blocked) {
        // every hidden-width operand blocked by 32 points, in a buffer of its own; the gamma(x) / gamma(d) columns' jobs read
        // a blocked dY against a narrow row-major X (rows of 64, zero-padded) whose column 0 is column n_begin of the Linear's
        // input: on the fp16 pipe too (grad_batch_narrow_pair_kernel), or - NERF_TRAIN_NARROW=f32 - on the fp32 one
        GradBatch& nb = narrow_pair_wanted() ? narrow_pair : narrow;
        const int np = narrow_pair_wanted() ? 1 : 0;
        job(pairs, feat, d_feat, a.W, ps.h[a.D - 1], a.W, 0, a.W, true, mx(kBwdMaxFeat), mx(kBwdMaxKept + a.D - 1), 3, 0);
        job(pairs, views, ps.g_hv, views.out, ps.feat_blk, a.W, 0, a.W, true, mx(kBwdMaxViews), mx(kBwdMaxFeatValue), 3, 0);
        if (views.in > a.W)
            job(nb, views, ps.g_hv, views.out, np ? ps.vcat : ps.vcat - a.W, ps.vcat_ld, a.W, views.in, false, mx(kBwdMaxViews),
                mx(kBwdMaxGammaD), 1);
        for (int i = a.D - 1; i >= 0; --i) {
            const LinearDesc& d = net.linears[i];
            if (d.in >= a.W) {
                const int lead = d.in - a.W;                                           // cat[gamma(x), h] (nerf.py:79-80)
                job(pairs, d, ps.dz[i], a.W, ps.h[i - 1], a.W, lead, d.in, true, mx(i), mx(kBwdMaxKept + i - 1), 3, lead);
                if (lead > 0) job(nb, d, ps.dz[i], a.W, ps.in[i], ps.in_ld[i], 0, lead, false, mx(i), mx(kBwdMaxGammaX), 1);
            } else {
                job(nb, d, ps.dz[i], a.W, ps.in[i], ps.in_ld[i], 0, d.in, true, mx(i), mx(kBwdMaxGammaX), 1);   // layer 0: gamma(x) only
            }
        }
    } else {
    job(hidden, feat, d_feat, a.W, hl, hl_ld, 0, a.W, true, mx(kBwdMaxFeat), mx(kBwdMaxKept + a.D - 1));
    job(feat_known ? pairs : wide, views, ps.g_hv, views.out, ps.vcat, ps.vcat_ld, 0, a.W, true, mx(kBwdMaxViews),
        mx(kBwdMaxFeatValue));                                                        // cat[feature, gamma(d)] (nerf.py:93)
    if (views.in > a.W) job(narrow, views, ps.g_hv, views.out, ps.vcat, ps.vcat_ld, a.W, views.in, false);
    for (int i = a.D - 1; i >= 0; --i) {
        const LinearDesc& d = net.linears[i];
        if (d.in >= a.W) {
            const int lead = d.in - a.W;                                               // cat[gamma(x), h] (nerf.py:79-80)
            job(hidden, d, ps.dz[i], a.W, ps.in[i], ps.in_ld[i], lead, d.in, true, mx(i), mx(kBwdMaxKept + i - 1));
            if (lead > 0) job(narrow, d, ps.dz[i], a.W, ps.in[i], ps.in_ld[i], 0, lead, false);
        } else {
            job(narrow, d, ps.dz[i], a.W, ps.in[i], ps.in_ld[i], 0, d.in, true);       // layer 0: gamma(x) only
        }
    }
    }
    for (GradBatch* b : {&pairs, &wide, &narrow}) {
        if (b->n == 0) continue;
        int n_slices = 256 / b->n;
        const int64_t cap = (ps.P + 255) / 256;
        if (n_slices > cap) n_slices = (int)cap;
        if (n_slices < 1) n_slices = 1;
        int64_t pps = (ps.P + n_slices - 1) / n_slices;
        pps = (pps + kSlicePointQuantum - 1) / kSlicePointQuantum * kSlicePointQuantum;
        b->n_slices = n_slices;
        b->pts_per_slice = pps;
        b->P = ps.P;
        b->accumulate = sc.accumulate;
        TrainTimer timer(ps.ctx, s, b == &narrow ? 3 : 2, ps.P);
        if (b == &pairs && alpha_rides) {      // (job 0 of the batch is feature_linear's)
            b->job[0].y = ps.d_raw + 3;
            b->job[0].ldy = ps.C;
            const GradRider rider{0, grad + alpha.w_off, grad + alpha.b_off, grad_exps(net, a.D + 2, eq)};
            HIP_TRY(launch_grad_batch_with_rider(*b, rider, sc.part, sc.part_floats, sc.dbp, sc.dbp_floats, s));
        } else {
            HIP_TRY(launch_grad_batch(*b, b != &narrow, sc.part, sc.part_floats, sc.dbp, sc.dbp_floats, s, b == &pairs));
        }
    }
    return narrow_pair_batch(ps, sc, narrow_pair, s);
}

// Networks without view directions (use_viewdirs=False: output_linear on the trunk, nerf.py:109): the same fused fp32 launch
// from d raw to every pre-activation gradient - the head's transpose as vector products inside the kernel - then the trunk's
// weight gradients batched as above and output_linear's as one small job. views_linears exists in the module
// (nerf/nerf.py:43) and is never evaluated: its gradient is zero.
int backward_pass_fused_noviews(Pass& ps, const TnScratch& sc, hipStream_t s) {
    const PackedNet& net = *ps.net;
    const nerf_arch& a = net.arch;
    const LinearDesc &views = net.linears[a.D], &out = net.linears[a.D + 1];
    const bool eq = ps.eq;
    {
        const int rc = refresh_bwd(const_cast<PackedNet&>(net), eq, ps.pair_backward, s);
        if (rc != NERF_OK) return rc;
    }
    MlpBwdLaunch b{};
    b.stream = net.train.d_stream_bwd;
    b.n_chunks = net.train.n_chunks_bwd;
    b.bias = eq ? net.d_bias_h2 : net.d_bias;
    b.n_bias_tiles = net.n_bias_tiles;
    b.D = a.D;
    b.n_points = ps.P;
    b.d_raw = ps.d_raw;
    b.C = ps.C;
    b.d_raw_ld = ps.dC;
    b.use_viewdirs = 0;
    for (int i = 0; i < a.D; ++i) {
        b.fwd.h[i] = ps.h[i];
        b.fwd.h_ld[i] = ps.h_ld[i];
        b.out.h[i] = ps.dz[i];
        b.out.h_ld[i] = a.W;
    }
    b.out.blocked = ps.blocked ? 1 : 0;
    const bool pair_dw = ps.precision == NERF_PRECISION_F16X2 && pair_dw_allowed() && ps.maxes != nullptr;
    if (pair_dw || ps.pair_backward) b.maxes = ps.maxes;
    if (ps.pair_backward) {
        // the fp16-pair kernel on the equalised transposed weights: W_output^T is the chain's first chunk, the masks are bits
        b.stream_h2 = net.train.d_stream_bwd_h2;
        b.descale = net.train.d_descale_bwd;
        b.gain = net.train.d_gain_bwd;
        b.loose = ps.loose;
        for (int i = 0; i < a.D; ++i) b.fwd.mask[i] = ps.mask[i];
        TrainTimer timer(ps.ctx, s, 1, ps.P);
        HIP_TRY(launch_mlp_bwd_h2(b, s));
    } else {
        TrainTimer timer(ps.ctx, s, 1, ps.P);
        HIP_TRY(launch_mlp_bwd(b, s));
    }
    int rc;
    if (ps.blocked && !(pair_dw && ps.pair_backward && a.input_ch <= 64 && a.D <= kMaxGradJobs && gemm_tn_is_small(out.out, a.W))) {
        set_error("internal: the blocked activation layout was chosen for a pass whose weight gradients cannot read it");
        return NERF_E_INVALID;
    }
    if ((rc = grad_linear(net, out, ps.d_raw, ps.dC, ps.h[a.D - 1], ps.h_ld[a.D - 1], ps.P, sc, s, eq, ps.blocked))) return rc;
    if (!sc.accumulate)
        HIP_TRY(hipMemsetAsync(net.train.d_grad + views.w_off, 0, ((size_t)views.out * views.in + views.out) * sizeof(float), s));
    float* grad = net.train.d_grad;
    GradBatch wide{}, narrow{}, pairs{}, narrow_pair{};
    auto job = [&](GradBatch& bt, const LinearDesc& d, const float* dY, const float* X, int ldx, int n0, int n1, bool with_db,
                   const unsigned* a_max = nullptr, const unsigned* b_max = nullptr, int blocked = 0, int b_first = 0) {
        GradJob& j = bt.job[bt.n++];
        j = GradJob{dY, a.W, X, ldx, d.out, n0, n1, grad + d.w_off, d.in, with_db ? grad + d.b_off : nullptr,
                    nullptr, nullptr, a_max, b_max, grad_exps(net, (int)(&d - &net.linears[0]), eq)};
        j.blocked = blocked;
        j.b_first = b_first;
    };
    if (ps.blocked) {
        // as backward_pass_fused: hidden-width operands blocked by 32 points, the gamma(x) columns' jobs a blocked dY against the
        // narrow row-major gamma(x) (rows of 64, zero-padded)
        GradBatch& nb = narrow_pair_wanted() ? narrow_pair : narrow;
        for (int i = a.D - 1; i >= 0; --i) {
            const LinearDesc& d = net.linears[i];
            if (d.in >= a.W) {
                const int lead = d.in - a.W;
                job(pairs, d, ps.dz[i], ps.h[i - 1], a.W, lead, d.in, true, ps.maxes + i, ps.maxes + kBwdMaxKept + i - 1, 3, lead);
                if (lead > 0) job(nb, d, ps.dz[i], ps.in[i], ps.in_ld[i], 0, lead, false, ps.maxes + i, ps.maxes + kBwdMaxGammaX, 1);
            } else {
                job(nb, d, ps.dz[i], ps.in[i], ps.in_ld[i], 0, d.in, true, ps.maxes + i, ps.maxes + kBwdMaxGammaX, 1);
            }
        }
    } else if (a.input_ch > 64 || a.D > kMaxGradJobs) {
        for (int i = a.D - 1; i >= 0; --i)
            if ((rc = grad_linear(net, net.linears[i], ps.dz[i], a.W, ps.in[i], ps.in_ld[i], ps.P, sc, s, eq))) return rc;
        return NERF_OK;
    }
    if (!ps.blocked) {
    GradBatch& hidden = pair_dw ? pairs : wide;
    for (int i = a.D - 1; i >= 0; --i) {
        const LinearDesc& d = net.linears[i];
        if (d.in >= a.W) {
            const int lead = d.in - a.W;                                               // cat[gamma(x), h] (nerf.py:79-80)
            job(hidden, d, ps.dz[i], ps.in[i], ps.in_ld[i], lead, d.in, true, pair_dw ? ps.maxes + i : nullptr,
                pair_dw ? ps.maxes + kBwdMaxKept + i - 1 : nullptr);
            if (lead > 0) job(narrow, d, ps.dz[i], ps.in[i], ps.in_ld[i], 0, lead, false);
        } else {
            job(narrow, d, ps.dz[i], ps.in[i], ps.in_ld[i], 0, d.in, true);            // layer 0: gamma(x) only
        }
    }
    }
    for (GradBatch* bt : {&pairs, &wide, &narrow}) {
        if (bt->n == 0) continue;
        int n_slices = 256 / bt->n;
        const int64_t cap = (ps.P + 255) / 256;
        if (n_slices > cap) n_slices = (int)cap;
        if (n_slices < 1) n_slices = 1;
        int64_t pps = (ps.P + n_slices - 1) / n_slices;
        pps = (pps + kSlicePointQuantum - 1) / kSlicePointQuantum * kSlicePointQuantum;
        bt->n_slices = n_slices;
        bt->pts_per_slice = pps;
        bt->P = ps.P;
        bt->accumulate = sc.accumulate;
        TrainTimer timer(ps.ctx, s, bt == &narrow ? 3 : 2, ps.P);
        HIP_TRY(launch_grad_batch(*bt, bt != &narrow, sc.part, sc.part_floats, sc.dbp, sc.dbp_floats, s, bt == &pairs));
    }
    return narrow_pair_batch(ps, sc, narrow_pair, s);
}

int backward_pass(Pass& ps, const TnScratch& sc, hipStream_t s) {
    if (ps.fused_backward) return backward_pass_fused(ps, sc, s);
    const PackedNet& net = *ps.net;
    const nerf_arch& a = net.arch;
    const bool eq = ps.eq;
    const float* prm = eq ? net.d_params_eq : net.d_params;      // the network whose units the forward pass kept
    const float* hl = ps.h[a.D - 1];
    const int hl_ld = ps.h_ld[a.D - 1];
    float* dh = ps.g_a;     // gradient w.r.t. the (post-ReLU, masked to pre-activation) output of the current layer
    float* dh_next = ps.g_b;
    int rc;
    if (a.use_viewdirs) {
        const LinearDesc &views = net.linears[a.D], &feat = net.linears[a.D + 1], &alpha = net.linears[a.D + 2],
                         &rgb = net.linears[a.D + 3];
        // rgb_linear
        if ((rc = grad_linear(net, rgb, ps.d_raw, ps.C, ps.hv, views.out, ps.P, sc, s, eq))) return rc;
        GemmRows g1{ps.d_raw, ps.C, prm + rgb.w_off, rgb.in, ps.g_hv, views.out, ps.P, views.out, 3, nullptr, 0,
                    ps.hv, views.out, 0};
        HIP_TRY(launch_gemm_rows(g1, s));                       // d(pre-activation of the view layer)
        // views_linears[0]
        if ((rc = grad_linear(net, views, ps.g_hv, views.out, ps.vcat, ps.vcat_ld, ps.P, sc, s, eq))) return rc;
        GemmRows g2{ps.g_hv, views.out, prm + views.w_off, views.in, dh_next, a.W, ps.P, a.W, views.out, nullptr, 0,
                    nullptr, 0, 0};
        HIP_TRY(launch_gemm_rows(g2, s));                       // d feature (first W input columns; no ReLU on feature)
        // feature_linear and alpha_linear both read the last trunk output
        if ((rc = grad_linear(net, feat, dh_next, a.W, hl, hl_ld, ps.P, sc, s, eq))) return rc;
        if ((rc = grad_linear(net, alpha, ps.d_raw + 3, ps.C, hl, hl_ld, ps.P, sc, s, eq))) return rc;
        GemmRows g3{dh_next, a.W, prm + feat.w_off, a.W, dh, a.W, ps.P, a.W, a.W, nullptr, 0, nullptr, 0, 0};
        HIP_TRY(launch_gemm_rows(g3, s));
        GemmRows g4{ps.d_raw + 3, ps.C, prm + alpha.w_off, a.W, dh, a.W, ps.P, a.W, 1, nullptr, 0, hl, hl_ld, 1};
        HIP_TRY(launch_gemm_rows(g4, s));                       // += dsigma * w_alpha, then ReLU mask of the trunk output
    } else {
        const LinearDesc& out = net.linears[a.D + 1];
        if ((rc = grad_linear(net, out, ps.d_raw, ps.C, hl, hl_ld, ps.P, sc, s, eq))) return rc;
        GemmRows g1{ps.d_raw, ps.C, prm + out.w_off, a.W, dh, a.W, ps.P, a.W, out.out, nullptr, 0, hl, hl_ld, 0};
        HIP_TRY(launch_gemm_rows(g1, s));
        // views_linears is never evaluated without viewdirs: its gradient is zero
        const LinearDesc& views = net.linears[a.D];
        if (!sc.accumulate)
            HIP_TRY(hipMemsetAsync(net.train.d_grad + views.w_off, 0, ((size_t)views.out * views.in + views.out) * sizeof(float), s));
    }
    for (int i = a.D - 1; i >= 0; --i) {
        const LinearDesc& d = net.linears[i];
        if ((rc = grad_linear(net, d, dh, a.W, ps.in[i], ps.in_ld[i], ps.P, sc, s, eq))) return rc;
        if (i == 0) break;
        // d h_{i-1} = dh W_i[:, hidden columns] masked by ReLU'(layer i-1)
        const int col0 = ((net.skip_in_mask >> i) & 1) ? a.input_ch : 0;
        GemmRows g{dh, a.W, prm + d.w_off + col0, d.in, dh_next, a.W, ps.P, a.W, a.W, nullptr, 0,
                   ps.h[i - 1], ps.h_ld[i - 1], 0};
        HIP_TRY(launch_gemm_rows(g, s));
        float* t = dh;
        dh = dh_next;
        dh_next = t;
    }
    return NERF_OK;
}

// Which network's units a pass runs in, and which backward kernel follows (after precision / fused_backward are set): the
// fp16-pair forward kernel evaluates the row-equalised network, so a pass it opens stays in that network's units to the
// end - the kept activations, the backward-data pass on the equalised transposed weights, the weight gradients, which
// are brought back to the plain parameters' by exact powers of two where their slices are added up (GradJob::ex).
void set_units(Pass& ps) {
    const nerf_arch& a = ps.net->arch;
    const bool fused_forward = !gemm_forward_requested() && a.D <= kMaxDepth && a.W == kWidth;
    // (networks without view directions, round 4: the same three fp16-pair kernels, output_linear as one more chunk in both
    // directions - when the whole pass can stay on them: a head of <= 8 channels and the fused backward stream, api.cpp)
    static const bool noviews_pair = [] {      // NERF_TRAIN_NOVIEWS=f32: the fused fp32 kernels for them (A/B)
        const char* e = getenv("NERF_TRAIN_NOVIEWS");
        return !(e && (e[0] == 'f' || e[0] == 'F') && e[1] == '3');
    }();
    const bool arch_ok = a.use_viewdirs || (noviews_pair && ps.fused_backward && pair_bwd_allowed() && ps.net->out_ch <= 8 && a.D >= 2);
    ps.eq = ps.precision == NERF_PRECISION_F16X2 && pair_forward_allowed() && fused_forward && arch_ok &&
            (uint64_t)ps.P * (uint64_t)(a.W + a.input_ch + 4) * 4u < ((uint64_t)1 << 32);
    ps.pair_backward = ps.eq && ps.fused_backward && pair_bwd_allowed();
    // Blocked by 32 points (MlpStore::blocked; NERF_TRAIN_BLOCKED=0: row-major, the A/B switch): when the three fp16-pair
    // kernels run the pass and the weight gradients go through the batched launches that read the layout - the conditions
    // backward_pass_fused puts on `batched`, `pair_dw`, `feat_known` and the alpha rider
    static const bool blocked_wanted = [] {
        const char* e = getenv("NERF_TRAIN_BLOCKED");
        return !(e && *e == '0');
    }();
    static const bool rider_on = [] {
        const char* e = getenv("NERF_TRAIN_DW_RIDER");
        return !(e && *e == '0');
    }();
    const bool dw_reads_blocked =
        a.use_viewdirs ? rider_on && grad_pair_takes_riders() && gemm_tn_is_direct(a.W / 2) && a.input_ch_views <= 64 &&
                             a.D + 2 <= kMaxGradJobs && ps.net->out_ch == 4
                       : a.D <= kMaxGradJobs && gemm_tn_is_small(ps.net->out_ch, a.W);      // (backward_pass_fused_noviews)
    ps.blocked = blocked_wanted && ps.pair_backward && pair_dw_allowed() && gemm_tn_is_direct(a.W) && a.input_ch <= 64 &&
                 dw_reads_blocked && (uint64_t)ps.P * 1024u < ((uint64_t)1 << 32);
}

}  // namespace

extern "C" {

int nerf_train_step(nerf_ctx* c, const nerf_train_args* r) {
    if (!c || !r || !r->rays || !r->target || r->n_rays <= 0) {
        set_error("nerf_train_step: invalid argument");
        return NERF_E_INVALID;
    }
    const int64_t N = r->n_rays;
    const int Sc = r->N_samples, Si = r->N_importance, Sf = Sc + Si;
    if (r->ray_stride != 8 && r->ray_stride != 11) {
        set_error("ray_stride must be 8 or 11 floats (got %d)", r->ray_stride);
        return NERF_E_INVALID;
    }
    if (Sc < 1 || Si < 0 || Sf > 4096 || (Si > 0 && Sc < 3)) {
        set_error("unsupported sample counts N_samples=%d N_importance=%d", Sc, Si);
        return NERF_E_INVALID;
    }
    if (r->perturb && (!r->t_rand || (Si > 0 && !r->u_rand))) {
        set_error("perturb > 0 requires t_rand (and u_rand with N_importance > 0): the caller owns the RNG");
        return NERF_E_INVALID;
    }
    if (r->slot_coarse < 0 || r->slot_coarse >= NERF_NUM_SLOTS || !c->nets[r->slot_coarse].loaded) {
        set_error("no weights loaded in slot %d", r->slot_coarse);
        return NERF_E_STATE;
    }
    PackedNet& nc = c->nets[r->slot_coarse];
    PackedNet* nfp = &nc;
    if (Si > 0 && r->slot_fine >= 0) {
        if (r->slot_fine >= NERF_NUM_SLOTS || !c->nets[r->slot_fine].loaded) {
            set_error("no weights loaded in slot %d", r->slot_fine);
            return NERF_E_STATE;
        }
        nfp = &c->nets[r->slot_fine];
    }
    PackedNet& nf = *nfp;
    // network_fine=None with N_importance > 0 (nerf.ipynb:471: run_fn = network_fn): both passes go through one
    // network and its gradient is the sum over the passes
    const bool shared = (&nf == &nc) && Si > 0;
    for (PackedNet* n : {&nc, &nf}) {
        if (n->arch.use_viewdirs && r->ray_stride < 11) {
            set_error("the model uses viewdirs but rays carry only %d columns", r->ray_stride);
            return NERF_E_INVALID;
        }
        if (n->out_ch < 4) {
            set_error("training needs a model with >= 4 output channels");
            return NERF_E_INVALID;
        }
    }
    DeviceGuard guard(c->device);
    hipStream_t s = (hipStream_t)r->stream;
    ScratchScope scope(c, s);
    HIP_TRY(scope.status);
    int rc;
    // Precision guard (nerf_mi355x.h): events counted by work that has completed since the last look - earlier steps of
    // this loop, typically - move the training path to the fp32 kernels, from this step on and until nerf_set_precision
    bool fell_back = false;
    if (c->train_precision == NERF_PRECISION_F16X2 && !c->train_force_f32 && take_new_loose_train(c) > 0) {
        c->train_force_f32 = true;
        fell_back = true;
    }
    const int precision = c->train_force_f32 ? NERF_PRECISION_F32 : c->train_precision;
    for (PackedNet* n : {&nc, &nf}) {
        const bool fresh = !n->train.ready;
        if ((rc = ensure_train_state(c, *n))) return rc;
        if (fresh && (rc = mark_params_changed(*n, s, false))) return rc;
    }

    // workspace: sampling buffers + both passes + split-K partials
    const int64_t Pc = N * Sc, Pf = Si ? N * Sf : 0;
    const int n_slices = 256;
    const size_t part_floats = (size_t)n_slices * 256 * (size_t)(nc.arch.W + nc.arch.input_ch + 64);
    const size_t small = (size_t)N * (Sc * 2 + (Si ? Si + Sf * 2 : 0) + 24) + 8192;
    rc = ensure_workspace(c, arena_bytes({small, pass_floats(nc, Pc), Si ? pass_floats(nf, Pf) : 1, part_floats,
                                          (size_t)n_slices * 512}) +
                                 (1 << 20));
    if (rc != NERF_OK) return rc;
    Arena ar(c->ws);
    float* z_c = ar.take((size_t)N * Sc);
    float* w_c = ar.take((size_t)N * Sc);
    float* rgb_c = ar.take((size_t)N * 3);
    float* g_c = ar.take((size_t)N * 3);
    float* z_s = Si ? ar.take((size_t)N * Si) : nullptr;
    float* z_f = Si ? ar.take((size_t)N * Sf) : nullptr;
    float* w_f = Si ? ar.take((size_t)N * Sf) : nullptr;
    float* rgb_f = Si ? ar.take((size_t)N * 3) : nullptr;
    float* g_f = Si ? ar.take((size_t)N * 3) : nullptr;
    double* red = (double*)ar.take(2048);   // 2 x 512 doubles of per-block partials
    float* loss_dev = ar.take(4);
    TnScratch sc{ar.take(part_floats), ar.take((size_t)n_slices * 512), n_slices, 0, 0, part_floats, (size_t)n_slices * 512};

    const bool fused_glue = !glue_legacy();
    // one block of words that the prologue zeroes: both passes' running maxima and the loss kernel's ticket
    unsigned* zero_block = (unsigned*)ar.take(2 * kBwdMaxSlots + 32);
    double* loss_part = (double*)ar.take((size_t)4 * N + 16);      // [2][N] per-ray sums of squares
    Pass pc;
    pc.net = &nc;
    pc.N = N;
    pc.P = Pc;
    pc.S = Sc;
    pc.fused_backward = !gemm_backward_requested() && nc.train.d_stream_bwd != nullptr && (nc.out_ch == 4 || !nc.arch.use_viewdirs);
    pc.precision = precision;
    pc.ctx = c;
    pc.loose = c->d_loose + kLooseTrain;
    pc.maxes = zero_block;
    set_units(pc);
    carve_pass(ar, pc);
    Pass pf;
    if (Si) {
        pf.net = &nf;
        pf.N = N;
        pf.P = Pf;
        pf.S = Sf;
        pf.fused_backward = !gemm_backward_requested() && nf.train.d_stream_bwd != nullptr && (nf.out_ch == 4 || !nf.arch.use_viewdirs);
        pf.precision = precision;
        pf.ctx = c;
        pf.loose = c->d_loose + kLooseTrain;
        pf.maxes = zero_block + kBwdMaxSlots;
        set_units(pf);
        carve_pass(ar, pf);
    }
    unsigned* ticket = zero_block + 2 * kBwdMaxSlots;
    float* rgb_last = Si ? rgb_f : rgb_c;

    if (fused_glue) {
        // ---- forward (render(..., retraw=True, **render_kwargs_train), nerf.ipynb:1258), the small stages fused ----
        const Prologue pro{r->lindisp, r->perturb ? r->t_rand : nullptr, zero_block, 2 * kBwdMaxSlots + 32};
        if ((rc = forward_pass(pc, r->rays, r->ray_stride, z_c, s, &pro))) return rc;
        if (Si) {
            HIP_TRY(launch_train_mid(pc.raw, pc.C, z_c, r->rays + 3, r->ray_stride, r->noise0, r->white_bkgd, N, Sc, rgb_c, w_c,
                                     r->perturb ? r->u_rand : nullptr, Si, z_f, s));      // z_samples are detached (nerf.ipynb:464)
            if (r->z_vals_fine_in)      // (parity tests: the fine pass at the reference's depths)
                HIP_TRY(hipMemcpyAsync(z_f, r->z_vals_fine_in, (size_t)N * Sf * sizeof(float), hipMemcpyDeviceToDevice, s));
            if ((rc = forward_pass(pf, r->rays, r->ray_stride, z_f, s))) return rc;
        }
        // ---- raw2outputs of the last pass, loss = img2mse(rgb, target) [+ img2mse(rgb0, target)] (nerf.ipynb:1262-1272),
        //      and the backward of both raw2outputs: one launch ----
        Pass& pl = Si ? pf : pc;
        TrainEpilogue e{};
        e.rays_d = r->rays + 3;
        e.d_ld = r->ray_stride;
        e.target = r->target;
        e.N = N;
        e.white_bkgd = r->white_bkgd;
        e.raw_l = pl.raw; e.C_l = pl.C; e.z_l = Si ? z_f : z_c; e.noise_l = Si ? r->noise : r->noise0; e.S_l = Si ? Sf : Sc;
        e.d_raw_l = pl.d_raw;
        e.dC_l = pl.dC;
        if (Si) {
            e.raw_c = pc.raw; e.C_c = pc.C; e.z_c = z_c; e.noise_c = r->noise0; e.S_c = Sc; e.d_raw_c = pc.d_raw; e.dC_c = pc.dC; e.rgb_c = rgb_c;
        }
        e.out_rgb = r->rgb_map ? r->rgb_map : rgb_last;
        e.out_rgb0 = Si ? r->rgb0 : nullptr;
        e.part = loss_part;
        e.ticket = ticket;
        e.loss_dev = loss_dev;
        e.out_loss = r->loss;
        e.out_stats = r->stats;
        HIP_TRY(launch_train_epilogue(e, s));
        // ---- backward ----
        sc.P = Pc;
        if ((rc = backward_pass(pc, sc, s))) return rc;
        nc.train.grads_valid = true;
        if (Si) {
            sc.P = Pf;
            sc.accumulate = shared ? 1 : 0;
            if ((rc = backward_pass(pf, sc, s))) return rc;
            nf.train.grads_valid = true;
        }
    } else {
    // ---- forward (render(..., retraw=True, **render_kwargs_train), nerf.ipynb:1258) ----
    for (Pass* p : {&pc, &pf})
        if (p->maxes) HIP_TRY(hipMemsetAsync(p->maxes, 0, kBwdMaxSlots * sizeof(unsigned), s));
    HIP_TRY(launch_stratified(r->rays, r->ray_stride, N, Sc, r->lindisp, r->perturb ? r->t_rand : nullptr, z_c, s));
    if ((rc = forward_pass(pc, r->rays, r->ray_stride, z_c, s))) return rc;
    HIP_TRY(launch_composite(pc.raw, pc.C, z_c, r->rays + 3, r->ray_stride, r->noise0, r->white_bkgd, N, Sc, rgb_c, nullptr,
                             nullptr, w_c, nullptr, s));
    if (Si) {
        HIP_TRY(launch_sample_pdf(nullptr, w_c, Sc, 1, z_c, r->perturb ? r->u_rand : nullptr, N, Sc - 1, Si, z_s, z_f,
                                  nullptr, s));                     // z_samples are detached (nerf.ipynb:464)
        if (r->z_vals_fine_in)
            HIP_TRY(hipMemcpyAsync(z_f, r->z_vals_fine_in, (size_t)N * Sf * sizeof(float), hipMemcpyDeviceToDevice, s));
        if ((rc = forward_pass(pf, r->rays, r->ray_stride, z_f, s))) return rc;
        HIP_TRY(launch_composite(pf.raw, pf.C, z_f, r->rays + 3, r->ray_stride, r->noise, r->white_bkgd, N, Sf, rgb_f,
                                 nullptr, nullptr, w_f, nullptr, s));
    }
    // ---- loss = img2mse(rgb, target) [+ img2mse(rgb0, target)] and its gradients (nerf.ipynb:1262-1272) ----
    HIP_TRY(launch_mse(rgb_last, r->target, N * 3, Si ? g_f : g_c, red, loss_dev, s));
    if (Si) HIP_TRY(launch_mse(rgb_c, r->target, N * 3, g_c, red + 512, loss_dev + 1, s));
    if (r->loss) {
        HIP_TRY(hipMemcpyAsync(r->loss, loss_dev, sizeof(float) * (Si ? 2 : 1), hipMemcpyDeviceToDevice, s));
    }
    if (r->stats) HIP_TRY(launch_train_stats(loss_dev, Si > 0, r->stats, s));
    if (r->rgb_map) HIP_TRY(hipMemcpyAsync(r->rgb_map, rgb_last, (size_t)N * 3 * sizeof(float), hipMemcpyDeviceToDevice, s));
    if (r->rgb0 && Si) HIP_TRY(hipMemcpyAsync(r->rgb0, rgb_c, (size_t)N * 3 * sizeof(float), hipMemcpyDeviceToDevice, s));

    // ---- backward ----
    HIP_TRY(launch_composite_bwd(pc.raw, pc.C, z_c, r->rays + 3, r->ray_stride, r->noise0, r->white_bkgd, N, Sc, g_c,
                                 pc.d_raw, s, pc.dC));
    sc.P = Pc;
    if ((rc = backward_pass(pc, sc, s))) return rc;
    nc.train.grads_valid = true;
    if (Si) {
        HIP_TRY(launch_composite_bwd(pf.raw, pf.C, z_f, r->rays + 3, r->ray_stride, r->noise, r->white_bkgd, N, Sf, g_f,
                                     pf.d_raw, s, pf.dC));
        sc.P = Pf;
        sc.accumulate = shared ? 1 : 0;
        if ((rc = backward_pass(pf, sc, s))) return rc;
        nf.train.grads_valid = true;
    }
    }
    // ---- optimizer.step() (torch.optim.Adam, nerf.ipynb:905, :1275) ----
    bool mirrored = false;
    if (r->apply_update) {
        if (r->step < 1) {
            set_error("nerf_train_step: step must be the 1-based Adam step count");
            return NERF_E_INVALID;
        }
        for (PackedNet* n : {&nc, &nf}) {
            HIP_TRY(launch_adam(n->d_params, n->train.d_grad, n->train.d_m, n->train.d_v, (int64_t)n->n_params, r->lr,
                                r->beta1, r->beta2, r->eps, r->step, s));
            if ((rc = mark_params_changed(*n, s, true))) return rc;
            if (!Si || shared) break;
        }
        // the equalised copies of both networks follow at once (one pair of launches for the two): the next step - or a
        // render - needs them anyway
        if (pc.eq) {
            PackedNet* both[2] = {&nc, &nf};
            const int n_nets = (Si && !shared) ? 2 : 1;
            if (fused_glue) {
                if ((rc = refresh_after_step(c, both, n_nets, s, &mirrored))) return rc;
            } else if ((rc = refresh_h2_many(both, n_nets, s))) return rc;
        }
    }
    if (!mirrored) HIP_TRY(mirror_loose(c, s));
    if (fell_back) {
        set_error("nerf_train_step: the fp16-pair kernels' output-scale bound was loose in an earlier step (some activations "
                  "kept fewer than 24 bits with these weights); training continues on the fp32 kernels");
        return NERF_W_PRECISION_FALLBACK;
    }
    return NERF_OK;
}

static int copy_flat(nerf_ctx* c, int slot, const float* d_flat, float* const* tensors, int n_tensors, const char* what) {
    if (!c || !tensors || slot < 0 || slot >= NERF_NUM_SLOTS || !c->nets[slot].loaded) {
        set_error("%s: invalid slot or NULL argument", what);
        return NERF_E_INVALID;
    }
    PackedNet& net = c->nets[slot];
    if (n_tensors != (int)net.linears.size() * 2) {
        set_error("%s: expected %d tensors, got %d", what, (int)net.linears.size() * 2, n_tensors);
        return NERF_E_INVALID;
    }
    if (!d_flat) {
        set_error("%s: nothing to read (no training step has run on slot %d)", what, slot);
        return NERF_E_STATE;
    }
    DeviceGuard guard(c->device);
    HIP_TRY(hipDeviceSynchronize());
    for (size_t k = 0; k < net.linears.size(); ++k) {
        const LinearDesc& d = net.linears[k];
        HIP_TRY(hipMemcpy(tensors[2 * k], d_flat + d.w_off, (size_t)d.out * d.in * sizeof(float), hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(tensors[2 * k + 1], d_flat + d.b_off, (size_t)d.out * sizeof(float), hipMemcpyDeviceToHost));
    }
    return NERF_OK;
}

int nerf_get_weights(nerf_ctx* c, int slot, float* const* tensors, int n_tensors) {
    return copy_flat(c, slot, (c && slot >= 0 && slot < NERF_NUM_SLOTS) ? c->nets[slot].d_params : nullptr, tensors,
                     n_tensors, "nerf_get_weights");
}

int nerf_get_gradients(nerf_ctx* c, int slot, float* const* tensors, int n_tensors) {
    const float* g = nullptr;
    if (c && slot >= 0 && slot < NERF_NUM_SLOTS && c->nets[slot].train.grads_valid) g = c->nets[slot].train.d_grad;
    return copy_flat(c, slot, g, tensors, n_tensors, "nerf_get_gradients");
}

/* torch.optim.Adam's per-parameter state (exp_avg, exp_avg_sq; the step count is the caller's), so that the
 * optimizer_state_dict of a checkpoint (nerf.ipynb:1290-1299, reloaded at :925-932) survives a round trip. */
int nerf_get_adam_state(nerf_ctx* c, int slot, float* const* exp_avg, float* const* exp_avg_sq, int n_tensors) {
    const float *m = nullptr, *v = nullptr;
    if (c && slot >= 0 && slot < NERF_NUM_SLOTS && c->nets[slot].train.ready) {
        m = c->nets[slot].train.d_m;
        v = c->nets[slot].train.d_v;
    }
    int rc = copy_flat(c, slot, m, exp_avg, n_tensors, "nerf_get_adam_state");
    if (rc == NERF_OK) rc = copy_flat(c, slot, v, exp_avg_sq, n_tensors, "nerf_get_adam_state");
    return rc;
}

int nerf_set_adam_state(nerf_ctx* c, int slot, const float* const* exp_avg, const float* const* exp_avg_sq, int n_tensors) {
    if (!c || !exp_avg || !exp_avg_sq || slot < 0 || slot >= NERF_NUM_SLOTS || !c->nets[slot].loaded) {
        set_error("nerf_set_adam_state: invalid slot or NULL argument");
        return NERF_E_INVALID;
    }
    PackedNet& net = c->nets[slot];
    if (n_tensors != (int)net.linears.size() * 2) {
        set_error("nerf_set_adam_state: expected %d tensors, got %d", (int)net.linears.size() * 2, n_tensors);
        return NERF_E_INVALID;
    }
    DeviceGuard guard(c->device);
    const bool fresh = !net.train.ready;
    int rc = ensure_train_state(c, net);
    if (rc != NERF_OK) return rc;
    if (fresh && (rc = mark_params_changed(net, nullptr, false))) return rc;
    HIP_TRY(hipDeviceSynchronize());
    for (size_t k = 0; k < net.linears.size(); ++k) {
        const LinearDesc& d = net.linears[k];
        const size_t nw = (size_t)d.out * d.in * sizeof(float), nb = (size_t)d.out * sizeof(float);
        for (int which = 0; which < 2; ++which) {
            const float* const* src = which ? exp_avg_sq : exp_avg;
            float* dst = which ? net.train.d_v : net.train.d_m;
            if (!src[2 * k] || !src[2 * k + 1]) {
                set_error("nerf_set_adam_state: tensor %zu is NULL", 2 * k);
                return NERF_E_INVALID;
            }
            HIP_TRY(hipMemcpy(dst + d.w_off, src[2 * k], nw, hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy(dst + d.b_off, src[2 * k + 1], nb, hipMemcpyHostToDevice));
        }
    }
    return NERF_OK;
}

}  // extern "C"
