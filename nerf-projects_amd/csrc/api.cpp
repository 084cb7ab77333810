// C ABI of libnerf_mi355x.so (declared in include/nerf_mi355x.h): context, weights,
// workspace, the stage entry points and the render_rays pipeline. Host code only; every
// device function lives in mlp_kernel.hip / ray_kernels.hip.
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <utility>
#include <vector>

#include "nerf_internal.h"

namespace nerf {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

}  // namespace nerf

#include "ctx_internal.h"

using namespace nerf;

namespace {

int run_mlp(nerf_ctx* c, MlpLaunch& a, const PackedNet& net_in, int mode, hipStream_t s) {
    PackedNet& net = const_cast<PackedNet&>(net_in);   // (the lazily refreshed fp16-pair data are a cache of the parameters)
    if (c->precision == NERF_PRECISION_F16X2 && net.h2_dirty) {
        const int rc = refresh_h2(net, s);          // the weights have been trained since the last fp16-pair launch
        if (rc != NERF_OK) return rc;
    }
    if (c->precision != NERF_PRECISION_F16X2 && net.f32_dirty) {
        const int rc = refresh_f32(net, s);
        if (rc != NERF_OK) return rc;
    }
    a.stream = net.d_stream;
    a.stream_h2 = net.d_stream_h2;
    a.descale = net.d_descale;
    a.gain = net.d_gain;
    a.loose = c->d_loose;
    a.bias = c->precision == NERF_PRECISION_F16X2 ? net.d_bias_h2 : net.d_bias;
    a.n_chunks = net.n_chunks;
    a.n_bias_tiles = net.n_bias_tiles;
    a.D = net.arch.D;
    a.skip_in_mask = net.skip_in_mask;
    a.use_viewdirs = net.arch.use_viewdirs;
    a.out_ch = net.out_ch;
    a.in_ch = net.arch.input_ch;
    a.in_ch_views = net.arch.input_ch_views;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (c->profiling) {
        for (hipEvent_t* e : {&e0, &e1}) {
            if (!c->pool.empty()) {
                *e = c->pool.back();
                c->pool.pop_back();
            } else {
                HIP_TRY(hipEventCreate(e));
            }
        }
        HIP_TRY(hipEventRecord(e0, s));
    }
#ifdef NERF_STAMPS
    // diagnostic build: one wave of workgroup 0 samples s_memtime through its first tile; dumped after every launch
    static unsigned long long* d_stamps = nullptr;
    if (!d_stamps) HIP_TRY(hipMalloc((void**)&d_stamps, 8192 * sizeof(unsigned long long)));
    HIP_TRY(hipMemsetAsync(d_stamps, 0, 8192 * sizeof(unsigned long long), s));
    a.stamps = d_stamps;
#endif
    if (c->precision == NERF_PRECISION_F16X2)
        HIP_TRY(launch_mlp_h2(a, mode, s));
    else
        HIP_TRY(launch_mlp(a, mode, s));
#ifdef NERF_STAMPS
    if (const char* path = getenv("NERF_STAMPS_FILE")) {
        std::vector<unsigned long long> h(8192);
        HIP_TRY(hipStreamSynchronize(s));
        HIP_TRY(hipMemcpy(h.data(), d_stamps, h.size() * sizeof(h[0]), hipMemcpyDeviceToHost));
        if (FILE* f = fopen(path, "wb")) {
            fwrite(h.data(), sizeof(h[0]), h.size(), f);
            fclose(f);
        }
    }
#endif
    if (c->profiling) {
        HIP_TRY(hipEventRecord(e1, s));
        c->events.emplace_back(e0, e1);
        c->prof_points += a.n_points;
    }
    return NERF_OK;
}

}  // namespace

namespace nerf {
// Precision guard: the counter of loose scale bounds follows the work of this call to the pinned host mirror.
hipError_t mirror_loose(nerf_ctx* c, hipStream_t s) {
    if ((c->precision != NERF_PRECISION_F16X2 && c->train_precision != NERF_PRECISION_F16X2) || !c->h_loose) return hipSuccess;
    return hipMemcpyAsync(c->h_loose, c->d_loose, kLooseWords * sizeof(unsigned), hipMemcpyDeviceToHost, s);
}
// events in the mirror that no call has reported yet; marks them reported
static unsigned take_new(const nerf_ctx* c, int word, unsigned& seen) {
    const unsigned v = c->h_loose ? ((volatile unsigned*)c->h_loose)[word] : 0u;
    const unsigned n = v - seen;      // (a counter only grows between resets; a reset zeroes both)
    seen = v;
    return n;
}
unsigned take_new_loose(nerf_ctx* c) { return take_new(c, 0, c->loose_seen); }
unsigned take_new_loose_train(nerf_ctx* c) { return take_new(c, kLooseTrain, c->train_loose_seen); }

EqualiseRefs equalise_refs(const nerf_arch& a, const std::vector<LinearDesc>& linears) {
    EqualiseRefs r{};
    const int D = a.D, W = a.W;
    r.n = (int)linears.size();
    for (int k = 0; k < r.n; ++k) {
        r.out[k] = linears[k].out;
        r.in[k] = linears[k].in;
        r.w_off[k] = (unsigned)linears[k].w_off;
        r.b_off[k] = (unsigned)linears[k].b_off;
        r.col_src[k] = -1;
    }
    auto reads = [&](int k, int src, int col0, int n) {
        r.col_src[k] = src;
        r.hid_col0[k] = col0;
        r.n_hid[k] = n;
    };
    int n = 0;
    for (int i = 0; i < D; ++i) {                      // trunk: relu(L_i h); a skip layer reads cat[gamma(x), h] (nerf.py:79-80)
        r.scale_rows[i] = 1;
        if (i > 0) reads(i, i - 1, linears[i].in - W, W);
        r.order[n++] = i;
    }
    if (a.use_viewdirs) {                              // linears: ..., views (D), feature (D+1), alpha (D+2), rgb (D+3)
        r.scale_rows[D + 1] = 1;                       // feature_linear: no ReLU, linear (nerf.py:89)
        reads(D + 1, D - 1, 0, W);
        reads(D + 2, D - 1, 0, W);                     // alpha_linear reads the trunk output; its row is an output
        r.scale_rows[D] = 1;                           // views_linears[0] on cat[feature, gamma(d)] (nerf.py:93-98)
        reads(D, D + 1, 0, W);
        reads(D + 3, D, 0, linears[D].out);            // rgb_linear
        r.order[n++] = D + 1;
        r.order[n++] = D + 2;
        r.order[n++] = D;
        r.order[n++] = D + 3;
    } else {                                           // linears: ..., views (D, unused by forward), output (D+1)
        reads(D + 1, D - 1, 0, W);
        r.order[n++] = D;
        r.order[n++] = D + 1;
    }
    return r;
}

int refresh_f32(PackedNet& net, hipStream_t s) {
    HIP_TRY(launch_gather(net.d_params, net.train.d_stream_table, (int64_t)net.stream_table.size(), net.d_stream, s));
    HIP_TRY(launch_gather(net.d_params, net.train.d_bias_table, (int64_t)net.bias_table.size(), net.d_bias, s));
    net.f32_dirty = false;
    return NERF_OK;
}

int refresh_h2_many(PackedNet* const* nets, int n, hipStream_t s) {
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
    HIP_TRY(launch_equalise_rows(n, params, refs, out, rexp, s));
    for (int i = 0; i < n; ++i) {
        PackedNet& net = *nets[i];
        HIP_TRY(launch_gather(net.d_params_eq, net.train.d_stream_table, (int64_t)net.stream_table.size(), net.d_stream_eq, s));
        HIP_TRY(launch_gather(net.d_params_eq, net.train.d_bias_table, (int64_t)net.bias_table.size(), net.d_bias_h2, s));
        HIP_TRY(launch_convert_stream_h2(net.d_stream_eq, net.d_chunk_layer, net.n_chunks, net.d_chunk_max, net.d_stream_h2,
                                         net.d_descale, s));
        HIP_TRY(launch_layer_gains(net.d_params_eq, gain_refs(net.arch, net.linears), net.d_gain, s));
        net.h2_dirty = false;
        if (net.train.bwd_is_eq) net.train.bwd_dirty = true;      // d_params_eq moved
    }
    return NERF_OK;
}

int refresh_h2(PackedNet& net, hipStream_t s) {
    PackedNet* one = &net;
    return refresh_h2_many(&one, 1, s);
}

GainRefs gain_refs(const nerf_arch& a, const std::vector<LinearDesc>& linears) {
    GainRefs r{};
    r.n = a.use_viewdirs ? a.D + 1 : a.D;
    for (int l = 0; l < r.n; ++l) {
        const LinearDesc& d = linears[l < a.D ? l : a.D + 1];   // feature_linear follows views_linears.0
        r.w_off[l] = d.w_off;
        r.b_off[l] = d.b_off;
        r.out[l] = d.out;
        r.in[l] = d.in;
    }
    return r;
}
}  // namespace nerf

namespace {

void free_net(PackedNet& n) {
    for (void* p : {(void*)n.d_stream, (void*)n.d_bias, (void*)n.d_params, (void*)n.train.d_grad, (void*)n.train.d_m,
                    (void*)n.train.d_v, (void*)n.train.d_wt, (void*)n.train.d_stream_table,
                    (void*)n.train.d_bias_table, (void*)n.train.d_bwd_table, (void*)n.train.d_stream_bwd, (void*)n.d_stream_h2, (void*)n.d_descale, (void*)n.d_chunk_layer,
                    (void*)n.d_chunk_max, (void*)n.d_gain, (void*)n.d_params_eq, (void*)n.d_stream_eq, (void*)n.d_bias_h2,
                    (void*)n.d_row_exp, (void*)n.d_eq_flags, (void*)n.train.d_stream_bwd_h2, (void*)n.train.d_descale_bwd, (void*)n.train.d_gain_bwd,
                    (void*)n.train.d_chunk_layer_bwd, (void*)n.train.d_chunk_max_bwd})
        if (p) (void)hipFree(p);
    n = PackedNet{};
}

// shapes of the state_dict tensors in the order nerf_load_weights receives them
std::vector<LinearDesc> describe_linears(const nerf_arch& a, uint32_t skip_in_mask) {
    std::vector<LinearDesc> L;
    size_t off = 0;
    auto add = [&](int out, int in) {
        LinearDesc d;
        d.out = out;
        d.in = in;
        d.w_off = off;
        off += (size_t)out * in;
        d.b_off = off;
        off += (size_t)out;
        L.push_back(d);
    };
    for (int i = 0; i < a.D; ++i)
        add(a.W, i == 0 ? a.input_ch : (((skip_in_mask >> i) & 1) ? a.W + a.input_ch : a.W));
    add(a.W / 2, a.input_ch_views + a.W);
    if (a.use_viewdirs) {
        add(a.W, a.W);
        add(1, a.W);
        add(3, a.W / 2);
    } else {
        add(a.output_ch, a.W);
    }
    return L;
}

const PackedNet* get_net(nerf_ctx* c, int slot) {
    if (slot < 0 || slot >= NERF_NUM_SLOTS) {
        set_error("slot %d out of range [0,%d)", slot, NERF_NUM_SLOTS);
        return nullptr;
    }
    if (!c->nets[slot].loaded) {
        set_error("no weights loaded in slot %d", slot);
        return nullptr;
    }
    return &c->nets[slot];
}

}  // namespace

extern "C" {

const char* nerf_last_error(void) { return g_err; }
const char* nerf_version(void) { return "nerf_mi355x 0.2 (gfx950; MLP arithmetic: fp16-pair MFMA [default] or fp32 MFMA)"; }

int nerf_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        set_error("hipGetDeviceCount failed: %s", hipGetErrorString(e));
        return NERF_E_HIP;
    }
    return n;
}

int nerf_num_weight_tensors(const nerf_arch* a) {
    if (!a) return NERF_E_INVALID;
    return 2 * a->D + 2 + (a->use_viewdirs ? 6 : 2);
}

int nerf_ctx_create(int device, nerf_ctx** out) {
    if (!out) {
        set_error("nerf_ctx_create: out is NULL");
        return NERF_E_INVALID;
    }
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        set_error("no HIP device available (%s); this library has no CPU fallback",
                  e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
        return NERF_E_HIP;
    }
    if (device < 0 || device >= n) {
        set_error("device %d out of range (%d visible)", device, n);
        return NERF_E_INVALID;
    }
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        set_error("device %d is %s; the kernels are built for gfx950 (MI355X) only", device, prop.gcnArchName);
        return NERF_E_INVALID;
    }
    nerf_ctx* c = new (std::nothrow) nerf_ctx();
    if (!c) return NERF_E_NOMEM;
    c->device = device;
    {
        DeviceGuard g(device);
        hipError_t e2 = hipMalloc((void**)&c->d_loose, kLooseWords * sizeof(unsigned));
        if (e2 == hipSuccess) e2 = hipMemset(c->d_loose, 0, kLooseWords * sizeof(unsigned));
        if (e2 == hipSuccess) e2 = hipHostMalloc((void**)&c->h_loose, kLooseWords * sizeof(unsigned), hipHostMallocDefault);
        if (e2 == hipSuccess) memset(c->h_loose, 0, kLooseWords * sizeof(unsigned));
        if (e2 == hipSuccess && hipHostGetDevicePointer((void**)&c->h_loose_dev, c->h_loose, 0) != hipSuccess) c->h_loose_dev = nullptr;
        if (e2 != hipSuccess) {
            set_error("nerf_ctx_create: device allocation failed: %s", hipGetErrorString(e2));
            delete c;
            return NERF_E_HIP;
        }
    }
    *out = c;
    return NERF_OK;
}

void nerf_ctx_destroy(nerf_ctx* c) {
    if (!c) return;
    DeviceGuard g(c->device);
    (void)hipDeviceSynchronize();
    for (auto& n : c->nets) free_net(n);
    if (c->ws) (void)hipFree(c->ws);
    if (c->frame_rays) (void)hipFree(c->frame_rays);
    if (c->d_loose) (void)hipFree(c->d_loose);
    if (c->h_loose) (void)hipHostFree(c->h_loose);
    if (c->scratch_done) (void)hipEventDestroy(c->scratch_done);
    for (auto& p : c->events) {
        (void)hipEventDestroy(p.first);
        (void)hipEventDestroy(p.second);
    }
    for (auto& sp : c->train_spans) {
        (void)hipEventDestroy(sp.e0);
        (void)hipEventDestroy(sp.e1);
    }
    for (auto e : c->pool) (void)hipEventDestroy(e);
    delete c;
}

int nerf_set_precision(nerf_ctx* c, int precision) {
    if (!c || (precision != NERF_PRECISION_F32 && precision != NERF_PRECISION_F16X2)) {
        set_error("nerf_set_precision: invalid argument");
        return NERF_E_INVALID;
    }
    c->precision = precision;
    c->train_precision = precision;
    c->train_force_f32 = false;      // an explicit choice ends a fallback of the training path (nerf_train_step)
    return NERF_OK;
}

int nerf_set_render_precision(nerf_ctx* c, int precision) {
    if (!c || (precision != NERF_PRECISION_F32 && precision != NERF_PRECISION_F16X2)) {
        set_error("nerf_set_render_precision: invalid argument");
        return NERF_E_INVALID;
    }
    c->precision = precision;        // (the training step's arithmetic and a fallback it is in stay as they are)
    return NERF_OK;
}

int nerf_get_precision(nerf_ctx* c) { return c ? c->precision : NERF_E_INVALID; }

int nerf_load_weights(nerf_ctx* c, int slot, const nerf_arch* arch, const float* const* tensors, int n_tensors) {
    if (!c || !arch || !tensors) {
        set_error("nerf_load_weights: NULL argument");
        return NERF_E_INVALID;
    }
    if (slot < 0 || slot >= NERF_NUM_SLOTS) {
        set_error("slot %d out of range [0,%d)", slot, NERF_NUM_SLOTS);
        return NERF_E_INVALID;
    }
    DeviceGuard g(c->device);
    float *hs = nullptr, *hb = nullptr;
    int nc = 0, nbt = 0, out_ch = 4;
    uint32_t mask = 0;
    int rc = pack_weights(*arch, tensors, n_tensors, &hs, &nc, &hb, &nbt, &mask, &out_ch);
    if (rc != NERF_OK) return rc;
    PackedNet& net = c->nets[slot];
    // the previous stream of this slot may still be in use by enqueued work
    hipError_t e = hipDeviceSynchronize();
    free_net(net);
    // flat master copy + the index tables that map the packed layouts back to it (for training)
    net.linears = describe_linears(*arch, mask);
    std::vector<float> flat, fake;
    {
        std::vector<const float*> fake_ptrs;
        size_t total = 0;
        for (const LinearDesc& d : net.linears) total += (size_t)d.out * d.in + d.out;
        flat.resize(total);
        fake.resize(total);
        for (size_t i = 0; i < total; ++i) fake[i] = (float)(i + 1);   // exact: total < 2^24
        if (total >= (1u << 24)) {
            free(hs);
            free(hb);
            set_error("model too large for the index tables (%zu parameters)", total);
            return NERF_E_INVALID;
        }
        for (size_t k = 0; k < net.linears.size(); ++k) {
            const LinearDesc& d = net.linears[k];
            memcpy(flat.data() + d.w_off, tensors[2 * k], (size_t)d.out * d.in * sizeof(float));
            memcpy(flat.data() + d.b_off, tensors[2 * k + 1], (size_t)d.out * sizeof(float));
            fake_ptrs.push_back(fake.data() + d.w_off);
            fake_ptrs.push_back(fake.data() + d.b_off);
        }
        float *ts = nullptr, *tb = nullptr;
        int tnc = 0, tnb = 0, toc = 0;
        uint32_t tm = 0;
        rc = pack_weights(*arch, fake_ptrs.data(), n_tensors, &ts, &tnc, &tb, &tnb, &tm, &toc);
        if (rc != NERF_OK) {
            free(hs);
            free(hb);
            return rc;
        }
        net.stream_table.resize((size_t)tnc * kChunkFloats);
        net.bias_table.resize((size_t)tnb * kBiasTileFloats);
        for (size_t i = 0; i < net.stream_table.size(); ++i) net.stream_table[i] = (int)ts[i] - 1;
        for (size_t i = 0; i < net.bias_table.size(); ++i) net.bias_table[i] = (int)tb[i] - 1;
        free(ts);
        free(tb);
        net.n_params = total;
        // index table of the fused backward-data kernel's stream (training; view-dependent networks only)
        net.bwd_table.clear();
        // (narrower networks, one-layer trunks without a view branch and heads of more than kBwdMaxOutRows channels train on
        // the layer-by-layer chain; so does a head whose per-register rows did not fit the bias block, pack_weights.cpp)
        if (arch->W == kWidth && (arch->use_viewdirs || (arch->output_ch <= kBwdMaxOutRows && arch->D >= 2 &&
                                                         nbt >= 8 * arch->D + 1 + 8 * arch->output_ch))) {
            float* tsb = nullptr;
            int nbc = 0;
            rc = pack_backward_stream(*arch, fake_ptrs.data(), mask, &tsb, &nbc);
            if (rc != NERF_OK) {
                free(hs);
                free(hb);
                return rc;
            }
            net.bwd_table.resize((size_t)nbc * kChunkFloats);
            for (size_t i = 0; i < net.bwd_table.size(); ++i) net.bwd_table[i] = (int)tsb[i] - 1;
            free(tsb);
        }
    }
    if (e == hipSuccess) e = hipMalloc((void**)&net.d_params, net.n_params * sizeof(float));
    if (e == hipSuccess) e = hipMemcpy(net.d_params, flat.data(), net.n_params * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc((void**)&net.d_stream, (size_t)nc * kChunkBytes);
    if (e == hipSuccess) e = hipMalloc((void**)&net.d_bias, (size_t)nbt * kBiasTileFloats * sizeof(float));
    if (e == hipSuccess) e = hipMemcpy(net.d_stream, hs, (size_t)nc * kChunkBytes, hipMemcpyHostToDevice);
    if (e == hipSuccess)
        e = hipMemcpy(net.d_bias, hb, (size_t)nbt * kBiasTileFloats * sizeof(float), hipMemcpyHostToDevice);
    free(hs);
    free(hb);
    // fp16-pair twin of the stream (NERF_PRECISION_F16X2), converted on the device
    const std::vector<int> layer_of = chunk_layers(*arch, mask);
    if (e == hipSuccess && (int)layer_of.size() != nc) {
        set_error("internal: %zu chunk scale groups for %d chunks", layer_of.size(), nc);
        return NERF_E_INVALID;
    }
    if (e == hipSuccess) e = hipMalloc((void**)&net.d_stream_h2, (size_t)(nc + kStreamTailChunks) * kChunkBytes);
    if (e == hipSuccess) e = hipMalloc((void**)&net.d_descale, (kMaxDepth + 3) * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void**)&net.d_chunk_layer, (size_t)nc * sizeof(int));
    if (e == hipSuccess) e = hipMalloc((void**)&net.d_chunk_max, (size_t)nc * sizeof(float));
    if (e == hipSuccess)
        e = hipMemcpy(net.d_chunk_layer, layer_of.data(), (size_t)nc * sizeof(int), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc((void**)&net.d_gain, 2 * (kMaxDepth + 2) * sizeof(float));
    // the row-equalised copy the fp16-pair kernel evaluates, and the index tables that cut streams out of parameters
    if (e == hipSuccess) e = hipMalloc((void**)&net.d_params_eq, net.n_params * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void**)&net.d_stream_eq, (size_t)nc * kChunkBytes);
    if (e == hipSuccess) e = hipMalloc((void**)&net.d_bias_h2, (size_t)nbt * kBiasTileFloats * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void**)&net.d_row_exp, (size_t)kMaxLinears * 256 * sizeof(int));
    if (e == hipSuccess) e = hipMalloc((void**)&net.train.d_stream_table, net.stream_table.size() * sizeof(int));
    if (e == hipSuccess) e = hipMalloc((void**)&net.train.d_bias_table, net.bias_table.size() * sizeof(int));
    if (e == hipSuccess)
        e = hipMemcpy(net.train.d_stream_table, net.stream_table.data(), net.stream_table.size() * sizeof(int), hipMemcpyHostToDevice);
    if (e == hipSuccess)
        e = hipMemcpy(net.train.d_bias_table, net.bias_table.data(), net.bias_table.size() * sizeof(int), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        net.arch = *arch;           // (refresh_h2 reads these; set again below with the rest)
        net.n_chunks = nc;
        net.n_bias_tiles = nbt;
        if (refresh_h2(net, nullptr) != NERF_OK) return NERF_E_HIP;
    }
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) {
        set_error("uploading packed weights failed: %s", hipGetErrorString(e));
        return NERF_E_HIP;
    }
    net.arch = *arch;
    net.n_chunks = nc;
    net.n_bias_tiles = nbt;
    net.skip_in_mask = mask;
    net.out_ch = out_ch;
    net.loaded = true;
    return NERF_OK;
}

int nerf_embed(nerf_ctx* c, const float* x, int64_t n, int multires, float* out, void* stream) {
    if (c && n == 0) return NERF_OK;
    if (!c || !x || !out || n < 0 || multires < 0 || multires > 16) {
        set_error("nerf_embed: invalid argument");
        return NERF_E_INVALID;
    }
    DeviceGuard g(c->device);
    HIP_TRY(launch_embed(x, n, multires, out, (hipStream_t)stream));
    return NERF_OK;
}

int nerf_mlp_forward(nerf_ctx* c, int slot, const float* x, int64_t B, float* out, void* stream) {
    if (c && B == 0) return NERF_OK;
    if (!c || !x || !out || B < 0) {
        set_error("nerf_mlp_forward: invalid argument");
        return NERF_E_INVALID;
    }
    const PackedNet* net = get_net(c, slot);
    if (!net) return NERF_E_STATE;
    DeviceGuard g(c->device);
    MlpLaunch a{};
    a.n_points = B;
    a.samples_per_ray = 1;
    a.x = x;
    a.x_ld = net->arch.input_ch + net->arch.input_ch_views;
    a.out = out;
    const int rc = run_mlp(c, a, *net, kInputEmbedded, (hipStream_t)stream);
    if (rc == NERF_OK) HIP_TRY(mirror_loose(c, (hipStream_t)stream));
    return rc;
}

int nerf_run_network(nerf_ctx* c, int slot, const float* pts, const float* viewdirs, int64_t n_rays,
                     int64_t n_samples, float* out, void* stream) {
    if (c && n_rays == 0) return NERF_OK;
    if (!c || !pts || !out || n_rays < 0 || n_samples <= 0) {
        set_error("nerf_run_network: invalid argument");
        return NERF_E_INVALID;
    }
    const PackedNet* net = get_net(c, slot);
    if (!net) return NERF_E_STATE;
    if (net->arch.use_viewdirs && !viewdirs) {
        set_error("nerf_run_network: the model uses viewdirs but none were given");
        return NERF_E_INVALID;
    }
    DeviceGuard g(c->device);
    MlpLaunch a{};
    a.n_points = n_rays * n_samples;
    a.samples_per_ray = n_samples;
    a.pts = pts;
    a.viewdirs = net->arch.use_viewdirs ? viewdirs : nullptr;
    a.out = out;
    const int rc = run_mlp(c, a, *net, kInputPoints, (hipStream_t)stream);
    if (rc == NERF_OK) HIP_TRY(mirror_loose(c, (hipStream_t)stream));
    return rc;
}

int nerf_raw2outputs(nerf_ctx* c, const float* raw, int C, const float* z_vals, const float* rays_d,
                     const float* noise, int white_bkgd, int64_t N, int S, float* rgb_map, float* disp_map,
                     float* acc_map, float* weights, float* depth_map, void* stream) {
    if (c && N == 0) return NERF_OK;
    if (!c || !raw || !z_vals || !rays_d || C < 4 || N < 0 || S < 1) {
        set_error("nerf_raw2outputs: invalid argument (C >= 4, S >= 1 required)");
        return NERF_E_INVALID;
    }
    DeviceGuard g(c->device);
    HIP_TRY(launch_composite(raw, C, z_vals, rays_d, 3, noise, white_bkgd, N, S, rgb_map, disp_map, acc_map, weights,
                             depth_map, (hipStream_t)stream));
    return NERF_OK;
}

int nerf_sample_pdf(nerf_ctx* c, const float* bins, const float* weights, const float* u, int64_t N, int M,
                    int n_samples, float* out, void* stream) {
    if (c && N == 0) return NERF_OK;
    if (!c || !bins || !weights || !out || N < 0 || M < 2 || n_samples < 1 || M > 4096 || n_samples > 4096) {
        set_error("nerf_sample_pdf: invalid argument (2 <= M <= 4096, 1 <= n_samples <= 4096)");
        return NERF_E_INVALID;
    }
    DeviceGuard g(c->device);
    HIP_TRY(launch_sample_pdf(bins, weights, M - 1, 0, nullptr, u, N, M, n_samples, out, nullptr, nullptr,
                              (hipStream_t)stream));
    return NERF_OK;
}

int nerf_stratified_z(nerf_ctx* c, const float* rays, int ray_stride, int64_t N, int N_samples, int lindisp,
                      const float* t_rand, float* z_vals, void* stream) {
    if (c && N == 0) return NERF_OK;
    if (!c || !rays || !z_vals || N < 0 || N_samples < 1 || ray_stride < 8) {
        set_error("nerf_stratified_z: invalid argument");
        return NERF_E_INVALID;
    }
    DeviceGuard g(c->device);
    HIP_TRY(launch_stratified(rays, ray_stride, N, N_samples, lindisp, t_rand, z_vals, (hipStream_t)stream));
    return NERF_OK;
}

int nerf_resample(nerf_ctx* c, const float* z_vals, const float* weights, const float* u, int64_t N, int S,
                  int n_samples, float* z_samples, float* z_merged, float* z_std, void* stream) {
    if (c && N == 0) return NERF_OK;
    if (!c || !z_vals || !weights || N < 0 || S < 3 || n_samples < 1 || S + n_samples > 4096 ||
        (!z_samples && !z_merged)) {
        set_error("nerf_resample: invalid argument (S >= 3, S + n_samples <= 4096, an output is required)");
        return NERF_E_INVALID;
    }
    DeviceGuard g(c->device);
    HIP_TRY(launch_sample_pdf(nullptr, weights, S, 1, z_vals, u, N, S - 1, n_samples, z_samples, z_merged, z_std,
                              (hipStream_t)stream));
    return NERF_OK;
}

static int render_rays_locked(nerf_ctx* c, const nerf_render_args* r) {
    if (!c || !r || (!r->rays && r->n_rays != 0) || r->n_rays < 0) {
        set_error("nerf_render_rays: NULL argument");
        return NERF_E_INVALID;
    }
    const int64_t N = r->n_rays;
    const int Sc = r->N_samples, Si = r->N_importance, Sf = Sc + Si;
    if (r->ray_stride != 8 && r->ray_stride != 11) {
        set_error("ray_stride must be 8 or 11 floats (got %d)", r->ray_stride);
        return NERF_E_INVALID;
    }
    if (Sc < 1 || Si < 0 || Sc > 4096 || Sf > 4096) {
        set_error("unsupported sample counts N_samples=%d N_importance=%d", Sc, Si);
        return NERF_E_INVALID;
    }
    if (Si > 0 && Sc < 3) {
        set_error("hierarchical sampling needs N_samples >= 3 (weights[...,1:-1] would be empty)");
        return NERF_E_INVALID;
    }
    if (r->perturb && !r->t_rand) {
        set_error("perturb > 0 requires t_rand (the caller owns the RNG)");
        return NERF_E_INVALID;
    }
    if (r->perturb && Si > 0 && !r->u_rand && !r->z_vals_fine_in) {
        set_error("perturb > 0 with N_importance > 0 requires u_rand (det = (perturb == 0))");
        return NERF_E_INVALID;
    }
    if (N * (int64_t)Sf > 0x7fffffffLL) {
        set_error("chunk too large: n_rays*(N_samples+N_importance) must stay below 2^31");
        return NERF_E_INVALID;
    }
    const PackedNet* nc = get_net(c, r->slot_coarse);
    if (!nc) return NERF_E_STATE;
    const PackedNet* nf = nc;
    if (Si > 0 && r->slot_fine >= 0) {
        nf = get_net(c, r->slot_fine);
        if (!nf) return NERF_E_STATE;
    }
    for (const PackedNet* n : {nc, nf}) {
        if (n->arch.use_viewdirs && r->ray_stride < 11) {
            set_error("the model uses viewdirs but rays carry only %d columns", r->ray_stride);
            return NERF_E_INVALID;
        }
        if (n->out_ch < 4) {
            set_error("render_rays needs a model with >= 4 output channels (got %d)", n->out_ch);
            return NERF_E_INVALID;
        }
    }
    if (N == 0) return NERF_OK;
    DeviceGuard g(c->device);
    hipStream_t s = (hipStream_t)r->stream;
    const int Cc = nc->out_ch, Cf = nf->out_ch;

    const size_t nN = (size_t)N;
    const bool raw_is_coarse = Si == 0;
    int rc = ensure_workspace(c, arena_bytes({nN * Sc, nN * Sc * Cc, nN * Sc, nN * (Si ? Si : 1),
                                              nN * (Si ? Sf : 1), nN * (Si ? (size_t)Sf * Cf : 1)}));
    if (rc != NERF_OK) return rc;
    Arena ar(c->ws);
    float* z_c = r->z_vals_coarse ? r->z_vals_coarse : ar.take(nN * Sc);
    float* raw_c = (raw_is_coarse && r->raw) ? r->raw : ar.take(nN * Sc * Cc);
    float* w_c = r->weights_coarse ? r->weights_coarse : ar.take(nN * Sc);

    HIP_TRY(launch_stratified(r->rays, r->ray_stride, N, Sc, r->lindisp, r->perturb ? r->t_rand : nullptr, z_c, s));
    MlpLaunch a{};
    a.n_points = N * Sc;
    a.samples_per_ray = Sc;
    a.rays = r->rays;
    a.ray_ld = r->ray_stride;
    a.z_vals = z_c;
    a.out = raw_c;
    rc = run_mlp(c, a, *nc, kInputRays, s);
    if (rc != NERF_OK) return rc;
    HIP_TRY(launch_composite(raw_c, Cc, z_c, r->rays + 3, r->ray_stride, r->noise0, r->white_bkgd, N, Sc,
                             Si ? r->rgb0 : r->rgb_map, Si ? r->disp0 : r->disp_map, Si ? r->acc0 : r->acc_map, w_c,
                             Si ? nullptr : r->depth_map, s));
    if (Si == 0) return NERF_OK;

    float* z_s = r->z_samples ? r->z_samples : ar.take(nN * Si);
    float* z_f = r->z_vals_fine ? r->z_vals_fine : ar.take(nN * Sf);
    float* raw_f = r->raw ? r->raw : ar.take(nN * Sf * Cf);
    // z_samples from the coarse weights[...,1:-1] over z_vals_mid; merged and sorted with z_vals
    HIP_TRY(launch_sample_pdf(nullptr, w_c, Sc, 1, z_c, r->perturb ? r->u_rand : nullptr, N, Sc - 1, Si, z_s, z_f,
                              r->z_std, s));
    if (r->z_vals_fine_in) {
        if (r->z_vals_fine) {
            HIP_TRY(hipMemcpyAsync(z_f, r->z_vals_fine_in, nN * Sf * sizeof(float), hipMemcpyDeviceToDevice, s));
        } else {
            z_f = const_cast<float*>(r->z_vals_fine_in);
        }
    }
    MlpLaunch b{};
    b.n_points = N * Sf;
    b.samples_per_ray = Sf;
    b.rays = r->rays;
    b.ray_ld = r->ray_stride;
    b.z_vals = z_f;
    b.out = raw_f;
    rc = run_mlp(c, b, *nf, kInputRays, s);
    if (rc != NERF_OK) return rc;
    HIP_TRY(launch_composite(raw_f, Cf, z_f, r->rays + 3, r->ray_stride, r->noise, r->white_bkgd, N, Sf, r->rgb_map,
                             r->disp_map, r->acc_map, r->weights_fine, r->depth_map, s));
    return NERF_OK;
}

int nerf_render_rays(nerf_ctx* c, const nerf_render_args* r) {
    if (!c || !r) {
        set_error("nerf_render_rays: NULL argument");
        return NERF_E_INVALID;
    }
    DeviceGuard g(c->device);
    ScratchScope scope(c, (hipStream_t)r->stream);
    HIP_TRY(scope.status);
    const int rc = render_rays_locked(c, r);
    if (rc == NERF_OK) HIP_TRY(mirror_loose(c, (hipStream_t)r->stream));
    return rc;
}

int nerf_generate_rays(nerf_ctx* c, const nerf_camera* cam, int64_t first_pixel, int64_t n_pixels, float* rays,
                       void* stream) {
    if (!c || !cam || first_pixel < 0 || n_pixels < 0) {
        set_error("nerf_generate_rays: invalid argument");
        return NERF_E_INVALID;
    }
    if (cam->H <= 0 || cam->W <= 0 || first_pixel + n_pixels > (int64_t)cam->H * cam->W) {
        set_error("nerf_generate_rays: pixels [%lld, %lld) outside the %dx%d image", (long long)first_pixel,
                  (long long)(first_pixel + n_pixels), cam->H, cam->W);
        return NERF_E_INVALID;
    }
    if (n_pixels == 0) return NERF_OK;
    if (!rays) {
        set_error("nerf_generate_rays: rays is NULL");
        return NERF_E_INVALID;
    }
    DeviceGuard g(c->device);
    HIP_TRY(launch_raygen(*cam, first_pixel, n_pixels, rays, (hipStream_t)stream));
    return NERF_OK;
}

int nerf_pack_rays(nerf_ctx* c, const nerf_camera* cam, const float* rays_o, int o_stride, const float* rays_d, int d_stride,
                   int64_t n, float* rays, void* stream) {
    if (!c || !cam || n < 0 || o_stride < 3 || d_stride < 3) {
        set_error("nerf_pack_rays: invalid argument");
        return NERF_E_INVALID;
    }
    if (n == 0) return NERF_OK;
    if (!rays_o || !rays_d || !rays) {
        set_error("nerf_pack_rays: NULL array");
        return NERF_E_INVALID;
    }
    if (cam->ndc && (cam->H <= 0 || cam->W <= 0 || !(cam->ndc_focal > 0.0))) {
        set_error("nerf_pack_rays: the NDC warp needs H, W and the focal length");
        return NERF_E_INVALID;
    }
    DeviceGuard g(c->device);
    HIP_TRY(launch_pack_rays(*cam, rays_o, o_stride, rays_d, d_stride, n, rays, (hipStream_t)stream));
    return NERF_OK;
}

int nerf_render_frame(nerf_ctx* c, const nerf_frame_args* f) {
    if (!c || !f || f->first_pixel < 0 || f->n_pixels < 0) {
        set_error("nerf_render_frame: invalid argument");
        return NERF_E_INVALID;
    }
    if (f->cam.H <= 0 || f->cam.W <= 0 || f->first_pixel + f->n_pixels > (int64_t)f->cam.H * f->cam.W) {
        set_error("nerf_render_frame: pixels [%lld, %lld) outside the %dx%d image", (long long)f->first_pixel,
                  (long long)(f->first_pixel + f->n_pixels), f->cam.H, f->cam.W);
        return NERF_E_INVALID;
    }
    if (f->n_pixels == 0) return NERF_OK;
    const int64_t chunk = f->chunk > 0 ? f->chunk : 32768;
    const int ld = f->cam.use_viewdirs ? 11 : 8;
    const int64_t per = chunk < f->n_pixels ? chunk : f->n_pixels;
    DeviceGuard g(c->device);
    hipStream_t s = (hipStream_t)f->stream;
    ScratchScope scope(c, s);
    HIP_TRY(scope.status);
    if ((size_t)per * ld > c->frame_rays_floats) {
        if (c->frame_rays) {
            HIP_TRY(hipDeviceSynchronize());
            HIP_TRY(hipFree(c->frame_rays));
            c->frame_rays = nullptr;
            c->frame_rays_floats = 0;
        }
        HIP_TRY(hipMalloc((void**)&c->frame_rays, (size_t)per * ld * sizeof(float)));
        c->frame_rays_floats = (size_t)per * ld;
    }
    if (f->precision_guard != NERF_GUARD_OFF && f->precision_guard != NERF_GUARD_REPORT &&
        f->precision_guard != NERF_GUARD_FALLBACK) {
        set_error("nerf_render_frame: precision_guard %d is not a NERF_GUARD_* value", f->precision_guard);
        return NERF_E_INVALID;
    }
    auto body = [&]() -> int {
    for (int64_t off = 0; off < f->n_pixels; off += chunk) {
        const int64_t n = off + chunk <= f->n_pixels ? chunk : f->n_pixels - off;
        // stream order makes reusing the one ray buffer safe: chunk k+1's generation runs after chunk k's kernels
        HIP_TRY(launch_raygen(f->cam, f->first_pixel + off, n, c->frame_rays, s));
        nerf_render_args r;
        memset(&r, 0, sizeof(r));
        r.rays = c->frame_rays;
        r.n_rays = n;
        r.ray_stride = ld;
        r.N_samples = f->N_samples;
        r.N_importance = f->N_importance;
        r.slot_coarse = f->slot_coarse;
        r.slot_fine = f->slot_fine;
        r.lindisp = f->lindisp;
        r.white_bkgd = f->white_bkgd;
        r.rgb_map = f->rgb_map ? f->rgb_map + off * 3 : nullptr;
        r.disp_map = f->disp_map ? f->disp_map + off : nullptr;
        r.acc_map = f->acc_map ? f->acc_map + off : nullptr;
        r.rgb0 = f->rgb0 ? f->rgb0 + off * 3 : nullptr;
        r.disp0 = f->disp0 ? f->disp0 + off : nullptr;
        r.acc0 = f->acc0 ? f->acc0 + off : nullptr;
        r.z_std = f->z_std ? f->z_std + off : nullptr;
        r.stream = f->stream;
        const int rc = render_rays_locked(c, &r);
        if (rc != NERF_OK) return rc;
    }
    return NERF_OK;
    };
    int rc = body();
    if (rc != NERF_OK) return rc;
    HIP_TRY(mirror_loose(c, s));
    if (f->precision_guard == NERF_GUARD_OFF || c->precision != NERF_PRECISION_F16X2) return NERF_OK;
    // the guard: wait for the frame, look at the counter (mirrored behind the frame's kernels)
    HIP_TRY(hipStreamSynchronize(s));
    const unsigned n_new = take_new_loose(c);
    if (n_new == 0) return NERF_OK;
    if (f->precision_guard == NERF_GUARD_REPORT) {
        set_error("nerf_render_frame: the fp16-pair kernel's output-scale bound was loose in %u (wavefront, layer) cases "
                  "of this frame: some activations kept fewer than 24 bits with these weights (NERF_PRECISION_F32 renders "
                  "them as the reference does)", n_new);
        return NERF_W_PRECISION;
    }
    c->precision = NERF_PRECISION_F32;
    rc = body();
    c->precision = NERF_PRECISION_F16X2;
    if (rc != NERF_OK) return rc;
    set_error("nerf_render_frame: the fp16-pair kernel's output-scale bound was loose in %u (wavefront, layer) cases of "
              "this frame; the range was rendered again with the fp32 kernel", n_new);
    return NERF_W_PRECISION_FALLBACK;
}

int nerf_shard_bounds(int64_t n_total, int world, int rank, int64_t* first_pixel, int64_t* n_pixels) {
    if (n_total < 0 || world <= 0 || rank < 0 || rank >= world || !first_pixel || !n_pixels) {
        set_error("nerf_shard_bounds: invalid argument (n_total %lld, world %d, rank %d)", (long long)n_total, world, rank);
        return NERF_E_INVALID;
    }
    const int64_t base = n_total / world, rem = n_total % world;
    *first_pixel = rank * base + (rank < rem ? rank : rem);
    *n_pixels = base + (rank < rem ? 1 : 0);
    return NERF_OK;
}

int nerf_render_shard(nerf_ctx* c, const nerf_frame_args* f, int world, int rank, int64_t* first_pixel,
                      int64_t* n_pixels) {
    if (!c || !f || f->cam.H <= 0 || f->cam.W <= 0) {
        set_error("nerf_render_shard: invalid argument");
        return NERF_E_INVALID;
    }
    int64_t lo = 0, n = 0;
    const int rc = nerf_shard_bounds((int64_t)f->cam.H * f->cam.W, world, rank, &lo, &n);
    if (rc != NERF_OK) return rc;
    if (first_pixel) *first_pixel = lo;
    if (n_pixels) *n_pixels = n;
    nerf_frame_args shard = *f;
    shard.first_pixel = lo;
    shard.n_pixels = n;
    return nerf_render_frame(c, &shard);
}

int nerf_image_metrics(nerf_ctx* c, const float* img1, const float* img2, int H, int W, float max_val, float* out,
                       void* stream) {
    if (!c || !img1 || !img2 || !out || H <= 0 || W <= 0) {
        set_error("nerf_image_metrics: invalid argument");
        return NERF_E_INVALID;
    }
    DeviceGuard g(c->device);
    ScratchScope scope(c, (hipStream_t)stream);
    HIP_TRY(scope.status);
    const size_t total = (size_t)H * W * 3;
    const size_t blocks = (total + 255) / 256;
    int rc = ensure_workspace(c, arena_bytes({5 * total, 4 * blocks + 16}));
    if (rc != NERF_OK) return rc;
    Arena ar(c->ws);
    float* tmp = ar.take(5 * total);
    double* partial = (double*)ar.take(4 * blocks + 16);
    HIP_TRY(launch_image_metrics(img1, img2, H, W, max_val, tmp, partial, out, (hipStream_t)stream));
    return NERF_OK;
}

int nerf_profile_enable(nerf_ctx* c, int on) {
    if (!c) return NERF_E_INVALID;
    c->profiling = on != 0;
    return NERF_OK;
}

int nerf_profile_read(nerf_ctx* c, double* mlp_ms, int64_t* launches, int64_t* points, int reset) {
    if (!c) return NERF_E_INVALID;
    DeviceGuard g(c->device);
    for (auto& p : c->events) {
        HIP_TRY(hipEventSynchronize(p.second));
        float ms = 0.0f;
        HIP_TRY(hipEventElapsedTime(&ms, p.first, p.second));
        c->prof_ms += ms;
        c->prof_launches += 1;
        c->pool.push_back(p.first);
        c->pool.push_back(p.second);
    }
    c->events.clear();
    if (mlp_ms) *mlp_ms = c->prof_ms;
    if (launches) *launches = c->prof_launches;
    if (points) *points = c->prof_points;
    if (reset) {
        c->prof_ms = 0.0;
        c->prof_launches = 0;
        c->prof_points = 0;
    }
    return NERF_OK;
}

int nerf_profile_read_train(nerf_ctx* c, double* ms, int64_t* launches, int64_t* points, int reset) {
    if (!c) return NERF_E_INVALID;
    DeviceGuard g(c->device);
    for (auto& sp : c->train_spans) {
        HIP_TRY(hipEventSynchronize(sp.e1));
        float t = 0.0f;
        HIP_TRY(hipEventElapsedTime(&t, sp.e0, sp.e1));
        if (sp.kind >= 0 && sp.kind < 4) {
            c->train_ms[sp.kind] += t;
            c->train_launches[sp.kind] += 1;
            c->train_points[sp.kind] += sp.points;
        }
        c->pool.push_back(sp.e0);
        c->pool.push_back(sp.e1);
    }
    c->train_spans.clear();
    for (int k = 0; k < 4; ++k) {
        if (ms) ms[k] = c->train_ms[k];
        if (launches) launches[k] = c->train_launches[k];
        if (points) points[k] = c->train_points[k];
        if (reset) {
            c->train_ms[k] = 0.0;
            c->train_launches[k] = 0;
            c->train_points[k] = 0;
        }
    }
    return NERF_OK;
}

int64_t nerf_workspace_bytes(nerf_ctx* c) { return c ? (int64_t)c->ws_bytes : 0; }

int nerf_precision_status(nerf_ctx* c, int64_t* loose_bound_events, int reset) {
    if (!c || !loose_bound_events) {
        set_error("nerf_precision_status: NULL argument");
        return NERF_E_INVALID;
    }
    DeviceGuard g(c->device);
    unsigned v[kLooseWords] = {};
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(v, c->d_loose, sizeof(v), hipMemcpyDeviceToHost));
    if (reset) {
        HIP_TRY(hipMemset(c->d_loose, 0, kLooseWords * sizeof(unsigned)));
        if (c->h_loose) memset(c->h_loose, 0, kLooseWords * sizeof(unsigned));
        c->loose_seen = c->train_loose_seen = 0u;
    }
    *loose_bound_events = (int64_t)v[0] + (int64_t)v[kLooseTrain];      // rendering's and the training step's
    return NERF_OK;
}

int nerf_precision_detail(nerf_ctx* c, int64_t* counts, int reset) {
    if (!c || !counts) {
        set_error("nerf_precision_detail: NULL argument");
        return NERF_E_INVALID;
    }
    DeviceGuard g(c->device);
    unsigned v[kLooseWords] = {};
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(v, c->d_loose, sizeof(v), hipMemcpyDeviceToHost));
    if (reset) {
        HIP_TRY(hipMemset(c->d_loose, 0, sizeof(v)));
        if (c->h_loose) memset(c->h_loose, 0, kLooseWords * sizeof(unsigned));
        c->loose_seen = c->train_loose_seen = 0u;
    }
    for (int i = 0; i < kLooseTrain; ++i) counts[i] = (int64_t)v[i] + (int64_t)v[kLooseTrain + i];
    return NERF_OK;
}

int nerf_precision_peek(nerf_ctx* c, int64_t* new_events) {
    if (!c || !new_events) {
        set_error("nerf_precision_peek: NULL argument");
        return NERF_E_INVALID;
    }
    *new_events = (int64_t)take_new_loose(c);
    return NERF_OK;
}

int nerf_precision_check(nerf_ctx* c, void* stream, int64_t* new_events) {
    if (!c || !new_events) {
        set_error("nerf_precision_check: NULL argument");
        return NERF_E_INVALID;
    }
    DeviceGuard g(c->device);
    HIP_TRY(mirror_loose(c, (hipStream_t)stream));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    *new_events = (int64_t)take_new_loose(c);
    return NERF_OK;
}

}  // extern "C"
