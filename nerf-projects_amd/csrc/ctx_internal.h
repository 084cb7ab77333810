// Context definition and small host helpers shared by api.cpp and train_api.cpp.
#pragma once
#include <cstddef>
#include <initializer_list>
#include <mutex>
#include <utility>
#include <vector>

#include "nerf_internal.h"

#define HIP_TRY(expr)                                                                        \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess) {                                                              \
            nerf::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return NERF_E_HIP;                                                               \
        }                                                                                    \
    } while (0)

struct nerf_ctx {
    int device = 0;
    int precision = NERF_PRECISION_F16X2;   // arithmetic of the fused MLP kernel in the RENDERING calls (nerf_set_precision,
                                            // nerf_set_render_precision)
    int train_precision = NERF_PRECISION_F16X2;   // ... in nerf_train_step (nerf_set_precision only)
    nerf::PackedNet nets[NERF_NUM_SLOTS];
    unsigned* d_loose = nullptr;   // see nerf_precision_status
    // The precision guard (nerf_mi355x.h, "Precision guard"): a pinned host mirror of d_loose, refreshed by a 64-byte copy
    // enqueued behind every render / training call, so that a later call can see - without synchronising - whether the
    // fp16-pair kernel's scale bound was loose in work that has completed. Rendering counts in word 0, the training step in
    // word kLooseTrain, and each has its own cursor of what has been reported (`loose_seen`, `train_loose_seen`): a frame
    // rendered after a loose training step does not take that step's events, nor the other way round.
    unsigned* h_loose = nullptr;
    unsigned* h_loose_dev = nullptr;   // the mirror's address as the device sees it (kernels may write it: refresh_kernels.hip)
    unsigned loose_seen = 0;
    unsigned train_loose_seen = 0;
    bool train_force_f32 = false;  // set by nerf_train_step when it sees new events: training continues on the fp32 kernels
    char* ws = nullptr;          // workspace arena
    size_t ws_bytes = 0;
    float* frame_rays = nullptr;   // ray record of the chunk being rendered by nerf_render_frame
    size_t frame_rays_floats = 0;
    // The scratch above is shared by every call on this context (nerf_render_rays, nerf_render_frame, nerf_train_step,
    // nerf_image_metrics). Calls are asynchronous, so two calls may only reuse it in stream order: ScratchScope
    // (below) serialises host threads with `scratch_mutex` and, when a call arrives on another stream than the one
    // before it, makes that stream wait for `scratch_done`, recorded when the previous call finished enqueueing.
    std::mutex scratch_mutex;
    hipEvent_t scratch_done = nullptr;
    hipStream_t scratch_stream = nullptr;
    bool scratch_used = false;
    bool profiling = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events;   // one pair per MLP launch
    std::vector<hipEvent_t> pool;
    int64_t prof_points = 0;
    double prof_ms = 0.0;
    int64_t prof_launches = 0;
    // the training step's kernels while profiling is on (nerf_profile_read_train): kind 0 forward pass, 1 backward-data pass,
    // 2 hidden-width weight gradients, 3 the other weight gradients
    struct TrainSpan {
        int kind;
        hipEvent_t e0, e1;
        int64_t points;
    };
    std::vector<TrainSpan> train_spans;
    double train_ms[4] = {};
    int64_t train_points[4] = {}, train_launches[4] = {};
};

namespace nerf {

hipError_t mirror_loose(nerf_ctx* c, hipStream_t s);     // api.cpp: the precision guard's counter mirror
unsigned take_new_loose(nerf_ctx* c);                    // rendering's events not yet reported; marks them reported
unsigned take_new_loose_train(nerf_ctx* c);              // the training step's

// HIP events around a stretch of a training step's launches (only while nerf_profile_enable is on)
struct TrainTimer {
    nerf_ctx* c;
    hipStream_t s;
    hipEvent_t e0 = nullptr;
    int kind;
    int64_t points;
    TrainTimer(nerf_ctx* ctx, hipStream_t stream, int k, int64_t pts) : c(ctx), s(stream), kind(k), points(pts) {
        if (!c->profiling) return;
        if (!c->pool.empty()) {
            e0 = c->pool.back();
            c->pool.pop_back();
        } else if (hipEventCreate(&e0) != hipSuccess) {
            e0 = nullptr;
            return;
        }
        (void)hipEventRecord(e0, s);
    }
    ~TrainTimer() {
        if (!e0) return;
        hipEvent_t e1 = nullptr;
        if (!c->pool.empty()) {
            e1 = c->pool.back();
            c->pool.pop_back();
        } else if (hipEventCreate(&e1) != hipSuccess) {
            c->pool.push_back(e0);
            return;
        }
        (void)hipEventRecord(e1, s);
        c->train_spans.push_back({kind, e0, e1, points});
    }
    TrainTimer(const TrainTimer&) = delete;
    TrainTimer& operator=(const TrainTimer&) = delete;
};

struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) ok = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

// Ownership of the context's scratch for the duration of one API call (see nerf_ctx::scratch_mutex). Recursive use
// (nerf_render_frame -> nerf_render_rays) is handled by the caller passing its own scope down: only the outermost call
// creates one.
struct ScratchScope {
    nerf_ctx* c;
    hipStream_t s;
    hipError_t status = hipSuccess;
    ScratchScope(nerf_ctx* ctx, hipStream_t stream) : c(ctx), s(stream) {
        c->scratch_mutex.lock();
        if (!c->scratch_done) status = hipEventCreateWithFlags(&c->scratch_done, hipEventDisableTiming);
        if (status == hipSuccess && c->scratch_used && c->scratch_stream != s)
            status = hipStreamWaitEvent(s, c->scratch_done, 0);
    }
    ~ScratchScope() {
        if (c->scratch_done && hipEventRecord(c->scratch_done, s) == hipSuccess) {
            c->scratch_stream = s;
            c->scratch_used = true;
        }
        c->scratch_mutex.unlock();
    }
    ScratchScope(const ScratchScope&) = delete;
    ScratchScope& operator=(const ScratchScope&) = delete;
};

inline int ensure_workspace(nerf_ctx* c, size_t bytes) {
    if (bytes <= c->ws_bytes) return NERF_OK;
    // growing is rare (first call at a given chunk size); it synchronises the device
    if (c->ws) {
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipFree(c->ws));
        c->ws = nullptr;
        c->ws_bytes = 0;
    }
    const size_t want = bytes + bytes / 8;
    hipError_t e = hipMalloc((void**)&c->ws, want);
    if (e != hipSuccess) {
        set_error("hipMalloc(%zu) for the render workspace failed: %s", want, hipGetErrorString(e));
        return NERF_E_NOMEM;
    }
    c->ws_bytes = want;
    return NERF_OK;
}

struct Arena {
    char* base;
    size_t off = 0;
    explicit Arena(char* b) : base(b) {}
    float* take(size_t n_floats) {
        float* p = (float*)(base + off);
        off += (n_floats * sizeof(float) + 255) & ~(size_t)255;
        return p;
    }
};
inline size_t arena_bytes(std::initializer_list<size_t> float_counts) {
    size_t t = 0;
    for (size_t n : float_counts) t += (n * sizeof(float) + 255) & ~(size_t)255;
    return t;
}

}  // namespace nerf
