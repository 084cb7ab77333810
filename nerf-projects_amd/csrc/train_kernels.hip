// Training-step kernels: everything loss.backward() + optimizer.step() need
// around the reference's training iteration (nerf.ipynb:1258-1282), for gfx950.
//
// Training keeps every layer's activations (autograd does too), so here the MLP is evaluated layer
// by layer on explicit row-major [points, channels] buffers with two fp32-MFMA GEMM kernels:
//   gemm_rows : C[M,N] = A[M,K] * B[K,N] (+bias, ReLU, ReLU-mask, accumulate), M = points
//               forward  Y = X W^T + b   (B = W^T, kept transposed by the optimizer step)
//               backward dX = dY W       (B = W as nn.Linear stores it)
//   gemm_tn   : C[Mo,No] = sum_p A[p,Mo] * B[p,No]   (dW = dY^T X, db = dY^T 1), split over points
//               into per-slice partials that a second kernel adds in a fixed order (deterministic)
// Both use v_mfma_f32_32x32x2_f32 (exact fp32). At 131 kFLOP per point per 256x256 layer against 2 KB
// of activation traffic the GEMMs stay MFMA-bound; the inference path keeps its fused kernel.
// Plus: encode-into-concat-buffers, compositing backward (wavefront suffix scans), MSE loss + gradient,
// Adam, weight transposes.
#include <math.h>

#include "nerf_internal.h"
#include "ray_device.h"

namespace nerf {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// ---------------------------------------------------------------------------------------------
// Both GEMMs keep their LDS tiles K-contiguous per row ([row][32 k + 4 pad]) so that an MFMA operand
// fragment - 16 k-steps of one row - is four ds_read_b128 (conflict-free: bank = 36*row mod 64), and they
// software-pipeline those reads by hand like the inference kernel: the fragment of step n+1 is requested
// right after the first MFMA of step n (one wave per SIMD has nothing else to hide LDS latency behind).
// k-step t of an MFMA contracts index 16*h + t of the tile for lane half h; A and B use the same order.
// ---------------------------------------------------------------------------------------------
constexpr int kLd = 36;

struct F16 {
    f32x4 q[4];
};
__device__ __forceinline__ F16 frag_at(const float* tile, int row, int h) {
    F16 f;
#pragma unroll
    for (int q = 0; q < 4; ++q) f.q[q] = *(const f32x4*)&tile[row * kLd + 16 * h + 4 * q];
    return f;
}
template <int LO, int HI>
__device__ __forceinline__ void mma_frag(f32x16& acc, const F16& a, const F16& b) {
#pragma unroll
    for (int t = LO; t < HI; ++t) acc = mfma32(a.q[t >> 2][t & 3], b.q[t >> 2][t & 3], acc);
}
// The order of a step is pinned with scheduling fences: its first MFMA, the four reads of the next step's fragment
// into a second register set, then the other 15 MFMAs. Left to itself hipcc reads into the registers the MFMAs still
// use, i.e. after them, and waits for the first quad at the head of every step.
#define FENCE() __builtin_amdgcn_sched_barrier(0)

// The loaders' barrier: LDS writes visible (lgkmcnt(0)), then s_barrier - and nothing else. __syncthreads() also waits for
// vmcnt(0), i.e. for the global loads of the NEXT tile that a loader has just requested: with it every k-tile cost its
// load time PLUS its MFMA time (timing-only ablations of round 2: 31 + 108 + 66 = 205 us per launch, the parts adding up
// exactly), the prefetch distance notwithstanding. The data registers of those loads are tracked by the compiler as usual:
// it waits for them where deposit() reads them.
__device__ __forceinline__ void loader_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// ---------------------------------------------------------------------------------------------
// C[M,N] = A[M,K] * B[K,N]      (N <= 256, any K; one workgroup = 128 rows x all N columns)
//
// Eight waves per workgroup with fixed roles (two per SIMD): waves 0-3 are MFMA consumers (32 rows x 256
// columns each, 128 accumulator registers), waves 4-7 are loaders that stage the NEXT k-tile (global ->
// registers -> LDS, transposing B so that both tiles are K-contiguous) into the other half of a double
// buffer while the consumers run the 128 MFMAs of the current one. One barrier per k-tile. The loaders'
// predicated dword loads and address arithmetic run on the vector/memory pipes beside the consumers'
// matrix pipe instead of in front of it.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void gemm_rows_kernel(GemmRows g) {
    extern __shared__ __attribute__((aligned(16))) float lds_rows[];
    float* As = lds_rows;                       // [2][128 * kLd]   rows of A, K-contiguous
    float* Bs = lds_rows + 2 * 128 * kLd;       // [2][256 * kLd]   columns of B, K-contiguous
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, j = lane & 31;
    const bool consumer = wave < 4;             // wave-uniform
    const int ptid = tid & 255;
    const int64_t row0 = (int64_t)blockIdx.x * 128;
    const int n_ct = (g.N + 31) >> 5;           // column tiles in use (<= 8), uniform
    const int n_kt = (g.K + 31) >> 5;

    // The two roles run separate loops with the same barrier count (s_barrier counts arrivals, not program points), so
    // that each role's registers are allocated on their own. A loader requests k-tile kt + 2 right after it has written
    // k-tile kt + 1 to LDS: a tile's global loads have a whole consumer period (128 MFMAs) to arrive before they are
    // needed, instead of sitting between two barriers with the conversion to LDS behind them.
    if (!consumer) {
        // (one register set here: the two-set form of gemm_tn_kernel needs 96 staging registers next to this kernel's 256-
        // register consumers and spills; this kernel is the fallback path since the fused forward / backward kernels)
        float ra[16], rb[32];
        auto request = [&](int kt) {
            const int k0 = kt * 32;
#pragma unroll
            for (int it = 0; it < 16; ++it) {
                const int e = it * 256 + ptid, r = e >> 5, c = e & 31;
                const int64_t gr = row0 + r;
                ra[it] = (gr < g.M && k0 + c < g.K) ? g.A[gr * g.lda + k0 + c] : 0.0f;
            }
#pragma unroll
            for (int it = 0; it < 32; ++it)
                rb[it] = (k0 + it < g.K && ptid < g.N) ? g.B[(int64_t)(k0 + it) * g.ldb + ptid] : 0.0f;
        };
        auto deposit = [&](int buf) {
            float* as = As + buf * 128 * kLd;
            float* bs = Bs + buf * 256 * kLd;
#pragma unroll
            for (int it = 0; it < 16; ++it) {
                const int e = it * 256 + ptid;
                as[(e >> 5) * kLd + (e & 31)] = ra[it];
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {           // thread = column n: its 32 k values, 16 bytes at a time
                f32x4 v = {rb[4 * q], rb[4 * q + 1], rb[4 * q + 2], rb[4 * q + 3]};
                *(f32x4*)&bs[ptid * kLd + 4 * q] = v;
            }
        };
        request(0);
        deposit(0);
        if (n_kt > 1) request(1);
        loader_barrier();
        for (int kt = 0; kt < n_kt; ++kt) {
            if (kt + 1 < n_kt) {
                deposit((kt + 1) & 1);
                if (kt + 2 < n_kt) request(kt + 2);
            }
            loader_barrier();
        }
        return;
    }

    f32x16 acc[8];
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.0f;
    __syncthreads();
    for (int kt = 0; kt < n_kt; ++kt) {
        {
            const float* as = As + (kt & 1) * 128 * kLd;
            const float* bs = Bs + (kt & 1) * 256 * kLd;
            const F16 a = frag_at(as, 32 * wave + j, h);
            F16 cur = frag_at(bs, j, h);
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                if (c < n_ct) {
                    mma_frag<0, 1>(acc[c], a, cur);
                    FENCE();
                    F16 nxt = cur;
                    if (c + 1 < n_ct) nxt = frag_at(bs, 32 * (c + 1) + j, h);
                    FENCE();
                    mma_frag<1, 16>(acc[c], a, cur);
                    FENCE();
                    cur = nxt;
                }
            }
        }
        __syncthreads();
    }
    // D[row][col]: col = lane & 31 (-> n), row = (reg & 3) + 8*(reg >> 2) + 4*h (-> point).
    // The optional reads (old C for accumulate, the ReLU mask) are batched per column tile from clamped
    // addresses under wave-uniform flags, so 16 loads are in flight at a time instead of one.
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        if (c >= n_ct) continue;
        const int n = 32 * c + j;
        const bool n_ok = n < g.N;
        const int ncl = n_ok ? n : g.N - 1;
        const float bias = g.bias ? g.bias[ncl] : 0.0f;
        int64_t rowv[16];
        float oldv[16], maskv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int64_t row = row0 + 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * h;
            rowv[r] = row < g.M ? row : g.M - 1;
        }
        if (g.accumulate) {
#pragma unroll
            for (int r = 0; r < 16; ++r) oldv[r] = g.C[rowv[r] * g.ldc + ncl];
        }
        if (g.mask) {
#pragma unroll
            for (int r = 0; r < 16; ++r) maskv[r] = g.mask[rowv[r] * g.ldm + ncl];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int64_t row = row0 + 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * h;
            float v = acc[c][r];
            if (g.accumulate) v += oldv[r];
            if (g.bias) v += bias;
            if (g.relu) v = fmaxf(v, 0.0f);
            if (g.mask && !(maskv[r] > 0.0f)) v = 0.0f;   // ReLU'(pre) = [post > 0]
            if (n_ok && row < g.M) g.C[row * g.ldc + n] = v;
        }
    }
}

constexpr size_t kRowsLds = (size_t)(2 * 128 * kLd + 2 * 256 * kLd) * sizeof(float);   // 108 KiB
constexpr size_t kTnLds = (size_t)(2 * 256 * kLd + 2 * 128 * kLd) * sizeof(float);

static hipError_t raise_lds(const void* fn, size_t bytes, bool* done) {
    if (*done) return hipSuccess;
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess) *done = true;
    return e;
}

hipError_t launch_gemm_rows(const GemmRows& g, hipStream_t s) {
    if (g.M <= 0 || g.N <= 0) return hipSuccess;
    if (g.N > 256) return hipErrorInvalidValue;
    static bool raised[64] = {};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 64 && (e = raise_lds((const void*)gemm_rows_kernel, kRowsLds, &raised[dev])) != hipSuccess) return e;
    hipLaunchKernelGGL(gemm_rows_kernel, dim3((unsigned)((g.M + 127) / 128)), dim3(512), kRowsLds, s, g);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// part[slice][Mo, No] = sum over the slice's points of A[p, Mo]^T B[p, No]   (dW = dY^T X)
// dbp[slice][Mo]      = sum over the slice's points of A[p, Mo]               (db = dY^T 1)
// One workgroup = all Mo (<= 256) rows x 128 columns; the contraction index is the point, so both tiles
// are transposed while staging: As[m][p], Bs[n][p]. Same loader / consumer split as gemm_rows; the bias
// gradient costs nothing: the loader thread that stages column m of dY keeps its running sum in a register.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void gemm_tn_kernel(GemmTN g) {
    extern __shared__ __attribute__((aligned(16))) float lds_tn[];
    float* As = lds_tn;                         // [2][256 * kLd]
    float* Bs = lds_tn + 2 * 256 * kLd;         // [2][128 * kLd]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, j = lane & 31;
    const bool consumer = wave < 4;
    const int ptid = tid & 255;
    const int slice = blockIdx.x;
    const int n0 = blockIdx.y * 128;
    const int no_eff = g.No;
    const int64_t p_begin = (int64_t)slice * g.pts_per_slice;
    int64_t p_end = p_begin + g.pts_per_slice;
    if (p_end > g.P) p_end = g.P;
    const int n_pt = p_end > p_begin ? (int)((p_end - p_begin + 31) / 32) : 0;
    const int m_tiles = (g.Mo + 31) >> 5;       // <= 8
    const int mt0 = 2 * (wave & 3), mt1 = mt0 + 1;

    const int bn = ptid & 127, bp0 = (ptid >> 7) * 16;   // B staging: thread = (column, half of the 32 points)
    float colsum = 0.0f;                        // loaders: sum over this slice's points of A[:, ptid]

    // Like gemm_rows, the two roles run separate loops with the same barrier count. A loader requests point-tile t + 2
    // right after it has written tile t + 1 to LDS, so a tile's global loads have a whole consumer period (128 MFMAs)
    // to arrive before they are needed instead of sitting, with the LDS writes behind them, inside one period.
    if (!consumer) {
        // Two register sets, tile k in set k & 1, and the loop unrolled by two: with ONE set carried around the loop
        // hipcc copies the freshly loaded registers at the back edge, i.e. waits for the loads it has just issued, and
        // the prefetch distance is gone (the kernel then costs its load time plus its MFMA time; round-2 ablations).
        float ra0[32], rb0[16], ra1[32], rb1[16];
        auto request = [&](int t, float (&ra)[32], float (&rb)[16]) {
            const int64_t p0 = p_begin + (int64_t)t * 32;
#pragma unroll
            for (int it = 0; it < 32; ++it) {
                const int64_t p = p0 + it;
#ifdef NERF_ABLATE_TN_LOADS
                ra[it] = (float)(p & 7);
#else
                ra[it] = (p < p_end && ptid < g.Mo) ? g.A[p * g.lda + ptid] : 0.0f;
#endif
            }
#pragma unroll
            for (int it = 0; it < 16; ++it) {
                const int64_t p = p0 + bp0 + it;
                const int n = n0 + bn;
#ifdef NERF_ABLATE_TN_LOADS
                rb[it] = (float)((p + n) & 7);
#else
                rb[it] = (p < p_end && n < g.No) ? g.B[p * g.ldb + n] : 0.0f;
#endif
            }
        };
        auto deposit = [&](int buf, const float (&ra)[32], const float (&rb)[16]) {
            float* as = As + buf * 256 * kLd;
            float* bs = Bs + buf * 128 * kLd;
#pragma unroll
            for (int it = 0; it < 32; ++it) colsum += ra[it];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                f32x4 v = {ra[4 * q], ra[4 * q + 1], ra[4 * q + 2], ra[4 * q + 3]};
                *(f32x4*)&as[ptid * kLd + 4 * q] = v;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 v = {rb[4 * q], rb[4 * q + 1], rb[4 * q + 2], rb[4 * q + 3]};
                *(f32x4*)&bs[bn * kLd + bp0 + 4 * q] = v;
            }
        };
        if (n_pt > 0) {
            request(0, ra0, rb0);
            deposit(0, ra0, rb0);
            if (n_pt > 1) request(1, ra1, rb1);
        }
        loader_barrier();
        for (int t = 0; t < n_pt; t += 2) {
            // period t: tile t + 1 (set 1) to buffer 1, tile t + 2 requested into set 0
            if (t + 1 < n_pt) {
                deposit(1, ra1, rb1);
                if (t + 2 < n_pt) request(t + 2, ra0, rb0);
            }
            loader_barrier();
            if (t + 1 >= n_pt) break;
            // period t + 1: tile t + 2 (set 0) to buffer 0, tile t + 3 requested into set 1
            if (t + 2 < n_pt) {
                deposit(0, ra0, rb0);
                if (t + 3 < n_pt) request(t + 3, ra1, rb1);
            }
            loader_barrier();
        }
        if (g.dbp && blockIdx.y == 0 && ptid < g.Mo) g.dbp[(int64_t)slice * g.Mo + ptid] = colsum;
        return;
    }

    f32x16 acc[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][c][r] = 0.0f;
    __syncthreads();
    for (int t = 0; t < n_pt; ++t) {
        const float* as = As + (t & 1) * 256 * kLd;
        const float* bs = Bs + (t & 1) * 128 * kLd;
        const F16 a0 = frag_at(as, 32 * mt0 + j, h), a1 = frag_at(as, 32 * mt1 + j, h);
        F16 cur = frag_at(bs, j, h);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            mma_frag<0, 1>(acc[0][c], a0, cur);
            FENCE();
            F16 nxt = cur;
            if (c + 1 < 4) nxt = frag_at(bs, 32 * (c + 1) + j, h);
            FENCE();
#ifndef NERF_ABLATE_TN_MFMA
            mma_frag<1, 16>(acc[0][c], a0, cur);
            mma_frag<0, 16>(acc[1][c], a1, cur);
#endif
            FENCE();
            cur = nxt;
        }
        __syncthreads();
    }
    float* part = g.part + (int64_t)slice * g.Mo * no_eff;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        const int mt = a ? mt1 : mt0;
        if (mt >= m_tiles) continue;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int n = n0 + 32 * c + j;
            if (n >= no_eff) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (m < g.Mo) part[(int64_t)m * no_eff + n] = acc[a][c][r];
            }
        }
    }
}

// dW[m][n] = sum_s part[s][m][n] and db[m] = sum_s dbp[s][m], slices added in order (deterministic)
__global__ void reduce_slices_kernel(const float* __restrict__ part, const float* __restrict__ dbp, int n_slices,
                                     int Mo, int No, float* __restrict__ dW, int ldw, float* __restrict__ db,
                                     int accumulate, const GradExps ex) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t n_w = (int64_t)Mo * No;
    if (idx < n_w) {
        // eight loads are in flight at a time - with one dependent load per iteration the partials streamed at a quarter of
        // what the memory system gives
        float s = 0.0f;
        int k = 0;
        for (; k + 8 <= n_slices; k += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = part[(int64_t)(k + u) * n_w + idx];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; k < n_slices; ++k) s += part[(int64_t)k * n_w + idx];
        const int m = (int)(idx / No), n = (int)(idx % No);
        s = __builtin_ldexpf(s, ex.of(m, n));      // GradJob: equalised units
        float* w = dW + (int64_t)m * ldw + n;
        *w = accumulate ? *w + s : s;   // second pass through a shared network: .grad accumulates (nerf.ipynb:1270)
    } else if (idx < n_w + Mo && db && dbp) {
        const int m = (int)(idx - n_w);
        float s = 0.0f;
        for (int k = 0; k < n_slices; ++k) s += dbp[(int64_t)k * Mo + m];
        s = __builtin_ldexpf(s, ex.of_row(m));
        db[m] = accumulate ? db[m] + s : s;
    }
}

// the same sum over MANY slices of a small matrix (gemm_tn_small_kernel: up to 2048 slices of <= 4 x 320): a workgroup owns 16
// consecutive elements and splits the slices 16 ways (eight loads in flight per thread); the 16 group sums are added in
// group order through LDS. With one thread per element walking all 2048 slices the pass took 121 us.
__global__ __launch_bounds__(256) void reduce_many_slices_kernel(const float* __restrict__ part, const float* __restrict__ dbp,
                                                                 int n_slices, int Mo, int No, float* __restrict__ dW, int ldw,
                                                                 float* __restrict__ db, int accumulate, const GradExps ex) {
    __shared__ float red[16][16];
    const int64_t n_w = (int64_t)Mo * No, n_all = n_w + Mo;       // weights, then the bias entries (from dbp)
    const int e = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const int64_t idx = (int64_t)blockIdx.x * 16 + e;
    float sum = 0.0f;
    if (idx < n_all && (idx < n_w || (db && dbp))) {
        const float* src = idx < n_w ? part + idx : dbp + (idx - n_w);
        const int64_t stride = idx < n_w ? n_w : Mo;
        int k = grp;
        for (; k + 7 * 16 < n_slices; k += 8 * 16) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = src[(int64_t)(k + 16 * u) * stride];
#pragma unroll
            for (int u = 0; u < 8; ++u) sum += v[u];
        }
        for (; k < n_slices; k += 16) sum += src[(int64_t)k * stride];
    }
    red[grp][e] = sum;
    __syncthreads();
    if (grp == 0 && idx < n_all) {
        float t = red[0][e];
#pragma unroll
        for (int g2 = 1; g2 < 16; ++g2) t += red[g2][e];
        if (idx < n_w) {
            const int m = (int)(idx / No), n = (int)(idx % No);
            t = __builtin_ldexpf(t, ex.of(m, n));
            float* w = dW + (int64_t)m * ldw + n;
            *w = accumulate ? *w + t : t;
        } else if (db && dbp) {
            const int m = (int)(idx - n_w);
            t = __builtin_ldexpf(t, ex.of_row(m));
            db[m] = accumulate ? db[m] + t : t;
        }
    }
}

// dW of a Linear with at most four output rows (rgb_linear, alpha_linear, a 4- or 5-channel output_linear goes to the
// staged kernel): a weighted column sum of X, bound by reading X once. Thread = column; the row's dY values are
// wave-uniform (scalar loads). 145 us per launch on the 256-row MFMA kernel, which computed 252 rows of zeros.
__global__ __launch_bounds__(256) void gemm_tn_small_kernel(GemmTN g) {
    const int n = blockIdx.y * 256 + threadIdx.x;
    const int slice = blockIdx.x;
    const int64_t p_begin = (int64_t)slice * g.pts_per_slice;
    int64_t p_end = p_begin + g.pts_per_slice;
    if (p_end > g.P) p_end = g.P;
    float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f}, bsum[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    const bool live = n < g.No;
    int64_t p = p_begin;
    for (; p + 8 <= p_end; p += 8) {
        float x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) x[u] = live ? g.B[(p + u) * g.ldb + n] : 0.0f;
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int m = 0; m < 4; ++m)
                if (m < g.Mo) {
                    const float a = g.A[(p + u) * g.lda + m];
                    acc[m] = fmaf(a, x[u], acc[m]);
                    bsum[m] += a;
                }
    }
    for (; p < p_end; ++p) {
        const float x = live ? g.B[p * g.ldb + n] : 0.0f;
#pragma unroll
        for (int m = 0; m < 4; ++m)
            if (m < g.Mo) {
                const float a = g.A[p * g.lda + m];
                acc[m] = fmaf(a, x, acc[m]);
                bsum[m] += a;
            }
    }
    if (live)
        for (int m = 0; m < g.Mo; ++m) g.part[((int64_t)slice * g.Mo + m) * g.No + n] = acc[m];
    if (g.dbp && blockIdx.y == 0 && threadIdx.x == 0)
        for (int m = 0; m < g.Mo; ++m) g.dbp[(int64_t)slice * g.Mo + m] = bsum[m];
}

// The same sum for No = 256 / 128 / 64 with 16-byte-aligned rows of X (what the network's own small layers are): a thread
// takes FOUR columns, so a wave's load instruction moves a KiB instead of 256 bytes and the No / 4 threads of a point leave
// room in the workgroup for 256 / (No / 4) points side by side (their sums are added through LDS, in order).
typedef float f32x4v __attribute__((ext_vector_type(4)));
// MM: rows held per thread (4: rgb_linear, alpha_linear; 8: an output_linear of up to 8 channels). g.b_blocked: X is blocked by 32
// points (MlpStore::blocked; No = 256): the four features 4 i .. 4 i + 3 of point p sit in piece i / 2 of p's group.
template <int MM>
__global__ __launch_bounds__(256) void gemm_tn_small4_kernel(GemmTN g) {
    __shared__ float red[MM * 1024];           // [point lane][m][column]: 256 / (No / 4) x MM x No floats
    __shared__ float redb[16][MM];
    const int tpp = g.No >> 2;                 // threads per point: 64, 32 or 16
    const int pl = threadIdx.x / tpp, c = 4 * (threadIdx.x % tpp), n_pl = 256 / tpp;
    const int slice = blockIdx.x;
    const int64_t p_begin = (int64_t)slice * g.pts_per_slice;
    int64_t p_end = p_begin + g.pts_per_slice;
    if (p_end > g.P) p_end = g.P;
    auto x_at = [&](int64_t p) -> const float* {
        return g.b_blocked ? g.B + (p >> 5) * (32 * (int64_t)256) + (c >> 3) * 256 + (int)(p & 31) * 8 + ((c >> 2) & 1) * 4
                           : g.B + p * g.ldb + c;
    };
    f32x4v acc[MM];
    float bsum[MM];
#pragma unroll
    for (int m = 0; m < MM; ++m) {
        acc[m] = f32x4v{0.0f, 0.0f, 0.0f, 0.0f};
        bsum[m] = 0.0f;
    }
    int64_t p = p_begin + pl;
    const int64_t step = n_pl;
    for (; p + 3 * step < p_end; p += 4 * step) {
        f32x4v x[4];
        float a[4][MM];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            x[u] = *(const f32x4v*)x_at(p + u * step);
#pragma unroll
            for (int m = 0; m < MM; ++m) a[u][m] = g.A[(p + u * step) * g.lda + (m < g.Mo ? m : 0)];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int m = 0; m < MM; ++m) {
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[m][q] = fmaf(a[u][m], x[u][q], acc[m][q]);
                bsum[m] += a[u][m];
            }
    }
    for (; p < p_end; p += step) {
        const f32x4v x = *(const f32x4v*)x_at(p);
#pragma unroll
        for (int m = 0; m < MM; ++m) {
            const float a = g.A[p * g.lda + (m < g.Mo ? m : 0)];
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[m][q] = fmaf(a, x[q], acc[m][q]);
            bsum[m] += a;
        }
    }
#pragma unroll
    for (int m = 0; m < MM; ++m) {
        *(f32x4v*)&red[(pl * MM + m) * g.No + c] = acc[m];
        if (c == 0) redb[pl][m] = bsum[m];
    }
    __syncthreads();
    for (int e = threadIdx.x; e < g.Mo * g.No; e += 256) {
        const int m = e / g.No, n = e % g.No;
        float t = red[m * g.No + n];
        for (int q = 1; q < n_pl; ++q) t += red[(q * MM + m) * g.No + n];
        g.part[((int64_t)slice * g.Mo + m) * g.No + n] = t;
    }
    if (g.dbp && threadIdx.x < g.Mo) {
        float t = redb[0][threadIdx.x];
        for (int q = 1; q < n_pl; ++q) t += redb[q][threadIdx.x];
        g.dbp[(int64_t)slice * g.Mo + threadIdx.x] = t;
    }
}

// NERF_TRAIN_STAGED_DW=1 keeps the LDS-staged kernel for every layer (A/B and fallback)
static bool staged_dw_requested() {
    static const bool on = [] {
        const char* e = getenv("NERF_TRAIN_STAGED_DW");
        return e && *e && *e != '0';
    }();
    return on;
}

// workgroups per slice of the launch that does the bulk of a layer's dW (the caller sizes the number of slices with it)
// the weighted-column-sum kernel wants many short slices (its only parallelism besides the columns)
// (five to eight rows - an output_linear of more than four channels - only against an X the 16-byte kernel reads)
bool gemm_tn_is_small(int Mo, int No) {
    return (Mo <= 4 || (Mo <= 8 && (No == 256 || No == 128 || No == 64))) && !staged_dw_requested();
}

int gemm_tn_col_blocks(int Mo, int No) {
    if (gemm_tn_is_small(Mo, No)) return (No + 255) / 256;
    if (Mo % 128 == 0 && Mo <= 256 && !staged_dw_requested()) return 1;
    return (No + 127) / 128;
}

bool gemm_tn_is_direct(int Mo) { return Mo % 128 == 0 && Mo <= 256 && !staged_dw_requested(); }

// One Linear at a time (the layer-by-layer backward pass, the small layers, the staged fallback); the fused backward pass
// batches its direct-eligible layers itself (train_api.cpp).
hipError_t launch_gemm_tn(const GemmTN& g, int n_slices, float* dW, int ldw, float* db, int accumulate, hipStream_t s) {
    if (g.Mo <= 0 || g.No <= 0) return hipSuccess;
    if (g.Mo > 256) return hipErrorInvalidValue;
    if (gemm_tn_is_direct(g.Mo)) {
        // the hidden-width columns as one job of 256 (for the skip layer they are NOT the first ones: that concatenation puts
        // gamma(x) in front; the view layer's puts gamma(d) behind), what is left as jobs of up to 64
        const int wide = g.No >= 256 ? 256 : 0;
        const int rest = g.No - wide;
        const int wide_begin = g.narrow_first ? rest : 0;
        const size_t part_floats = (size_t)n_slices * g.Mo * g.No, dbp_floats = (size_t)n_slices * g.Mo;
        GradBatch b{};
        b.n_slices = n_slices;
        b.accumulate = accumulate;
        b.P = g.P;
        b.pts_per_slice = g.pts_per_slice;
        if (wide) {
            b.n = 1;
            b.job[0] = GradJob{g.A, g.lda, g.B, g.ldb, g.Mo, wide_begin, wide_begin + wide, dW, ldw, db, nullptr, nullptr,
                               nullptr, nullptr, g.ex};
            hipError_t e = launch_grad_batch(b, true, g.part, part_floats, g.dbp, dbp_floats, s);
            if (e != hipSuccess) return e;
        }
        if (rest > 0) {
            const int nb = g.narrow_first ? 0 : wide;
            b.n = 0;
            for (int c = 0; c < rest && b.n < kMaxGradJobs; c += 64, ++b.n)
                b.job[b.n] = GradJob{g.A, g.lda, g.B, g.ldb, g.Mo, nb + c, nb + (c + 64 < rest ? c + 64 : rest), dW, ldw,
                                     (!wide && c == 0) ? db : nullptr, nullptr, nullptr, nullptr, nullptr, g.ex};
            if ((rest + 63) / 64 > kMaxGradJobs) return hipErrorInvalidValue;
            // (the wide job's partials have been reduced by now: stream order)
            return launch_grad_batch(b, false, g.part, part_floats, g.dbp, dbp_floats, s);
        }
        return hipSuccess;
    }
    if (gemm_tn_is_small(g.Mo, g.No)) {
        const bool by_four = (g.No == 256 || g.No == 128 || g.No == 64) && ((uintptr_t)g.B & 15) == 0 &&
                             (g.b_blocked ? g.No == 256 : g.ldb % 4 == 0);
        if (g.b_blocked && !by_four) return hipErrorInvalidValue;
        if (by_four && g.Mo <= 4)
            hipLaunchKernelGGL(gemm_tn_small4_kernel<4>, dim3((unsigned)n_slices), dim3(256), 0, s, g);
        else if (by_four)
            hipLaunchKernelGGL(gemm_tn_small4_kernel<8>, dim3((unsigned)n_slices), dim3(256), 0, s, g);
        else if (g.Mo <= 4)
            hipLaunchKernelGGL(gemm_tn_small_kernel, dim3((unsigned)n_slices, (unsigned)((g.No + 255) / 256)), dim3(256), 0, s, g);
        else
            return hipErrorInvalidValue;      // (more than four rows against an X that is not 16-byte aligned: no caller has one)
        const int64_t n_all = (int64_t)g.Mo * g.No + g.Mo;
        hipLaunchKernelGGL(reduce_many_slices_kernel, dim3((unsigned)((n_all + 15) / 16)), dim3(256), 0, s, g.part, g.dbp,
                           n_slices, g.Mo, g.No, dW, ldw, db, accumulate, g.ex);
        return hipGetLastError();
    } else {
        static bool raised[64] = {};
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return e;
        if (dev >= 0 && dev < 64 && (e = raise_lds((const void*)gemm_tn_kernel, kTnLds, &raised[dev])) != hipSuccess) return e;
        hipLaunchKernelGGL(gemm_tn_kernel, dim3((unsigned)n_slices, (unsigned)((g.No + 127) / 128)), dim3(512), kTnLds, s, g);
    }
    const int64_t total = (int64_t)g.Mo * g.No + g.Mo;
    hipLaunchKernelGGL(reduce_slices_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, g.part, g.dbp, n_slices,
                       g.Mo, g.No, dW, ldw, db, accumulate, g.ex);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// inputs: pts = o + d z, gamma(pts) -> x0[:, 0:in_ch] (ld ld0) and, when given, x1[:, 0:in_ch] (the skip layer's concat
// buffer), gamma(viewdir) -> vcat[:, voff: voff+in_ch_views]. A workgroup takes 64 points: its threads share the
// sincosf calls ((point, axis, frequency) items), the rows are put together in LDS and written out with consecutive
// threads on consecutive floats of a row (a thread per point wrote 63 floats 252 bytes apart from its neighbour's, and
// every destination was a launch of its own that evaluated the encoding again).
// ---------------------------------------------------------------------------------------------
constexpr int kEmbedPoints = 64;
// PROLOGUE (the training step's first launch): the depths are not read but MADE here - render_rays' stratified sampling
// (stratified_z, the expression of stratified_kernel) - and written to `z` for the kernels that follow, and workgroup 0
// zeroes the step's small accumulators (`zero`, n_zero words: the passes' running maxima, the loss kernel's ticket): one
// launch where there were four.
struct EmbedPrologue {
    const float* t_rand;     // [N, S] jitter, or nullptr
    int lindisp;
    unsigned* zero;
    int n_zero;
    int pad;                 // x0 and vcat rows are zero-padded to their strides
};
template <bool PROLOGUE>
__global__ __launch_bounds__(256) void embed_train_kernel(const float* __restrict__ rays, int ray_ld, float* __restrict__ z,
                                                          int64_t P, int S, int Lx, int Lv, float* __restrict__ x0, int ld0,
                                                          float* __restrict__ x1, int ld1, float* __restrict__ vcat, int ldv,
                                                          int voff, const EmbedPrologue pro) {
    __shared__ float sx[kEmbedPoints][64];   // 3 + 6 Lx <= 63 columns
    __shared__ float sv[kEmbedPoints][28];   // 3 + 6 Lv <= 27 columns
    __shared__ float sz[kEmbedPoints];
    const int64_t p0 = (int64_t)blockIdx.x * kEmbedPoints;
    const int n_here = (int)((P - p0) < kEmbedPoints ? (P - p0) : kEmbedPoints);
    const int fx = Lx + 1, fv = Lv + 1;      // frequency slot 0 is the identity term
    if constexpr (PROLOGUE) {
        if (blockIdx.x == 0)
            for (int i = threadIdx.x; i < pro.n_zero; i += 256) pro.zero[i] = 0u;
    }
    if ((int)threadIdx.x < n_here) {
        const int64_t pt = p0 + threadIdx.x;
        if constexpr (PROLOGUE) {
            const int64_t ray = pt / S;
            const float zz = stratified_z(rays[ray * ray_ld + 6], rays[ray * ray_ld + 7], (int)(pt - ray * S), S, pro.lindisp,
                                          pro.t_rand ? pro.t_rand + pt : nullptr);
            z[pt] = zz;
            sz[threadIdx.x] = zz;
        } else {
            sz[threadIdx.x] = z[pt];
        }
    }
    __syncthreads();
    for (int it = threadIdx.x; it < kEmbedPoints * 3 * fx; it += 256) {
        const int pl = it / (3 * fx), c = (it / fx) % 3, k = it % fx - 1;
        if (pl >= n_here) continue;
        const int64_t pt = p0 + pl;
        const float* r = rays + (pt / S) * ray_ld;
        const float p = __fadd_rn(r[c], __fmul_rn(r[3 + c], sz[pl]));   // nerf.ipynb:447
        if (k < 0) {
            sx[pl][c] = p;
        } else {
            float sn, cs;
            sincosf(p * (float)(1 << k), &sn, &cs);
            sx[pl][3 + 6 * k + c] = sn;
            sx[pl][3 + 6 * k + 3 + c] = cs;
        }
    }
    if (vcat)
        for (int it = threadIdx.x; it < kEmbedPoints * 3 * fv; it += 256) {
            const int pl = it / (3 * fv), c = (it / fv) % 3, k = it % fv - 1;
            if (pl >= n_here) continue;
            const float d = rays[((p0 + pl) / S) * ray_ld + ray_ld - 3 + c];
            if (k < 0) {
                sv[pl][c] = d;
            } else {
                float sn, cs;
                sincosf(d * (float)(1 << k), &sn, &cs);
                sv[pl][3 + 6 * k + c] = sn;
                sv[pl][3 + 6 * k + 3 + c] = cs;
            }
        }
    __syncthreads();
    const int in_ch = 3 + 6 * Lx, in_v = 3 + 6 * Lv;
    // (pro.pad: the rows of x0 and of vcat are padded with zeros to their strides - the fp16-pipe weight gradients of these
    // columns fetch whole rows of 64, train_dw_kernel.hip grad_batch_narrow_pair_kernel)
    const int w0 = pro.pad ? ld0 : in_ch, wv = pro.pad ? ldv - voff : in_v;
    for (int it = threadIdx.x; it < n_here * w0; it += 256) {
        const int pl = it / w0, col = it % w0;
        const float v = col < in_ch ? sx[pl][col] : 0.0f;
        x0[(p0 + pl) * ld0 + col] = v;
        if (x1 && col < in_ch) x1[(p0 + pl) * ld1 + col] = v;
    }
    if (vcat)
        for (int it = threadIdx.x; it < n_here * wv; it += 256) {
            const int pl = it / wv, col = it % wv;
            vcat[(p0 + pl) * ldv + voff + col] = col < in_v ? sv[pl][col] : 0.0f;
        }
}

hipError_t launch_embed_train(const float* rays, int ray_ld, const float* z, int64_t P, int S, int Lx, int Lv,
                              float* x0, int ld0, float* x1, int ld1, float* vcat, int ldv, int voff, hipStream_t s, int pad) {
    if (P <= 0) return hipSuccess;
    if (Lx < 0 || Lx > 10 || Lv < 0 || Lv > 4 || !x0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(embed_train_kernel<false>, dim3((unsigned)((P + kEmbedPoints - 1) / kEmbedPoints)), dim3(256), 0, s, rays,
                       ray_ld, const_cast<float*>(z), P, S, Lx, Lv, x0, ld0, x1, ld1, vcat, ldv, voff, EmbedPrologue{nullptr, 0, nullptr, 0, pad});
    return hipGetLastError();
}

hipError_t launch_train_prologue(const float* rays, int ray_ld, int64_t N, int S, int lindisp, const float* t_rand, float* z,
                                 int Lx, int Lv, float* x0, int ld0, float* x1, int ld1, float* vcat, int ldv, int voff,
                                 unsigned* zero, int n_zero, hipStream_t s, int pad) {
    const int64_t P = N * S;
    if (P <= 0) return hipSuccess;
    if (Lx < 0 || Lx > 10 || Lv < 0 || Lv > 4 || !x0 || !z) return hipErrorInvalidValue;
    hipLaunchKernelGGL(embed_train_kernel<true>, dim3((unsigned)((P + kEmbedPoints - 1) / kEmbedPoints)), dim3(256), 0, s, rays,
                       ray_ld, z, P, S, Lx, Lv, x0, ld0, x1, ld1, vcat, ldv, voff, EmbedPrologue{t_rand, lindisp, zero, n_zero, pad});
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// MID (between the passes): raw2outputs of the coarse pass and the resampling it feeds - composite_ray, then sample_pdf_ray
// on the weights the same wavefront has just written (nerf.ipynb:452-467) - one workgroup of 64 per ray, one launch for two.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void train_mid_kernel(const float* __restrict__ raw, int C, const float* __restrict__ z_c,
                                                       const float* __restrict__ rays_d, int d_ld,
                                                       const float* __restrict__ noise, int white_bkgd, int S,
                                                       float* __restrict__ rgb_c, float* __restrict__ w_c,
                                                       const float* __restrict__ u, int n_samples, int n_sort,
                                                       float* __restrict__ z_f) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int64_t ray = blockIdx.x;
    composite_ray(ray, threadIdx.x, raw, C, z_c, rays_d, d_ld, noise, white_bkgd, S, rgb_c, nullptr, nullptr, w_c, nullptr);
    __threadfence_block();
    __syncthreads();      // (one wavefront: the weights it wrote are what it reads next)
    sample_pdf_ray(ray, threadIdx.x, smem, nullptr, w_c, S, 1, z_c, u, S - 1, n_samples, n_sort, nullptr, z_f, nullptr);
}

hipError_t launch_train_mid(const float* raw, int C, const float* z_c, const float* rays_d, int d_ld, const float* noise,
                            int white_bkgd, int64_t N, int S, float* rgb_c, float* w_c, const float* u, int n_samples,
                            float* z_f, hipStream_t s) {
    if (N <= 0) return hipSuccess;
    if (N > 0x7fffffffLL || S < 3) return hipErrorInvalidValue;
    const int M = S - 1;
    int n_sort = 2;
    while (n_sort < M + 1 + n_samples) n_sort <<= 1;
    const size_t lds = sizeof(float) * (size_t)(2 * M + n_sort);
    static size_t raised[64] = {};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (lds > 48 * 1024 && lds > raised[dev]) {
        e = hipFuncSetAttribute((const void*)train_mid_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        raised[dev] = lds;
    }
    hipLaunchKernelGGL(train_mid_kernel, dim3((unsigned)N), dim3(64), lds, s, raw, C, z_c, rays_d, d_ld, noise, white_bkgd, S,
                       rgb_c, w_c, u, n_samples, n_sort, z_f);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// loss = mean((rgb - target)^2) (img2mse, nerf_helpers.py:8); grad = 2 (rgb - target) / numel
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double wsum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wsumf(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

__global__ __launch_bounds__(256) void mse_partial_kernel(const float* __restrict__ x, const float* __restrict__ t,
                                                          int64_t n, float* __restrict__ grad, double* __restrict__ part) {
    double s = 0.0;
    const float scale = 2.0f / (float)n;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float d = __fsub_rn(x[i], t[i]);
        s += (double)__fmul_rn(d, d);
        if (grad) grad[i] = __fmul_rn(scale, d);
    }
    __shared__ double red[4];
    s = wsum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ void mse_final_kernel(const double* __restrict__ part, int nb, double n, float* __restrict__ loss) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        double s = 0.0;
        for (int i = 0; i < nb; ++i) s += part[i];
        *loss = (float)(s / n);
    }
}

hipError_t launch_mse(const float* x, const float* t, int64_t n, float* grad, double* part, float* loss, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    int nb = (int)((n + 255) / 256);
    if (nb > 256) nb = 256;
    hipLaunchKernelGGL(mse_partial_kernel, dim3(nb), dim3(256), 0, s, x, t, n, grad, part);
    hipLaunchKernelGGL(mse_final_kernel, dim3(1), dim3(64), 0, s, part, nb, (double)n, loss);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// backward of raw2outputs (nerf.ipynb:254-349) w.r.t. raw for a loss on rgb_map only:
//   rgb_map = sum_i w_i c_i (+ 1 - sum_i w_i with white_bkgd), w_i = alpha_i T_i, T_i = prod_{j<i}(1-alpha_j+1e-10)
//   dL/dc_i = g w_i ;  dL/dalpha_k = g.(T_k c_k - S_k/(1-alpha_k+1e-10)) - bg (T_k - A_k/(1-alpha_k+1e-10))
//   with suffix sums S_k = sum_{i>k} w_i c_i, A_k = sum_{i>k} w_i and bg = sum(g) for white_bkgd.
// One wavefront per ray; suffix sums are wavefront scans run from the far end.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void composite_bwd_kernel(const float* __restrict__ raw, int C,
                                                           const float* __restrict__ z_vals,
                                                           const float* __restrict__ rays_d, int d_ld,
                                                           const float* __restrict__ noise, int white_bkgd, int S,
                                                           const float* __restrict__ g_rgb, float* __restrict__ d_raw, int dC) {
    extern __shared__ float Tsh[];   // exclusive transmittance of every sample of this ray
    const int64_t ray = blockIdx.x;
    composite_bwd_ray(ray, threadIdx.x, Tsh, raw, C, z_vals, rays_d, d_ld, noise, white_bkgd, S, g_rgb[ray * 3 + 0],
                      g_rgb[ray * 3 + 1], g_rgb[ray * 3 + 2], d_raw, dC);
}

hipError_t launch_composite_bwd(const float* raw, int C, const float* z, const float* rays_d, int d_ld,
                                const float* noise, int white_bkgd, int64_t N, int S, const float* g_rgb,
                                float* d_raw, hipStream_t s, int dC) {
    if (N <= 0) return hipSuccess;
    hipLaunchKernelGGL(composite_bwd_kernel, dim3((unsigned)N), dim3(64), (size_t)S * sizeof(float), s, raw, C, z,
                       rays_d, d_ld, noise, white_bkgd, S, g_rgb, d_raw, dC > 0 ? dC : C);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// EPILOGUE of the forward direction and first launch of the backward one: raw2outputs of the LAST pass (composite_ray),
// both MSE losses with their gradients (the expressions of mse_partial_kernel / mse_final_kernel), and the backward of
// raw2outputs for the last pass and - when there are two - the coarse one (composite_bwd_ray): one workgroup of 64 per ray.
// The loss VALUES need every ray: each workgroup leaves its ray's two sums of squares (double) and takes a ticket; the last
// one adds all of them up in index order (the same sums whichever workgroup that is), divides, and writes the losses, their
// sum and the two PSNRs (mse2psnr, nerf_helpers.py:14: -10 log(x) / log(10), rounded like the three tensor operations the
// Python mirror used to spend on it). One launch where there were eight, and three device-to-device copies less.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void train_epilogue_kernel(const TrainEpilogue e) {
    extern __shared__ float Tsh[];
    __shared__ float rgb_sh[3];
    __shared__ int last_sh;
    const int64_t ray = blockIdx.x;
    const int lane = threadIdx.x;
    float rgb[3] = {0.0f, 0.0f, 0.0f};
    composite_ray(ray, lane, e.raw_l, e.C_l, e.z_l, e.rays_d, e.d_ld, e.noise_l, e.white_bkgd, e.S_l, e.out_rgb, nullptr, nullptr,
                  nullptr, nullptr, rgb);
    if (lane == 0) {
        rgb_sh[0] = rgb[0];
        rgb_sh[1] = rgb[1];
        rgb_sh[2] = rgb[2];
    }
    __syncthreads();
    const float scale = 2.0f / (float)(e.N * 3);
    const float t0 = e.target[ray * 3 + 0], t1 = e.target[ray * 3 + 1], t2 = e.target[ray * 3 + 2];
    const float d0 = __fsub_rn(rgb_sh[0], t0), d1 = __fsub_rn(rgb_sh[1], t1), d2 = __fsub_rn(rgb_sh[2], t2);
    double sq_l = (double)__fmul_rn(d0, d0) + (double)__fmul_rn(d1, d1) + (double)__fmul_rn(d2, d2), sq_c = 0.0;
    composite_bwd_ray(ray, lane, Tsh, e.raw_l, e.C_l, e.z_l, e.rays_d, e.d_ld, e.noise_l, e.white_bkgd, e.S_l,
                      __fmul_rn(scale, d0), __fmul_rn(scale, d1), __fmul_rn(scale, d2), e.d_raw_l, e.dC_l > 0 ? e.dC_l : e.C_l);
    if (e.raw_c) {
        const float c0 = e.rgb_c[ray * 3 + 0], c1 = e.rgb_c[ray * 3 + 1], c2 = e.rgb_c[ray * 3 + 2];
        const float f0 = __fsub_rn(c0, t0), f1 = __fsub_rn(c1, t1), f2 = __fsub_rn(c2, t2);
        sq_c = (double)__fmul_rn(f0, f0) + (double)__fmul_rn(f1, f1) + (double)__fmul_rn(f2, f2);
        if (lane < 3 && e.out_rgb0) e.out_rgb0[ray * 3 + lane] = e.rgb_c[ray * 3 + lane];
        __syncthreads();      // (Tsh is reused)
        composite_bwd_ray(ray, lane, Tsh, e.raw_c, e.C_c, e.z_c, e.rays_d, e.d_ld, e.noise_c, e.white_bkgd, e.S_c,
                          __fmul_rn(scale, f0), __fmul_rn(scale, f1), __fmul_rn(scale, f2), e.d_raw_c, e.dC_c > 0 ? e.dC_c : e.C_c);
    }
    // ---- the loss values: the last workgroup to arrive adds the rays' sums up ----
    if (lane == 0) {
        e.part[ray] = sq_l;
        e.part[e.N + ray] = sq_c;
        __threadfence();
        last_sh = atomicAdd(e.ticket, 1u) == (unsigned)(e.N - 1);
    }
    __syncthreads();
    if (!last_sh) return;
    __threadfence();
    double s_l = 0.0, s_c = 0.0;
    for (int64_t i = lane; i < e.N; i += 64) {
        s_l += __builtin_nontemporal_load(e.part + i);
        s_c += __builtin_nontemporal_load(e.part + e.N + i);
    }
    s_l = wsum(s_l);
    s_c = wsum(s_c);
    if (lane == 0) {
        const double n = (double)(e.N * 3);
        const float loss_l = (float)(s_l / n), loss_c = (float)(s_c / n);
        const float inv_ln10 = __fdiv_rn(1.0f, 2.302585092994046f);       // (torch divides by a Python scalar as x * (1 / s))
        e.loss_dev[0] = loss_l;
        if (e.out_loss) e.out_loss[0] = loss_l;
        if (e.raw_c) {
            e.loss_dev[1] = loss_c;
            if (e.out_loss) e.out_loss[1] = loss_c;
        }
        if (e.out_stats) {      // img_loss, img_loss0, loss, psnr, psnr0
            e.out_stats[0] = loss_l;
            e.out_stats[1] = e.raw_c ? loss_c : 0.0f;
            e.out_stats[2] = e.raw_c ? __fadd_rn(loss_l, loss_c) : loss_l;
            e.out_stats[3] = __fmul_rn(__fmul_rn(logf(loss_l), -10.0f), inv_ln10);
            e.out_stats[4] = e.raw_c ? __fmul_rn(__fmul_rn(logf(loss_c), -10.0f), inv_ln10) : 0.0f;
        }
        *e.ticket = 0u;      // for the next step (the prologue zeroes it as well)
    }
}

// (legacy glue: the same five numbers from the two losses the MSE kernels left)
__global__ void train_stats_kernel(const float* __restrict__ loss, int two, float* __restrict__ stats) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const float inv_ln10 = __fdiv_rn(1.0f, 2.302585092994046f);
    const float l = loss[0], c = two ? loss[1] : 0.0f;
    stats[0] = l;
    stats[1] = c;
    stats[2] = two ? __fadd_rn(l, c) : l;
    stats[3] = __fmul_rn(__fmul_rn(logf(l), -10.0f), inv_ln10);
    stats[4] = two ? __fmul_rn(__fmul_rn(logf(c), -10.0f), inv_ln10) : 0.0f;
}
hipError_t launch_train_stats(const float* loss, bool two, float* stats, hipStream_t s) {
    hipLaunchKernelGGL(train_stats_kernel, dim3(1), dim3(64), 0, s, loss, two ? 1 : 0, stats);
    return hipGetLastError();
}

hipError_t launch_train_epilogue(const TrainEpilogue& e, hipStream_t s) {
    if (e.N <= 0) return hipSuccess;
    if (e.N > 0x7fffffffLL || !e.part || !e.ticket || !e.loss_dev) return hipErrorInvalidValue;
    const int S = e.S_l > e.S_c ? e.S_l : e.S_c;
    hipLaunchKernelGGL(train_epilogue_kernel, dim3((unsigned)e.N), dim3(64), (size_t)S * sizeof(float), s, e);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Adam (torch.optim.Adam defaults except lr/betas given at nerf.ipynb:905), transposes
// ---------------------------------------------------------------------------------------------
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, int64_t n, float lr, float b1, float b2, float eps, float bc1,
                            float bc2_sqrt) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float gi = g[i];
    const float mi = m[i] + (1.0f - b1) * (gi - m[i]);              // exp_avg.lerp_(grad, 1 - beta1)
    const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;             // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] = p[i] - (lr / bc1) * (mi / denom);
}

hipError_t launch_adam(float* p, const float* g, float* m, float* v, int64_t n, float lr, float b1, float b2, float eps,
                       int step, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    const float bc1 = (float)(1.0 - pow((double)b1, (double)step));
    const float bc2_sqrt = (float)sqrt(1.0 - pow((double)b2, (double)step));
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p, g, m, v, n, lr, b1, b2, eps,
                       bc1, bc2_sqrt);
    return hipGetLastError();
}

__global__ void transpose_kernel(const float* __restrict__ src, int rows, int cols, float* __restrict__ dst) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)rows * cols) return;
    const int r = (int)(i / cols), c = (int)(i % cols);
    dst[(int64_t)c * rows + r] = src[i];
}

hipError_t launch_transpose(const float* src, int rows, int cols, float* dst, hipStream_t s) {
    const int64_t n = (int64_t)rows * cols;
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(transpose_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src, rows, cols, dst);
    return hipGetLastError();
}

// stream[i] = table[i] >= 0 ? params[table[i]] : 0   (re-pack of the fused inference stream after an update)
__global__ void gather_kernel(const float* __restrict__ params, const int* __restrict__ table, int64_t n,
                              float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int t = table[i];
    out[i] = t >= 0 ? params[t] : 0.0f;
}

hipError_t launch_gather(const float* params, const int* table, int64_t n, float* out, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(gather_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, params, table, n, out);
    return hipGetLastError();
}

}  // namespace nerf
