// Per-ray kernels around the MLP: stratified depths, positional encoding (stage API),
// sigma->alpha compositing, inverse-CDF resampling + depth merge. gfx950 only.
//
// All of these are bandwidth-trivial next to the MLP (<0.5 % of a frame):
// one 64-lane wavefront owns one ray, per-ray state lives in registers/LDS, prefix
// products and sums are wavefront shuffles. Rounding follows the reference op by op
// (explicit __f*_rn so the compiler cannot contract across PyTorch's op boundaries), and
// the two scans accumulate in fp64 and round every prefix to fp32 because that is what
// torch.cumsum / torch.cumprod do on the reference's CPU path (ATen acc_type<float> = double).
#include <math.h>

#include "nerf_internal.h"

namespace nerf {

// torch.linspace(0, 1, S)[i] in fp32: both halves are a single fused multiply-add of the
// fp32 step (checked bit-for-bit against torch 2.10, tests/golden/linspace.npz).
__device__ __forceinline__ float linspace01(int i, int S) {
    if (S <= 1) return 0.0f;
    const float step = __fdiv_rn(1.0f, (float)(S - 1));
    return i < S / 2 ? fmaf(step, (float)i, 0.0f) : fmaf(-step, (float)(S - 1 - i), 1.0f);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// inclusive scans over the 64 lanes of a wavefront (Kogge-Stone on shuffles)
__device__ __forceinline__ double wave_scan_mul(double v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const double n = __shfl_up(v, o);
        if (lane >= o) v *= n;
    }
    return v;
}
__device__ __forceinline__ double wave_scan_add(double v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const double n = __shfl_up(v, o);
        if (lane >= o) v += n;
    }
    return v;
}

// ---- R3: Embedder.embed (nerf/embedder.py:72-80) -----------------------------------------
__global__ void embed_kernel(const float* __restrict__ x, int64_t n, int multires, float* __restrict__ out) {
    const int C = 3 + 6 * multires;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * C) return;
    const int64_t p = idx / C;
    const int c = (int)(idx - p * C);
    float v;
    if (c < 3) {
        v = x[p * 3 + c];
    } else {
        const int g = c - 3, k = g / 6, r = g % 6;
        const float arg = x[p * 3 + (r % 3)] * (float)(1 << k);   // exact: power-of-two frequency
        v = r < 3 ? sinf(arg) : cosf(arg);
    }
    out[idx] = v;
}

hipError_t launch_embed(const float* x, int64_t n, int multires, float* out, hipStream_t s) {
    const int64_t total = n * (3 + 6 * multires);
    if (total <= 0) return hipSuccess;
    hipLaunchKernelGGL(embed_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, n, multires, out);
    return hipGetLastError();
}

// ---- R2: stratified depths (nerf.ipynb:418-444) ------------------------------------------
__device__ __forceinline__ float z_at(float near, float far, int i, int S, int lindisp) {
    const float t = linspace01(i, S);
    const float omt = __fsub_rn(1.0f, t);
    if (!lindisp) return __fadd_rn(__fmul_rn(near, omt), __fmul_rn(far, t));            // :421
    const float a = __fmul_rn(__fdiv_rn(1.0f, near), omt);
    const float b = __fmul_rn(__fdiv_rn(1.0f, far), t);
    return __fdiv_rn(1.0f, __fadd_rn(a, b));                                             // :424
}

__global__ void stratified_kernel(const float* __restrict__ rays, int ray_ld, int64_t N, int S, int lindisp,
                                  const float* __restrict__ t_rand, float* __restrict__ z_vals) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= N * S) return;
    const int64_t ray = idx / S;
    const int i = (int)(idx - ray * S);
    const float near = rays[ray * ray_ld + 6], far = rays[ray * ray_ld + 7];
    float z = z_at(near, far, i, S, lindisp);
    if (t_rand) {                                                                         // :428-444
        const float zl = i > 0 ? z_at(near, far, i - 1, S, lindisp) : z;
        const float zu = i < S - 1 ? z_at(near, far, i + 1, S, lindisp) : z;
        const float lower = i > 0 ? __fmul_rn(0.5f, __fadd_rn(z, zl)) : z;
        const float upper = i < S - 1 ? __fmul_rn(0.5f, __fadd_rn(zu, z)) : z;
        z = __fadd_rn(lower, __fmul_rn(__fsub_rn(upper, lower), t_rand[idx]));
    }
    z_vals[idx] = z;
}

hipError_t launch_stratified(const float* rays, int ray_ld, int64_t N, int S, int lindisp, const float* t_rand,
                             float* z_vals, hipStream_t s) {
    const int64_t total = N * S;
    if (total <= 0) return hipSuccess;
    hipLaunchKernelGGL(stratified_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, rays, ray_ld, N,
                       S, lindisp, t_rand, z_vals);
    return hipGetLastError();
}

// ---- R6: raw2outputs (nerf.ipynb:254-349) ------------------------------------------------
// One wavefront per ray; samples are taken 64 at a time (lane = sample within the round)
// with a running fp64 transmittance carried between rounds.
__global__ __launch_bounds__(64) void composite_kernel(const float* __restrict__ raw, int C,
                                                       const float* __restrict__ z_vals,
                                                       const float* __restrict__ rays_d, int d_ld,
                                                       const float* __restrict__ noise, int white_bkgd, int S,
                                                       float* __restrict__ rgb_map, float* __restrict__ disp_map,
                                                       float* __restrict__ acc_map, float* __restrict__ weights,
                                                       float* __restrict__ depth_map) {
    const int64_t ray = blockIdx.x;
    const int lane = threadIdx.x;
    const float* d = rays_d + ray * d_ld;
    const float dx = d[0], dy = d[1], dz = d[2];
    const float norm = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz)));   // :305
    const float* z = z_vals + ray * S;
    const float* rw = raw + ray * (int64_t)S * C;

    double carry = 1.0;   // prod of (1 - alpha + 1e-10) over all earlier samples
    float sr = 0.0f, sg = 0.0f, sb = 0.0f, sd = 0.0f, sa = 0.0f;
    for (int base = 0; base < S; base += 64) {
        const int i = base + lane;
        const bool on = i < S;
        float alpha = 0.0f, r = 0.0f, g = 0.0f, b = 0.0f, zi = 0.0f;
        if (on) {
            zi = z[i];
            float dist = i < S - 1 ? __fsub_rn(z[i + 1], zi) : 1e10f;                   // :295-300
            dist = __fmul_rn(dist, norm);
            float sig = rw[(int64_t)i * C + 3];
            if (noise) sig = __fadd_rn(sig, noise[ray * S + i]);                         // :328
            sig = fmaxf(sig, 0.0f);
            alpha = __fsub_rn(1.0f, expf(__fmul_rn(-sig, dist)));                        // :291
            r = __fdiv_rn(1.0f, __fadd_rn(1.0f, expf(-rw[(int64_t)i * C + 0])));        // :308 sigmoid
            g = __fdiv_rn(1.0f, __fadd_rn(1.0f, expf(-rw[(int64_t)i * C + 1])));
            b = __fdiv_rn(1.0f, __fadd_rn(1.0f, expf(-rw[(int64_t)i * C + 2])));
        }
        const double tf = on ? (double)__fadd_rn(__fsub_rn(1.0f, alpha), 1e-10f) : 1.0;
        const double incl = wave_scan_mul(tf, lane);
        double excl = __shfl_up(incl, 1);
        if (lane == 0) excl = 1.0;
        const float T = (float)(carry * excl);                                           // exclusive cumprod (:329)
        carry *= __shfl(incl, 63);
        if (on) {
            const float w = __fmul_rn(alpha, T);
            if (weights) weights[ray * S + i] = w;
            sr += __fmul_rn(w, r);
            sg += __fmul_rn(w, g);
            sb += __fmul_rn(w, b);
            sd += __fmul_rn(w, zi);
            sa += w;
        }
    }
    sr = wave_sum(sr);                                                                    // :332
    sg = wave_sum(sg);
    sb = wave_sum(sb);
    sd = wave_sum(sd);                                                                    // :335
    sa = wave_sum(sa);                                                                    // :343
    if (lane == 0) {
        const float denom = fmaxf(1e-10f, sa);                                           // :339
        const float disp = __fdiv_rn(1.0f, fmaxf(__fdiv_rn(sd, denom), 1e-10f));         // :340
        if (white_bkgd) {                                                                 // :346-347
            const float bg = __fsub_rn(1.0f, sa);
            sr = __fadd_rn(sr, bg);
            sg = __fadd_rn(sg, bg);
            sb = __fadd_rn(sb, bg);
        }
        if (rgb_map) {
            rgb_map[ray * 3 + 0] = sr;
            rgb_map[ray * 3 + 1] = sg;
            rgb_map[ray * 3 + 2] = sb;
        }
        if (disp_map) disp_map[ray] = disp;
        if (acc_map) acc_map[ray] = sa;
        if (depth_map) depth_map[ray] = sd;
    }
}

hipError_t launch_composite(const float* raw, int C, const float* z, const float* rays_d, int d_ld,
                            const float* noise, int white_bkgd, int64_t N, int S, float* rgb, float* disp,
                            float* acc, float* weights, float* depth, hipStream_t s) {
    if (N <= 0) return hipSuccess;
    if (N > 0x7fffffffLL) return hipErrorInvalidValue;
    hipLaunchKernelGGL(composite_kernel, dim3((unsigned)N), dim3(64), 0, s, raw, C, z, rays_d, d_ld, noise,
                       white_bkgd, S, rgb, disp, acc, weights, depth);
    return hipGetLastError();
}

// ---- R7 + R8: sample_pdf (nerf_helpers.py:372-439), merge + sort (nerf.ipynb:466-467) -----
// LDS per ray: cdf[M], bins[M], then the merge buffer (power of two >= S + n_samples).
__global__ __launch_bounds__(64) void sample_pdf_kernel(const float* __restrict__ bins_in,
                                                        const float* __restrict__ weights, int w_ld, int w_off,
                                                        const float* __restrict__ z_coarse,
                                                        const float* __restrict__ u_in, int M, int n_samples,
                                                        int n_sort, float* __restrict__ samples_out,
                                                        float* __restrict__ z_merged, float* __restrict__ z_std) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* cdf = (float*)smem;
    float* bins = cdf + M;
    float* zall = bins + M;
    const int64_t ray = blockIdx.x;
    const int lane = threadIdx.x;
    const int S = M + 1;   // coarse samples when bins are the mid-points
    const int nb = M - 1;  // number of pdf bins

    // bins: given, or z_vals_mid = .5*(z[1:] + z[:-1]) (nerf.ipynb:460)
    for (int i = lane; i < M; i += 64)
        bins[i] = bins_in ? bins_in[ray * M + i]
                          : __fmul_rn(0.5f, __fadd_rn(z_coarse[ray * S + i + 1], z_coarse[ray * S + i]));

    // weights + 1e-5, pdf = w / sum(w) (nerf_helpers.py:396-397)
    const float* w = weights + ray * w_ld + w_off;
    float part = 0.0f;
    for (int i = lane; i < nb; i += 64) part += __fadd_rn(w[i], 1e-5f);
    const float total = wave_sum(part);

    // cdf = cat[0, cumsum(pdf)] (:398-400), prefixes accumulated in fp64 and rounded to fp32
    double carry = 0.0;
    if (lane == 0) cdf[0] = 0.0f;
    for (int base = 0; base < nb; base += 64) {
        const int i = base + lane;
        const double pdf = i < nb ? (double)__fdiv_rn(__fadd_rn(w[i], 1e-5f), total) : 0.0;
        const double incl = wave_scan_add(pdf, lane);
        if (i < nb) cdf[i + 1] = (float)(carry + incl);
        carry += __shfl(incl, 63);
    }
    __syncthreads();

    double sum = 0.0;
    for (int j = lane; j < n_samples; j += 64) {
        const float u = u_in ? u_in[ray * n_samples + j] : linspace01(j, n_samples);      // :404-407
        // searchsorted(cdf, u, right=True): first index with cdf[idx] > u (:423)
        int lo = 0, hi = M;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (cdf[mid] <= u) lo = mid + 1; else hi = mid;
        }
        const int below = max(0, lo - 1), above = min(M - 1, lo);                         // :424-425
        const float c0 = cdf[below], c1 = cdf[above], b0 = bins[below], b1 = bins[above];
        float denom = __fsub_rn(c1, c0);                                                  // :434
        if (denom < 1e-5f) denom = 1.0f;                                                  // :435
        const float t = __fdiv_rn(__fsub_rn(u, c0), denom);                               // :436
        const float smp = __fadd_rn(b0, __fmul_rn(t, __fsub_rn(b1, b0)));                 // :437
        if (samples_out) samples_out[ray * n_samples + j] = smp;
        if (z_merged) zall[S + j] = smp;
        sum += (double)smp;
    }
    if (z_std) {
        // torch.std(z_samples, unbiased=False) (nerf.ipynb:486); ATen accumulates in fp64
        const double mean = wave_sum(sum) / (double)n_samples;
        double m2 = 0.0;
        __syncthreads();
        for (int j = lane; j < n_samples; j += 64) {
            const double dv = (double)(z_merged ? zall[S + j] : samples_out[ray * n_samples + j]) - mean;
            m2 += dv * dv;
        }
        m2 = wave_sum(m2);
        if (lane == 0) z_std[ray] = (float)sqrt(m2 / (double)n_samples);
    }
    if (!z_merged) return;

    // z_vals = sort(cat[z_vals, z_samples]) (nerf.ipynb:467): bitonic network in LDS, padded with +inf
    for (int i = lane; i < S; i += 64) zall[i] = z_coarse[ray * S + i];
    for (int i = S + n_samples + lane; i < n_sort; i += 64) zall[i] = INFINITY;
    __syncthreads();
    for (int k = 2; k <= n_sort; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = lane; t < n_sort / 2; t += 64) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));   // index with bit j clear
                const int p = i | j;
                const float a = zall[i], b = zall[p];
                const bool up = (i & k) == 0;
                if ((a > b) == up) {
                    zall[i] = b;
                    zall[p] = a;
                }
            }
            __syncthreads();
        }
    }
    for (int i = lane; i < S + n_samples; i += 64) z_merged[ray * (int64_t)(S + n_samples) + i] = zall[i];
}

hipError_t launch_sample_pdf(const float* bins, const float* weights, int w_ld, int w_off, const float* z_coarse,
                             const float* u, int64_t N, int M, int n_samples, float* samples, float* z_merged,
                             float* z_std, hipStream_t s) {
    if (N <= 0) return hipSuccess;
    if (N > 0x7fffffffLL) return hipErrorInvalidValue;
    int n_sort = 2;
    while (n_sort < M + 1 + n_samples) n_sort <<= 1;
    const size_t lds = sizeof(float) * (size_t)(2 * M + n_sort);
    hipLaunchKernelGGL(sample_pdf_kernel, dim3((unsigned)N), dim3(64), lds, s, bins, weights, w_ld, w_off, z_coarse,
                       u, M, n_samples, n_sort, samples, z_merged, z_std);
    return hipGetLastError();
}

}  // namespace nerf
