// Device functions of the per-ray stages (stratified depths, raw2outputs, sample_pdf + merge), one ray per 64-lane wavefront:
// shared by the stage kernels of ray_kernels.hip and the fused launches of the training step (train_kernels.hip), so that both
// evaluate the same expressions in the same order - the fused launches are bit-identical to the stage kernels by construction.
#pragma once
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

__device__ __forceinline__ float z_at(float near, float far, int i, int S, int lindisp) {
    const float t = linspace01(i, S);
    const float omt = __fsub_rn(1.0f, t);
    if (!lindisp) return __fadd_rn(__fmul_rn(near, omt), __fmul_rn(far, t));            // :421
    const float a = __fmul_rn(__fdiv_rn(1.0f, near), omt);
    const float b = __fmul_rn(__fdiv_rn(1.0f, far), t);
    return __fdiv_rn(1.0f, __fadd_rn(a, b));                                             // :424
}

// one depth of render_rays' stratified sampling (nerf.ipynb:418-444): sample i of S between near and far, jittered inside
// its stratum by *t (a uniform in [0, 1)) when given
__device__ __forceinline__ float stratified_z(float near, float far, int i, int S, int lindisp, const float* t) {
    float z = z_at(near, far, i, S, lindisp);
    if (t) {                                                                              // :428-444
        const float zl = i > 0 ? z_at(near, far, i - 1, S, lindisp) : z;
        const float zu = i < S - 1 ? z_at(near, far, i + 1, S, lindisp) : z;
        const float lower = i > 0 ? __fmul_rn(0.5f, __fadd_rn(z, zl)) : z;
        const float upper = i < S - 1 ? __fmul_rn(0.5f, __fadd_rn(zu, z)) : z;
        z = __fadd_rn(lower, __fmul_rn(__fsub_rn(upper, lower), *t));
    }
    return z;
}

// ---- R6: raw2outputs (nerf.ipynb:254-349) ------------------------------------------------
// raw2outputs for ONE ray by the 64 lanes of a wavefront (`lane`). Lane 0 returns the ray's colour in rgb_out (when given).
__device__ __forceinline__ void composite_ray(int64_t ray, int lane, const float* __restrict__ raw, int C,
                                              const float* __restrict__ z_vals, const float* __restrict__ rays_d, int d_ld,
                                              const float* __restrict__ noise, int white_bkgd, int S,
                                              float* __restrict__ rgb_map, float* __restrict__ disp_map,
                                              float* __restrict__ acc_map, float* __restrict__ weights,
                                              float* __restrict__ depth_map, float* rgb_out = nullptr) {
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
        if (rgb_out) {
            rgb_out[0] = sr;
            rgb_out[1] = sg;
            rgb_out[2] = sb;
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

// ---- R7 + R8: sample_pdf (nerf_helpers.py:372-439), merge + sort (nerf.ipynb:466-467) -----
// sample_pdf + merge for ONE ray by a 64-thread workgroup; smem: (2 M + n_sort) floats
__device__ __forceinline__ void sample_pdf_ray(int64_t ray, int lane, char* smem, const float* __restrict__ bins_in,
                                               const float* __restrict__ weights, int w_ld, int w_off,
                                               const float* __restrict__ z_coarse, const float* __restrict__ u_in, int M,
                                               int n_samples, int n_sort, float* __restrict__ samples_out,
                                               float* __restrict__ z_merged, float* __restrict__ z_std) {
    float* cdf = (float*)smem;
    float* bins = cdf + M;
    float* zall = bins + M;
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

// ---- backward of raw2outputs (training step) ------------------------------------------------
// Backward of raw2outputs for ONE ray by a 64-thread workgroup; Tsh: S floats of LDS (the exclusive transmittance of every
// sample); (g0, g1, g2) = dL/d rgb_map of this ray.
__device__ __forceinline__ void composite_bwd_ray(int64_t ray, int lane, float* Tsh, const float* __restrict__ raw, int C,
                                                  const float* __restrict__ z_vals, const float* __restrict__ rays_d, int d_ld,
                                                  const float* __restrict__ noise, int white_bkgd, int S, float g0, float g1,
                                                  float g2, float* __restrict__ d_raw, int dC) {      // dC: row stride of d_raw (>= C; the rest zeroed)
    const float* d = rays_d + ray * d_ld;
    const float norm = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(d[0], d[0]), __fmul_rn(d[1], d[1])), __fmul_rn(d[2], d[2])));
    const float* z = z_vals + ray * S;
    const float* rw = raw + ray * (int64_t)S * C;
    float* dr = d_raw + ray * (int64_t)S * dC;
    const float gbg = white_bkgd ? (g0 + g1 + g2) : 0.0f;

    auto alpha_at = [&](int i, float& dist, float& sig) {
        dist = __fmul_rn(i < S - 1 ? __fsub_rn(z[i + 1], z[i]) : 1e10f, norm);
        sig = rw[(int64_t)i * C + 3];
        if (noise) sig = __fadd_rn(sig, noise[ray * S + i]);
        return __fsub_rn(1.0f, expf(__fmul_rn(-fmaxf(sig, 0.0f), dist)));
    };
    // forward pass: T_i exactly as the forward kernel computes it (fp64 scan, each prefix rounded to fp32)
    double carry_t = 1.0;
    for (int base = 0; base < S; base += 64) {
        const int i = base + lane;
        float dist, sig;
        const float alpha = i < S ? alpha_at(i, dist, sig) : 0.0f;
        double v = i < S ? (double)__fadd_rn(__fsub_rn(1.0f, alpha), 1e-10f) : 1.0;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const double nb = __shfl_up(v, o);
            if (lane >= o) v *= nb;
        }
        double excl = __shfl_up(v, 1);
        if (lane == 0) excl = 1.0;
        if (i < S) Tsh[i] = (float)(carry_t * excl);
        carry_t *= __shfl(v, 63);
    }
    __syncthreads();
    // backward pass from the far end; carry = sum over samples beyond this round of w_i (g.c_i - gbg)
    float carry = 0.0f;
    for (int rd = (S + 63) / 64 - 1; rd >= 0; --rd) {
        const int i = rd * 64 + lane;
        const bool on = i < S;
        float wi = 0.0f, c0 = 0.0f, c1 = 0.0f, c2 = 0.0f, alpha = 0.0f, dist = 0.0f, sig = 0.0f, T = 0.0f;
        if (on) {
            alpha = alpha_at(i, dist, sig);
            T = Tsh[i];
            wi = __fmul_rn(alpha, T);
            c0 = __fdiv_rn(1.0f, __fadd_rn(1.0f, expf(-rw[(int64_t)i * C + 0])));
            c1 = __fdiv_rn(1.0f, __fadd_rn(1.0f, expf(-rw[(int64_t)i * C + 1])));
            c2 = __fdiv_rn(1.0f, __fadd_rn(1.0f, expf(-rw[(int64_t)i * C + 2])));
        }
        const float gc = g0 * c0 + g1 * c1 + g2 * c2;
        const float q = on ? wi * (gc - gbg) : 0.0f;
        float incl = q;   // inclusive suffix sum over lanes (from lane 63 down)
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const float nb = __shfl_down(incl, o);
            if (lane + o < 64) incl += nb;
        }
        const float suffix_excl = incl - q + carry;   // sum_{i' > i} w_i' (g.c_i' - gbg)
        carry += __shfl(incl, 0);
        if (on) {
            const float om = __fadd_rn(__fsub_rn(1.0f, alpha), 1e-10f);
            const float dalpha_dsig = sig > 0.0f ? dist * (1.0f - alpha) : 0.0f;   // d/dsigma of 1 - exp(-relu(sigma) dist)
            dr[(int64_t)i * dC + 0] = g0 * wi * c0 * (1.0f - c0);
            dr[(int64_t)i * dC + 1] = g1 * wi * c1 * (1.0f - c1);
            dr[(int64_t)i * dC + 2] = g2 * wi * c2 * (1.0f - c2);
            dr[(int64_t)i * dC + 3] = dalpha_dsig * (T * (gc - gbg) - suffix_excl / om);
            for (int c = 4; c < dC; ++c) dr[(int64_t)i * dC + c] = 0.0f;
        }
    }
}

}  // namespace nerf
