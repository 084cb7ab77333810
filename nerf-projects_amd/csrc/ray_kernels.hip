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
    int n_sort = 0;                       // the merge buffer exists only when a merged output is asked for
    if (z_merged) {
        n_sort = 2;
        while (n_sort < M + 1 + n_samples) n_sort <<= 1;
    }
    const size_t lds = sizeof(float) * (size_t)(2 * M + n_sort);
    // up to 2 x 4096 + 8192 floats = 64 KiB at the documented argument limits: above the 48 KiB a kernel gets by default
    static size_t raised[64] = {};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (lds > 48 * 1024 && lds > raised[dev]) {
        e = hipFuncSetAttribute((const void*)sample_pdf_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        raised[dev] = lds;
    }
    hipLaunchKernelGGL(sample_pdf_kernel, dim3((unsigned)N), dim3(64), lds, s, bins, weights, w_ld, w_off, z_coarse,
                       u, M, n_samples, n_sort, samples, z_merged, z_std);
    return hipGetLastError();
}

}  // namespace nerf

namespace nerf {

// ---- R0 / f1: get_rays + render()'s ray packing (nerf_helpers.py:222-369, nerf.ipynb:596-629) ----
struct RayGenParams {
    nerf_camera cam;
    float ndc_cw, ndc_ch;   // -1/(W/(2 focal)), -1/(H/(2 focal)) evaluated in double, cast like torch does
    int64_t first, n;
};

__device__ __forceinline__ void cam_ray(const float* c2w, float fx, float fy, float cx, float cy, float i, float j,
                                        float (&o)[3], float (&d)[3]) {
    const float dir0 = __fdiv_rn(__fsub_rn(i, cx), fx);
    const float dir1 = -__fdiv_rn(__fsub_rn(j, cy), fy);
    const float dir2 = -1.0f;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        // torch.sum(dirs[..., None, :] * c2w[:3,:3], -1): three rounded products, summed left to right
        d[r] = __fadd_rn(__fadd_rn(__fmul_rn(dir0, c2w[4 * r + 0]), __fmul_rn(dir1, c2w[4 * r + 1])),
                         __fmul_rn(dir2, c2w[4 * r + 2]));
        o[r] = c2w[4 * r + 3];
    }
}

__global__ void raygen_kernel(const RayGenParams p, float* __restrict__ rays) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= p.n) return;
    const nerf_camera& c = p.cam;
    const int64_t pix = p.first + t;
    const float j = (float)(pix / c.W), i = (float)(pix % c.W);   // integer pixel centres, no +0.5
    float o[3], d[3], v[3] = {0.0f, 0.0f, 0.0f};
    cam_ray(c.c2w, c.fx, c.fy, c.cx, c.cy, i, j, o, d);
    if (c.use_viewdirs) {
        // unit direction of the *viewing* camera, before NDC and before the static-camera override
        const float nrm = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(d[0], d[0]), __fmul_rn(d[1], d[1])), __fmul_rn(d[2], d[2])));
#pragma unroll
        for (int k = 0; k < 3; ++k) v[k] = __fdiv_rn(d[k], nrm);
        if (c.has_static) cam_ray(c.c2w_static, c.fx, c.fy, c.cx, c.cy, i, j, o, d);
    }
    if (c.ndc) {
        const float near = 1.0f;                                                     // nerf.ipynb:619
        const float tt = __fdiv_rn(-__fadd_rn(near, o[2]), d[2]);                    // nerf_helpers.py:342
#pragma unroll
        for (int k = 0; k < 3; ++k) o[k] = __fadd_rn(o[k], __fmul_rn(tt, d[k]));
        const float o0 = __fdiv_rn(__fmul_rn(p.ndc_cw, o[0]), o[2]);
        const float o1 = __fdiv_rn(__fmul_rn(p.ndc_ch, o[1]), o[2]);
        const float o2 = __fadd_rn(1.0f, __fdiv_rn(2.0f * near, o[2]));
        const float d0 = __fmul_rn(p.ndc_cw, __fsub_rn(__fdiv_rn(d[0], d[2]), __fdiv_rn(o[0], o[2])));
        const float d1 = __fmul_rn(p.ndc_ch, __fsub_rn(__fdiv_rn(d[1], d[2]), __fdiv_rn(o[1], o[2])));
        const float d2 = __fdiv_rn(-2.0f * near, o[2]);
        o[0] = o0; o[1] = o1; o[2] = o2;
        d[0] = d0; d[1] = d1; d[2] = d2;
    }
    const int ld = c.use_viewdirs ? 11 : 8;
    float* r = rays + t * ld;
    r[0] = o[0]; r[1] = o[1]; r[2] = o[2];
    r[3] = d[0]; r[4] = d[1]; r[5] = d[2];
    r[6] = c.near; r[7] = c.far;
    if (c.use_viewdirs) { r[8] = v[0]; r[9] = v[1]; r[10] = v[2]; }
}

hipError_t launch_raygen(const nerf_camera& cam, int64_t first, int64_t n, float* rays, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    RayGenParams p;
    p.cam = cam;
    p.ndc_cw = (float)(-1.0 / ((double)cam.W / (2.0 * cam.ndc_focal)));
    p.ndc_ch = (float)(-1.0 / ((double)cam.H / (2.0 * cam.ndc_focal)));
    p.first = first;
    p.n = n;
    hipLaunchKernelGGL(raygen_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p, rays);
    return hipGetLastError();
}

// ---- f4: SSIM + MSE (nerf_helpers.py:8, 21-111) -----------------------------------------------
__constant__ float kGauss11[11];

// horizontal pass: 5 filtered planes (x, y, x^2, y^2, xy) of the clamped images, zero padded
__global__ void ssim_rows_kernel(const float* __restrict__ a, const float* __restrict__ b, int H, int W, float max_val,
                                 float* __restrict__ tmp) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // over H*W*3
    const int64_t total = (int64_t)H * W * 3;
    if (idx >= total) return;
    const int ch = (int)(idx % 3);
    const int col = (int)((idx / 3) % W);
    const int64_t row = idx / (3 * (int64_t)W);
    float s0 = 0, s1 = 0, s2 = 0, s3 = 0, s4 = 0;
#pragma unroll
    for (int k = 0; k < 11; ++k) {
        const int cc = col + k - 5;
        if (cc < 0 || cc >= W) continue;
        const int64_t q = (row * W + cc) * 3 + ch;
        const float x = fminf(fmaxf(a[q], 0.0f), max_val), y = fminf(fmaxf(b[q], 0.0f), max_val);
        const float w = kGauss11[k];
        s0 += __fmul_rn(w, x);
        s1 += __fmul_rn(w, y);
        s2 += __fmul_rn(w, __fmul_rn(x, x));
        s3 += __fmul_rn(w, __fmul_rn(y, y));
        s4 += __fmul_rn(w, __fmul_rn(x, y));
    }
    tmp[idx] = s0;
    tmp[total + idx] = s1;
    tmp[2 * total + idx] = s2;
    tmp[3 * total + idx] = s3;
    tmp[4 * total + idx] = s4;
}

// vertical pass + SSIM map + per-block partial sums of (ssim, squared error)
__global__ __launch_bounds__(256) void ssim_cols_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                        const float* __restrict__ tmp, int H, int W, float max_val,
                                                        double* __restrict__ partial) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = (int64_t)H * W * 3;
    double ssim = 0.0, se = 0.0;
    if (idx < total) {
        const int64_t row = idx / (3 * (int64_t)W);
        const int64_t in_row = idx % (3 * (int64_t)W);
        float m0 = 0, m1 = 0, e00 = 0, e11 = 0, e01 = 0;
#pragma unroll
        for (int k = 0; k < 11; ++k) {
            const int64_t rr = row + k - 5;
            if (rr < 0 || rr >= H) continue;
            const int64_t q = rr * 3 * (int64_t)W + in_row;
            const float w = kGauss11[k];
            m0 += __fmul_rn(w, tmp[q]);
            m1 += __fmul_rn(w, tmp[total + q]);
            e00 += __fmul_rn(w, tmp[2 * total + q]);
            e11 += __fmul_rn(w, tmp[3 * total + q]);
            e01 += __fmul_rn(w, tmp[4 * total + q]);
        }
        const float mu00 = __fmul_rn(m0, m0), mu11 = __fmul_rn(m1, m1), mu01 = __fmul_rn(m0, m1);
        const float s00 = fmaxf(__fsub_rn(e00, mu00), 0.0f), s11 = fmaxf(__fsub_rn(e11, mu11), 0.0f);
        float s01 = __fsub_rn(e01, mu01);
        const float lim = fminf(sqrtf(__fmul_rn(s00, s11)), fabsf(s01));
        s01 = s01 > 0.0f ? lim : (s01 < 0.0f ? -lim : 0.0f);
        const float c1 = (0.01f * max_val) * (0.01f * max_val), c2 = (0.03f * max_val) * (0.03f * max_val);
        const float numer = __fmul_rn(__fadd_rn(__fmul_rn(2.0f, mu01), c1), __fadd_rn(__fmul_rn(2.0f, s01), c2));
        const float denom = __fmul_rn(__fadd_rn(__fadd_rn(mu00, mu11), c1), __fadd_rn(__fadd_rn(s00, s11), c2));
        ssim = (double)__fdiv_rn(numer, denom);
        const float x = fminf(fmaxf(a[idx], 0.0f), 1.0f), y = fminf(fmaxf(b[idx], 0.0f), 1.0f);   // calculate_metrics clips to [0,1]
        const float dxy = __fsub_rn(x, y);
        se = (double)__fmul_rn(dxy, dxy);
    }
    __shared__ double red[2][4];
    ssim = wave_sum(ssim);
    se = wave_sum(se);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) { red[0][wv] = ssim; red[1][wv] = se; }
    __syncthreads();
    if (threadIdx.x == 0) {
        partial[2 * (int64_t)blockIdx.x + 0] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        partial[2 * (int64_t)blockIdx.x + 1] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    }
}

__global__ __launch_bounds__(256) void metrics_final_kernel(const double* __restrict__ partial, int64_t n_blocks,
                                                            double count, float* __restrict__ out) {
    double s = 0.0, e = 0.0;
    for (int64_t i = threadIdx.x; i < n_blocks; i += 256) { s += partial[2 * i]; e += partial[2 * i + 1]; }
    __shared__ double red[2][4];
    s = wave_sum(s);
    e = wave_sum(e);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) { red[0][wv] = s; red[1][wv] = e; }
    __syncthreads();
    if (threadIdx.x == 0) {
        out[0] = (float)((red[0][0] + red[0][1] + red[0][2] + red[0][3]) / count);
        out[1] = (float)((red[1][0] + red[1][1] + red[1][2] + red[1][3]) / count);
    }
}

hipError_t launch_image_metrics(const float* a, const float* b, int H, int W, float max_val, float* tmp, double* partial,
                                float* out, hipStream_t s) {
    // filt = exp(-0.5*((arange(11) - 5 + 0)/1.5)^2), normalised (nerf_helpers.py:59-62), evaluated in fp32 like torch
    static bool uploaded[64] = {};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 64 && !uploaded[dev]) {
        float f[11], sum = 0.0f;
        for (int k = 0; k < 11; ++k) {
            const float z = (float)(k - 5) / 1.5f;
            f[k] = expf(-0.5f * (z * z));
            sum += f[k];
        }
        for (int k = 0; k < 11; ++k) f[k] /= sum;
        e = hipMemcpyToSymbol(HIP_SYMBOL(kGauss11), f, sizeof(f));
        if (e != hipSuccess) return e;
        uploaded[dev] = true;
    }
    const int64_t total = (int64_t)H * W * 3;
    const unsigned blocks = (unsigned)((total + 255) / 256);
    hipLaunchKernelGGL(ssim_rows_kernel, dim3(blocks), dim3(256), 0, s, a, b, H, W, max_val, tmp);
    hipLaunchKernelGGL(ssim_cols_kernel, dim3(blocks), dim3(256), 0, s, a, b, tmp, H, W, max_val, partial);
    hipLaunchKernelGGL(metrics_final_kernel, dim3(1), dim3(256), 0, s, partial, (int64_t)blocks, (double)total, out);
    return hipGetLastError();
}

}  // namespace nerf
