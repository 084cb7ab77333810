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
#include "ray_device.h"

namespace nerf {

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
__global__ void stratified_kernel(const float* __restrict__ rays, int ray_ld, int64_t N, int S, int lindisp,
                                  const float* __restrict__ t_rand, float* __restrict__ z_vals) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= N * S) return;
    const int64_t ray = idx / S;
    const int i = (int)(idx - ray * S);
    const float near = rays[ray * ray_ld + 6], far = rays[ray * ray_ld + 7];
    z_vals[idx] = stratified_z(near, far, i, S, lindisp, t_rand ? t_rand + idx : nullptr);
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
    composite_ray(blockIdx.x, threadIdx.x, raw, C, z_vals, rays_d, d_ld, noise, white_bkgd, S, rgb_map, disp_map, acc_map,
                  weights, depth_map);
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
    sample_pdf_ray(blockIdx.x, threadIdx.x, smem, bins_in, weights, w_ld, w_off, z_coarse, u_in, M, n_samples, n_sort,
                   samples_out, z_merged, z_std);
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

// what render() does to a ray (o, d) between get_rays and the chunk loop (nerf.ipynb:600-629): the unit viewing direction,
// taken BEFORE the NDC warp, then ndc_rays (nerf_helpers.py:311-369, called with near = 1.0)
// TORCH_GPU_ORDER: torch.norm(d, dim=-1) on the device adds the squares as (x^2 + z^2) + y^2 (two accumulators over the row;
// tools/gpu/norm_probe.py: 0 of 2^20 rows differ, every other order 10 % of them) - what the training loop's batches went
// through while host.pack_rays packed them with tensor operations, and what nerf_pack_rays reproduces bit for bit; the frame
// path (raygen_kernel) keeps the left-to-right sum its fixtures from the reference's CPU run were checked with.
template <bool TORCH_GPU_ORDER = false>
__device__ __forceinline__ void unit_direction(const float (&d)[3], float (&v)[3]) {
    const float xx = __fmul_rn(d[0], d[0]), yy = __fmul_rn(d[1], d[1]), zz = __fmul_rn(d[2], d[2]);
    const float nrm = sqrtf(TORCH_GPU_ORDER ? __fadd_rn(__fadd_rn(xx, zz), yy) : __fadd_rn(__fadd_rn(xx, yy), zz));
#pragma unroll
    for (int k = 0; k < 3; ++k) v[k] = __fdiv_rn(d[k], nrm);
}
__device__ __forceinline__ void ndc_warp(float ndc_cw, float ndc_ch, float (&o)[3], float (&d)[3]) {
    const float near = 1.0f;                                                     // nerf.ipynb:619
    const float tt = __fdiv_rn(-__fadd_rn(near, o[2]), d[2]);                    // nerf_helpers.py:342
#pragma unroll
    for (int k = 0; k < 3; ++k) o[k] = __fadd_rn(o[k], __fmul_rn(tt, d[k]));
    const float o0 = __fdiv_rn(__fmul_rn(ndc_cw, o[0]), o[2]);
    const float o1 = __fdiv_rn(__fmul_rn(ndc_ch, o[1]), o[2]);
    const float o2 = __fadd_rn(1.0f, __fdiv_rn(2.0f * near, o[2]));
    const float d0 = __fmul_rn(ndc_cw, __fsub_rn(__fdiv_rn(d[0], d[2]), __fdiv_rn(o[0], o[2])));
    const float d1 = __fmul_rn(ndc_ch, __fsub_rn(__fdiv_rn(d[1], d[2]), __fdiv_rn(o[1], o[2])));
    const float d2 = __fdiv_rn(-2.0f * near, o[2]);
    o[0] = o0; o[1] = o1; o[2] = o2;
    d[0] = d0; d[1] = d1; d[2] = d2;
}
__device__ __forceinline__ void write_ray_record(float* r, const float (&o)[3], const float (&d)[3], const float (&v)[3],
                                                 float near, float far, bool use_viewdirs) {
    r[0] = o[0]; r[1] = o[1]; r[2] = o[2];
    r[3] = d[0]; r[4] = d[1]; r[5] = d[2];
    r[6] = near; r[7] = far;
    if (use_viewdirs) { r[8] = v[0]; r[9] = v[1]; r[10] = v[2]; }
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
        unit_direction(d, v);
        if (c.has_static) cam_ray(c.c2w_static, c.fx, c.fy, c.cx, c.cy, i, j, o, d);
    }
    if (c.ndc) ndc_warp(p.ndc_cw, p.ndc_ch, o, d);
    write_ray_record(rays + t * (c.use_viewdirs ? 11 : 8), o, d, v, c.near, c.far, c.use_viewdirs);
}

// render(rays=(rays_o, rays_d)) - the training loop's form (nerf.ipynb:1258) - packs a caller's batch the same way
__global__ void pack_rays_kernel(const float* __restrict__ rays_o, int o_ld, const float* __restrict__ rays_d, int d_ld,
                                 int64_t n, int ndc, float ndc_cw, float ndc_ch, float near, float far, int use_viewdirs,
                                 float* __restrict__ rays) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    float o[3] = {rays_o[t * o_ld], rays_o[t * o_ld + 1], rays_o[t * o_ld + 2]};
    float d[3] = {rays_d[t * d_ld], rays_d[t * d_ld + 1], rays_d[t * d_ld + 2]};
    float v[3] = {0.0f, 0.0f, 0.0f};
    if (use_viewdirs) unit_direction<true>(d, v);
    if (ndc) ndc_warp(ndc_cw, ndc_ch, o, d);
    write_ray_record(rays + t * (use_viewdirs ? 11 : 8), o, d, v, near, far, use_viewdirs != 0);
}

hipError_t launch_pack_rays(const nerf_camera& cam, const float* rays_o, int o_ld, const float* rays_d, int d_ld, int64_t n,
                            float* rays, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    const float cw = (float)(-1.0 / ((double)cam.W / (2.0 * cam.ndc_focal)));
    const float ch = (float)(-1.0 / ((double)cam.H / (2.0 * cam.ndc_focal)));
    hipLaunchKernelGGL(pack_rays_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, rays_o, o_ld, rays_d, d_ld, n,
                       cam.ndc, cw, ch, cam.near, cam.far, cam.use_viewdirs, rays);
    return hipGetLastError();
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
