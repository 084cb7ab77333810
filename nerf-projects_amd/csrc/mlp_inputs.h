// Input side of the fused encode+MLP kernels (shared by mlp_kernel.hip and mlp_kernel_h2.hip): where a
// point's coordinates come from and how gamma(xyz) / gamma(dir) are laid out over a wavefront.
#pragma once
#include "nerf_internal.h"

namespace nerf {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

// ---- inputs -----------------------------------------------------------------------------
// Slot maps of the encoded tiles (must match pack_weights.cpp).
__host__ __device__ constexpr int pe_col_xyz(int s, int h) {
    return s < 30 ? 3 + 6 * (s / 3) + 3 * h + (s % 3) : (s == 30 ? (h ? 2 : 0) : (h ? -1 : 1));
}
__host__ __device__ constexpr int pe_col_dir(int t, int h) {
    return t < 12 ? 3 + 6 * (t / 3) + 3 * h + (t % 3) : (t == 12 ? (h ? 2 : 0) : (t == 13 ? (h ? -1 : 1) : -1));
}

// gamma(xyz) and gamma(dir) of one point in the tile layout: half-wave 0 keeps the sines,
// half-wave 1 the cosines. x*2^k is exact in fp32 (embedder.py:48,61); sincosf is the
// accurate ocml routine (not v_sin_f32): arguments reach |x|*512.
__device__ __forceinline__ void encode_point(const float (&p)[3], const float (&d)[3], int h, bool dirs,
                                             f32x16& x0, f32x16& x1, f32x16& dd) {
#pragma unroll
    for (int s = 0; s < 30; ++s) {
        float sn, cs;
#ifdef NERF_ABLATE_PE
        sn = p[s % 3] * (float)(1 << (s / 3)); cs = sn + 1.0f;
#else
        sincosf(p[s % 3] * (float)(1 << (s / 3)), &sn, &cs);
#endif
        const float v = h ? cs : sn;
        if (s < 16) x0[s] = v; else x1[s - 16] = v;
    }
    x1[14] = h ? p[2] : p[0];
    x1[15] = h ? 0.0f : p[1];
    if (dirs) {
#pragma unroll
        for (int t = 0; t < 12; ++t) {
            float sn, cs;
#ifdef NERF_ABLATE_PE
            sn = d[t % 3] * (float)(1 << (t / 3)); cs = sn + 1.0f;
#else
            sincosf(d[t % 3] * (float)(1 << (t / 3)), &sn, &cs);
#endif
            dd[t] = h ? cs : sn;
        }
        dd[12] = h ? d[2] : d[0];
        dd[13] = h ? 0.0f : d[1];
        dd[14] = 0.0f;
        dd[15] = 0.0f;
    } else {
#pragma unroll
        for (int t = 0; t < 16; ++t) dd[t] = 0.0f;
    }
}

template <int MODE>
__device__ __forceinline__ void load_inputs(const MlpLaunch& a, int64_t pt, int h, f32x16& x0, f32x16& x1,
                                            f32x16& dd) {
    if (MODE == kInputEmbedded) {
        const float* row = a.x + pt * a.x_ld;
#pragma unroll
        for (int s = 0; s < 32; ++s) {
            const int c = h ? pe_col_xyz(s, 1) : pe_col_xyz(s, 0);
            const float v = (c >= 0 && c < a.in_ch) ? row[c] : 0.0f;
            if (s < 16) x0[s] = v; else x1[s - 16] = v;
        }
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int c = h ? pe_col_dir(t, 1) : pe_col_dir(t, 0);
            dd[t] = (a.use_viewdirs && c >= 0 && c < a.in_ch_views) ? row[a.in_ch + c] : 0.0f;
        }
        return;
    }
    float p[3], d[3] = {0.0f, 0.0f, 0.0f};
    const int64_t ray = pt / a.samples_per_ray;
    if (MODE == kInputPoints) {
        p[0] = a.pts[pt * 3 + 0];
        p[1] = a.pts[pt * 3 + 1];
        p[2] = a.pts[pt * 3 + 2];
        if (a.viewdirs) {
            d[0] = a.viewdirs[ray * 3 + 0];
            d[1] = a.viewdirs[ray * 3 + 1];
            d[2] = a.viewdirs[ray * 3 + 2];
        }
    } else {
        // pts = rays_o + rays_d * z (nerf.ipynb:447, :468): product and sum rounded separately
        const float* r = a.rays + ray * a.ray_ld;
        const float z = a.z_vals[pt];
#pragma unroll
        for (int c = 0; c < 3; ++c) p[c] = __fadd_rn(r[c], __fmul_rn(r[3 + c], z));
        if (a.ray_ld > 8) {
            d[0] = r[a.ray_ld - 3];
            d[1] = r[a.ray_ld - 2];
            d[2] = r[a.ray_ld - 1];
        }
    }
    encode_point(p, d, h, a.use_viewdirs != 0, x0, x1, dd);
}

}  // namespace nerf
