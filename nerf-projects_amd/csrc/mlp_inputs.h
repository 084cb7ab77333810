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
// gamma(xyz) and gamma(dir) of one point in the tile layout (slot maps: nerf_internal.h, pe_col_*): half-wave 0
// keeps the sines, half-wave 1 the cosines. Both half-waves of a point would evaluate the same 30 (12) sincosf,
// so they share the work: half-wave 0 runs the low five (two) frequencies, half-wave 1 the high ones on
// coordinates pre-multiplied by 2^5 (2^2); each lane keeps the component it owns and hands the other one to its
// partner lane (one __shfl_xor(.,32) per slot). x*2^k is exact in fp32 (embedder.py:48,61); sincosf is the
// accurate ocml routine (not v_sin_f32): arguments reach |x|*512.
template <bool WANT_XYZ, bool WANT_DIR>
__device__ __forceinline__ void encode_point(const float (&p)[3], const float (&d)[3], int h, bool dirs,
                                             f32x16& x0, f32x16& x1, f32x16& dd) {
    if constexpr (WANT_XYZ) {
        const float sc = h ? 32.0f : 1.0f;
        const float q[3] = {p[0] * sc, p[1] * sc, p[2] * sc};
#pragma unroll
        for (int j = 0; j < 15; ++j) {
            float sn, cs;
#ifdef NERF_ABLATE_PE
            sn = q[j % 3] * (float)(1 << (j / 3)); cs = sn + 1.0f;
#else
            sincosf(q[j % 3] * (float)(1 << (j / 3)), &sn, &cs);
#endif
            const float own = h ? cs : sn;
            const float other = __shfl_xor(h ? sn : cs, 32);
            x0[j] = own;                                   // slot j
            if (j == 0) x0[15] = other; else x1[j - 1] = other;   // slot j + 15
        }
        x1[14] = h ? p[2] : p[0];
        x1[15] = h ? 0.0f : p[1];
    }
    if constexpr (WANT_DIR) {
        if (dirs) {
            const float sc = h ? 4.0f : 1.0f;
            const float q[3] = {d[0] * sc, d[1] * sc, d[2] * sc};
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                float sn, cs;
#ifdef NERF_ABLATE_PE
                sn = q[j % 3] * (float)(1 << (j / 3)); cs = sn + 1.0f;
#else
                sincosf(q[j % 3] * (float)(1 << (j / 3)), &sn, &cs);
#endif
                dd[j] = h ? cs : sn;
                dd[j + 6] = __shfl_xor(h ? sn : cs, 32);
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
}

// bit 0: a position input of the point is NaN or infinite, bit 1: a direction input is. F.relu propagates NaN
// (nerf/nerf.py:72) and v_max_f32 does not, so the kernels restore the reference's result on such points at the end:
// every output channel NaN for a bad position, the colour channels NaN for a bad direction (sigma comes from the trunk).
constexpr unsigned kBadXyz = 1u, kBadDir = 2u;
__device__ __forceinline__ bool nonfinite(float v) { return !(fabsf(v) <= 3.4028234663852886e38f); }

// WANT_XYZ / WANT_DIR: which tiles the caller uses (the other is left untouched). dir_max: largest |component| of the
// direction (an upper bound of |gamma(dir)| together with 1), or of the encoded direction columns in embedded mode.
// bad: kBadXyz | kBadDir of this point's raw inputs (both halves of a point get the same value).
template <int MODE, bool WANT_XYZ = true, bool WANT_DIR = true>
__device__ __forceinline__ void load_inputs(const MlpLaunch& a, int64_t pt, int h, f32x16& x0, f32x16& x1,
                                            f32x16& dd, float* dir_max = nullptr, unsigned* bad = nullptr) {
    if (MODE == kInputEmbedded) {
        const float* row = a.x + pt * a.x_ld;
        if constexpr (WANT_XYZ) {
#pragma unroll
            for (int s = 0; s < 32; ++s) {
                const int c = h ? pe_col_xyz(s, 1) : pe_col_xyz(s, 0);
                const float v = (c >= 0 && c < a.in_ch) ? row[c] : 0.0f;
                if (s < 16) x0[s] = v; else x1[s - 16] = v;
            }
        }
        float m = 0.0f;
        if (WANT_DIR || dir_max) {
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const int c = h ? pe_col_dir(t, 1) : pe_col_dir(t, 0);
                const float v = (a.use_viewdirs && c >= 0 && c < a.in_ch_views) ? row[a.in_ch + c] : 0.0f;
                if constexpr (WANT_DIR) dd[t] = v;
                m = fmaxf(m, fabsf(v));
            }
        }
        if (dir_max) *dir_max = fmaxf(m, __shfl_xor(m, 32));
        if (bad) {
            bool bx = false, bd = false;
            for (int c = 0; c < a.in_ch; ++c) bx |= nonfinite(row[c]);
            if (a.use_viewdirs)
                for (int c = 0; c < a.in_ch_views; ++c) bd |= nonfinite(row[a.in_ch + c]);
            *bad = (bx ? kBadXyz : 0u) | (bd ? kBadDir : 0u);
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
    if (dir_max) *dir_max = fmaxf(fmaxf(fabsf(d[0]), fabsf(d[1])), fmaxf(fabsf(d[2]), 1.0f));
    if (bad)
        *bad = ((nonfinite(p[0]) || nonfinite(p[1]) || nonfinite(p[2])) ? kBadXyz : 0u) |
               ((nonfinite(d[0]) || nonfinite(d[1]) || nonfinite(d[2])) ? kBadDir : 0u);
    encode_point<WANT_XYZ, WANT_DIR>(p, d, h, a.use_viewdirs != 0, x0, x1, dd);
}

}  // namespace nerf
