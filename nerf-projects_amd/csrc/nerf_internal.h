// Internal declarations shared by the HIP translation units of libnerf_mi355x.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "nerf_mi355x.h"

namespace nerf {

// ---- packed weight stream ---------------------------------------------------------------
// The fused encode+MLP kernel consumes weights as a linear stream of 32 KiB chunks, each
// chunk being 32 groups x 64 lanes x 4 floats: exactly the bytes one workgroup copies
// HBM/L2 -> LDS with 32 `global_load_lds_dwordx4` wave-instructions and then reads back
// as `ds_read_b128` MFMA A-fragments (4 consecutive k-steps per lane per read).
constexpr int kChunkFloats = 8192;
constexpr int kChunkBytes = kChunkFloats * 4;
constexpr int kGroupFloats = 256;            // 64 lanes x 4 k-steps
constexpr int kBiasTileFloats = 32;          // [h(2)][16 accumulator registers]
constexpr int kBiasLdsBytes = 16384;         // up to 128 bias tiles
constexpr int kWidth = 256;                  // trunk width this build is specialised for
constexpr int kMaxDepth = 12;
constexpr int kPointsPerWave = 32;
constexpr int kWavesPerGroup = 4;
constexpr int kPointsPerGroup = kPointsPerWave * kWavesPerGroup;

struct PackedNet {
    nerf_arch arch{};
    bool loaded = false;
    float* d_stream = nullptr;   // n_chunks * kChunkFloats
    float* d_bias = nullptr;     // n_bias_tiles * kBiasTileFloats
    int n_chunks = 0;
    int n_bias_tiles = 0;
    uint32_t skip_in_mask = 0;   // bit i: trunk layer i reads [input_pts, h]
    int out_ch = 4;              // channels NeRF.forward returns
};

enum MlpInputMode { kInputEmbedded = 0, kInputPoints = 1, kInputRays = 2 };

struct MlpLaunch {
    const float* stream;
    const float* bias;
    int n_chunks;
    int n_bias_tiles;
    int D;
    uint32_t skip_in_mask;
    int use_viewdirs;
    int out_ch;
    int in_ch;            // encoded xyz width (columns of x in embedded mode)
    int in_ch_views;
    int64_t n_points;
    int64_t samples_per_ray;
    // kInputEmbedded
    const float* x;
    int x_ld;
    // kInputPoints
    const float* pts;
    const float* viewdirs;   // [n_rays,3] or nullptr
    // kInputRays
    const float* rays;
    int ray_ld;
    const float* z_vals;
    float* out;
};

// host-side packer (pack_weights.cpp)
int pack_weights(const nerf_arch& arch, const float* const* tensors, int n_tensors,
                 float** stream_out, int* n_chunks, float** bias_out, int* n_bias_tiles,
                 uint32_t* skip_in_mask, int* out_ch);

// kernel launchers (mlp_kernel.hip, ray_kernels.hip)
hipError_t launch_mlp(const MlpLaunch& a, int mode, hipStream_t s);
hipError_t launch_embed(const float* x, int64_t n, int multires, float* out, hipStream_t s);
hipError_t launch_stratified(const float* rays, int ray_ld, int64_t N, int S, int lindisp,
                             const float* t_rand, float* z_vals, hipStream_t s);
hipError_t launch_composite(const float* raw, int C, const float* z, const float* rays_d,
                            int d_ld, const float* noise, int white_bkgd, int64_t N, int S,
                            float* rgb, float* disp, float* acc, float* weights, float* depth,
                            hipStream_t s);
hipError_t launch_sample_pdf(const float* bins, const float* weights, int w_ld, int w_off,
                             const float* z_coarse, const float* u, int64_t N, int M,
                             int n_samples, float* samples, float* z_merged, float* z_std,
                             hipStream_t s);

hipError_t launch_raygen(const nerf_camera& cam, int64_t first, int64_t n, float* rays, hipStream_t s);
hipError_t launch_image_metrics(const float* a, const float* b, int H, int W, float max_val, float* tmp,
                                double* partial, float* out, hipStream_t s);

void set_error(const char* fmt, ...);

}  // namespace nerf
