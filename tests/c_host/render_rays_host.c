/* A host that is not Python: plain C over include/nerf_mi355x.h, the way a cgo / JNI / N-API binding would drive the
 * library. Loads two 8 x 256 view-dependent networks from a blob of fp32 tensors in
 * nerf_load_weights order, renders rays.bin ([n, 11] fp32) with render_rays' arguments and writes
 * rgb_map | disp_map | acc_map | rgb0. tests/test_c_host.py builds it on the GPU box and compares the output with the
 * Python mirror's, bit for bit.
 *
 *   render_rays_host <libnerf_mi355x.so is linked> weights.bin rays.bin n_rays N_samples N_importance out.bin
 */
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "nerf_mi355x.h"

#define CHECK(call)                                                                      \
    do {                                                                                 \
        if ((call) != 0) {                                                               \
            fprintf(stderr, "%s failed: %s\n", #call, nerf_last_error());                \
            return 2;                                                                    \
        }                                                                                \
    } while (0)
#define HIP(call)                                                                        \
    do {                                                                                 \
        hipError_t e_ = (call);                                                          \
        if (e_ != hipSuccess) {                                                          \
            fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_));                   \
            return 3;                                                                    \
        }                                                                                \
    } while (0)

static float* read_floats(const char* path, size_t* n) {
    FILE* f = fopen(path, "rb");
    if (!f) return NULL;
    fseek(f, 0, SEEK_END);
    long bytes = ftell(f);
    fseek(f, 0, SEEK_SET);
    float* p = (float*)malloc((size_t)bytes);
    if (p && fread(p, 1, (size_t)bytes, f) != (size_t)bytes) {
        free(p);
        p = NULL;
    }
    fclose(f);
    *n = (size_t)bytes / sizeof(float);
    return p;
}

int main(int argc, char** argv) {
    if (argc != 7) {
        fprintf(stderr, "usage: %s weights.bin rays.bin n_rays N_samples N_importance out.bin\n", argv[0]);
        return 1;
    }
    const int64_t n = atoll(argv[3]);
    const int Sc = atoi(argv[4]), Si = atoi(argv[5]);
    size_t nw = 0, nr = 0;
    float* w = read_floats(argv[1], &nw);
    float* rays = read_floats(argv[2], &nr);
    if (!w || !rays || nr != (size_t)n * 11) {
        fprintf(stderr, "bad input files\n");
        return 1;
    }

    nerf_arch arch = {0};
    arch.D = 8;
    arch.W = 256;
    arch.input_ch = 63;
    arch.input_ch_views = 27;
    arch.output_ch = 4;
    arch.n_skips = 1;
    arch.skips[0] = 4;
    arch.use_viewdirs = 1;
    const int nt = nerf_num_weight_tensors(&arch);
    /* tensor sizes in nerf_load_weights order (nerf/nerf.py:32-55) */
    size_t sizes[32];
    int k = 0;
    for (int i = 0; i < arch.D; ++i) {
        const size_t in = i == 0 ? 63 : (i == 5 ? 256 + 63 : 256);
        sizes[k++] = 256 * in;
        sizes[k++] = 256;
    }
    sizes[k++] = 128 * (256 + 27); sizes[k++] = 128;   /* views_linears.0 */
    sizes[k++] = 256 * 256;        sizes[k++] = 256;   /* feature_linear  */
    sizes[k++] = 256;              sizes[k++] = 1;     /* alpha_linear    */
    sizes[k++] = 3 * 128;          sizes[k++] = 3;     /* rgb_linear      */
    if (k != nt) {
        fprintf(stderr, "tensor count %d != %d\n", k, nt);
        return 1;
    }
    size_t per_net = 0;
    for (int i = 0; i < nt; ++i) per_net += sizes[i];
    if (nw != 2 * per_net) {
        fprintf(stderr, "weights.bin holds %zu floats, expected %zu\n", nw, 2 * per_net);
        return 1;
    }

    nerf_ctx* ctx = NULL;
    CHECK(nerf_ctx_create(0, &ctx));
    for (int slot = 0; slot < 2; ++slot) {
        const float* tensors[32];
        const float* p = w + (size_t)slot * per_net;
        for (int i = 0; i < nt; ++i) {
            tensors[i] = p;
            p += sizes[i];
        }
        CHECK(nerf_load_weights(ctx, slot, &arch, tensors, nt));
    }

    float *d_rays, *d_rgb, *d_disp, *d_acc, *d_rgb0;
    HIP(hipMalloc((void**)&d_rays, (size_t)n * 11 * sizeof(float)));
    HIP(hipMalloc((void**)&d_rgb, (size_t)n * 3 * sizeof(float)));
    HIP(hipMalloc((void**)&d_disp, (size_t)n * sizeof(float)));
    HIP(hipMalloc((void**)&d_acc, (size_t)n * sizeof(float)));
    HIP(hipMalloc((void**)&d_rgb0, (size_t)n * 3 * sizeof(float)));
    HIP(hipMemcpy(d_rays, rays, (size_t)n * 11 * sizeof(float), hipMemcpyHostToDevice));
    hipStream_t stream;
    HIP(hipStreamCreate(&stream));

    nerf_render_args a = {0};
    a.rays = d_rays;
    a.n_rays = n;
    a.ray_stride = 11;
    a.N_samples = Sc;
    a.N_importance = Si;
    a.slot_coarse = 0;
    a.slot_fine = 1;
    a.white_bkgd = 1;
    a.rgb_map = d_rgb;
    a.disp_map = d_disp;
    a.acc_map = d_acc;
    a.rgb0 = d_rgb0;
    a.stream = stream;
    CHECK(nerf_render_rays(ctx, &a));          /* enqueued on `stream` */
    HIP(hipStreamSynchronize(stream));

    float* out = (float*)malloc((size_t)n * 8 * sizeof(float));
    HIP(hipMemcpy(out, d_rgb, (size_t)n * 3 * sizeof(float), hipMemcpyDeviceToHost));
    HIP(hipMemcpy(out + n * 3, d_disp, (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
    HIP(hipMemcpy(out + n * 4, d_acc, (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
    HIP(hipMemcpy(out + n * 5, d_rgb0, (size_t)n * 3 * sizeof(float), hipMemcpyDeviceToHost));
    FILE* f = fopen(argv[6], "wb");
    if (!f || fwrite(out, sizeof(float), (size_t)n * 8, f) != (size_t)n * 8) {
        fprintf(stderr, "cannot write %s\n", argv[6]);
        return 1;
    }
    fclose(f);
    nerf_ctx_destroy(ctx);
    printf("ok: %lld rays\n", (long long)n);
    return 0;
}
