// After an optimiser step: everything the fp16-pair kernels read, rebuilt from the row-equalised parameters in TWO launches.
//
// The training step changes the master parameters once per iteration (nerf.ipynb:1275, optimizer.step()); the fused kernels read
// packed derivatives of them: per network the forward weight stream (fp32 -> fp16 pairs, one power-of-two scale per layer), the
// bias block, the per-layer gains that bound a layer's outputs, and - for the backward-data kernel - the transposed-weight
// stream with its scales and gains. Stage by stage that was 11 launches per network and direction (gathers, per-chunk maxima,
// conversions, a copy of the stream's head behind its end, memsets, gains: api.cpp refresh_h2_many, train_api.cpp
// refresh_bwd - which remain, for loading and for the first use of a stream). Here:
//   refresh_gather_kernel    every stream's chunks gathered through their index tables with the chunk's largest |w| found on the
//                            way (gather_kernel + chunk_absmax_kernel), the bias blocks gathered, the gain tables zeroed
//   refresh_convert_kernel   every chunk split into fp16 (hi, lo) pairs at its layer's scale (convert_stream_h2_kernel, incl.
//                            the copy of a stream's first chunks behind its end), the forward and backward gains
//                            (layer_gain_kernel, layer_gain_bwd_kernel), and the precision guard's counters copied to their
//                            host mirror
// Every value is a gather, a maximum or an exact power-of-two scaling followed by the same roundings: bit-identical to the stage
// kernels (tests/test_hip_parity.py::test_train_glue_is_bit_identical runs both).
#include "nerf_internal.h"

namespace nerf {

typedef float f32x4r __attribute__((ext_vector_type(4)));
typedef unsigned u32x4r __attribute__((ext_vector_type(4)));
typedef _Float16 h16x2r __attribute__((ext_vector_type(2)));
typedef float f32x2r __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(1024) void refresh_gather_kernel(const RefreshBatch b) {
    __shared__ float red[16];
    int blk = blockIdx.x;
    for (int si = 0; si < b.n_streams; ++si) {
        const RefreshStream& st = b.st[si];
        if (blk < st.n_chunks) {
            // one chunk: stream[i] = table[i] >= 0 ? params[table[i]] : 0 (gather_kernel), and its largest |value| (chunk_absmax_kernel)
            const int* t = st.table + (size_t)blk * kChunkFloats;
            float* o = st.stream + (size_t)blk * kChunkFloats;
            float m = 0.0f;
            for (int i = threadIdx.x; i < kChunkFloats; i += 1024) {
                const int ti = t[i];
                const float v = ti >= 0 ? st.params[ti] : 0.0f;
                o[i] = v;
                m = fmaxf(m, fabsf(v));
            }
            for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
            if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
            __syncthreads();
            if (threadIdx.x == 0) {
                float mm = red[0];
                for (int w = 1; w < 16; ++w) mm = fmaxf(mm, red[w]);
                st.chunk_max[blk] = mm;
            }
            return;
        }
        blk -= st.n_chunks;
    }
    for (int bi = 0; bi < b.n_bias; ++bi) {
        const RefreshBias& bs = b.bias[bi];
        const int n_blocks = (bs.n + 1023) / 1024;
        if (blk < n_blocks) {
            const int i = blk * 1024 + threadIdx.x;
            if (i < bs.n) {
                const int ti = bs.table[i];
                bs.out[i] = ti >= 0 ? bs.params[ti] : 0.0f;
            }
            return;
        }
        blk -= n_blocks;
    }
    // the last block: the gain tables start from zero (the forward gains are integer maxima of bit patterns)
    for (int gi = 0; gi < b.n_gain; ++gi)
        for (int i = threadIdx.x; i < 2 * b.gain[gi].n; i += 1024) b.gain_out[gi][i] = 0.0f;
}

__global__ __launch_bounds__(1024) void refresh_convert_kernel(const RefreshBatch b) {
    __shared__ float part[4][256];
    __shared__ float red[2][16];
    int blk = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int si = 0; si < b.n_streams; ++si) {
        const RefreshStream& st = b.st[si];
        if (blk < st.n_chunks) {
            // convert_stream_h2_kernel for this chunk: the layer's scale from its chunks' maxima, then 1024 work items
            const int chunk = blk, layer = st.chunk_layer[chunk];
            float m = 0.0f;
            bool first = true;
            for (int i = 0; i < st.n_chunks; ++i)
                if (st.chunk_layer[i] == layer) {
                    m = fmaxf(m, st.chunk_max[i]);
                    if (i < chunk) first = false;
                }
            int e = (m > 0.0f && m < __builtin_inff()) ? 13 - __builtin_amdgcn_frexp_expf(m) : 0;
            e = e < -60 ? -60 : (e > 60 ? 60 : e);
            const float sc = __builtin_ldexpf(1.0f, e);
            if (first && threadIdx.x == 0) st.descale[layer] = __builtin_ldexpf(1.0f, -e);
            const float* src = st.stream + (size_t)chunk * kChunkFloats;
            uint32_t* dst = st.out_h2 + (size_t)chunk * kChunkFloats;
            // the kernels' weight ring runs three chunks ahead across tile boundaries: a copy of the stream's head follows its end
            uint32_t* dst2 = chunk < kStreamTailChunks ? st.out_h2 + (size_t)(st.n_chunks + chunk) * kChunkFloats : nullptr;
            const int w = threadIdx.x;      // (unit u, k-slice s, lane): 8 x 2 x 64
            const int l = w & 63, s = (w >> 6) & 1, u = w >> 7;
            const f32x4r a = *(const f32x4r*)(src + ((4 * u + 2 * s) * 64 + l) * 4);
            const f32x4r c = *(const f32x4r*)(src + ((4 * u + 2 * s + 1) * 64 + l) * 4);
            const float v[8] = {a[0], a[1], a[2], a[3], c[0], c[1], c[2], c[3]};
            u32x4r hi, lo;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float x0 = v[2 * q] * sc, x1 = v[2 * q + 1] * sc;
                const f32x2r xx = {x0, x1};
                const h16x2r p = __builtin_convertvector(xx, h16x2r);
                const f32x2r rr = {x0 - (float)p[0], x1 - (float)p[1]};
                const h16x2r r = __builtin_convertvector(rr, h16x2r);
                hi[q] = __builtin_bit_cast(unsigned, p);
                lo[q] = __builtin_bit_cast(unsigned, r);
            }
            *(u32x4r*)(dst + ((4 * u + 2 * s) * 64 + l) * 4) = hi;
            *(u32x4r*)(dst + ((4 * u + 2 * s + 1) * 64 + l) * 4) = lo;
            if (dst2) {
                *(u32x4r*)(dst2 + ((4 * u + 2 * s) * 64 + l) * 4) = hi;
                *(u32x4r*)(dst2 + ((4 * u + 2 * s + 1) * 64 + l) * 4) = lo;
            }
            return;
        }
        blk -= st.n_chunks;
    }
    for (int gi = 0; gi < b.n_gain; ++gi) {
        // layer_gain_kernel: block = (layer, group of 16 rows), a wavefront per row
        const GainRefs& refs = b.gain[gi];
        int max_out = 1;
        for (int l = 0; l < refs.n; ++l) max_out = refs.out[l] > max_out ? refs.out[l] : max_out;
        const int groups = (max_out + 15) / 16, n_blocks = refs.n * groups;
        if (blk < n_blocks) {
            const int l = blk / groups, row = (blk % groups) * 16 + wave, n_in = refs.in[l];
            float g = 0.0f, bm = 0.0f;
            if (row < refs.out[l]) {
                const float* w = b.gain_params[gi] + refs.w_off[l] + (size_t)row * n_in;
                float s = 0.0f;
                for (int k = lane; k < n_in; k += 64) s += fabsf(w[k]);
                for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
                g = s;
                bm = fabsf(b.gain_params[gi][refs.b_off[l] + row]);
            }
            if (lane == 0) {
                red[0][wave] = g;
                red[1][wave] = bm;
            }
            __syncthreads();
            if (threadIdx.x == 0) {
                float gm = 0.0f, bb = 0.0f;
                for (int w = 0; w < 16; ++w) {
                    gm = fmaxf(gm, red[0][w]);
                    bb = fmaxf(bb, red[1][w]);
                }
                atomicMax((int*)&b.gain_out[gi][2 * l], __float_as_int(gm));
                atomicMax((int*)&b.gain_out[gi][2 * l + 1], __float_as_int(bb));
            }
            return;
        }
        blk -= n_blocks;
    }
    for (int gi = 0; gi < b.n_bgain; ++gi) {
        // layer_gain_bwd_kernel: block = backward layer, thread = (column r, row quarter q)
        const BwdGainRefs& refs = b.bgain[gi];
        if (blk < refs.n) {
            const int l = blk, r = threadIdx.x & 255, q = threadIdx.x >> 8;
            const float* params = b.bgain_params[gi];
            const float* w = params + refs.w_off[l] + refs.col0[l] + r;
            const int n_rows = refs.rows[l], per = (n_rows + 3) / 4;
            float sum = 0.0f;
            for (int c0 = q * per; c0 < (q + 1) * per; c0 += 8) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int c = c0 + u;
                    v[u] = w[(size_t)(c < n_rows ? c : n_rows - 1) * refs.ld[l]];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) sum += (c0 + u < (q + 1) * per && c0 + u < n_rows) ? fabsf(v[u]) : 0.0f;
            }
            part[q][r] = sum;
            __syncthreads();
            if (threadIdx.x < 256) {
                sum = part[0][r] + part[1][r] + part[2][r] + part[3][r];
                float am = (l == 1 && refs.alpha_off != 0xffffffffu) ? fabsf(params[refs.alpha_off + r]) : 0.0f;
                for (int o = 32; o > 0; o >>= 1) {
                    sum = fmaxf(sum, __shfl_xor(sum, o));
                    am = fmaxf(am, __shfl_xor(am, o));
                }
                if ((threadIdx.x & 63) == 0) {
                    red[0][threadIdx.x >> 6] = sum;
                    red[1][threadIdx.x >> 6] = am;
                }
            }
            __syncthreads();
            if (threadIdx.x == 0) {
                b.bgain_out[gi][2 * l] = fmaxf(fmaxf(red[0][0], red[0][1]), fmaxf(red[0][2], red[0][3]));
                b.bgain_out[gi][2 * l + 1] = fmaxf(fmaxf(red[1][0], red[1][1]), fmaxf(red[1][2], red[1][3]));
            }
            return;
        }
        blk -= refs.n;
    }
    // the last block: the precision guard's counters follow the step to their pinned host mirror (nerf_ctx::h_loose)
    if (b.mirror_dst && threadIdx.x < kLooseWords) {
        b.mirror_dst[threadIdx.x] = __builtin_nontemporal_load(b.mirror_src + threadIdx.x);
        __threadfence_system();
    }
}

hipError_t launch_refresh(const RefreshBatch& b, hipStream_t s) {
    if (b.n_streams < 0 || b.n_streams > 4 || b.n_bias < 0 || b.n_bias > 2 || b.n_gain < 0 || b.n_gain > 2 || b.n_bgain < 0 ||
        b.n_bgain > 2)
        return hipErrorInvalidValue;
    int chunks = 0, bias_blocks = 0, gain_blocks = 0, bgain_blocks = 0;
    for (int i = 0; i < b.n_streams; ++i) {
        if (b.st[i].n_chunks < kStreamTailChunks) return hipErrorInvalidValue;
        chunks += b.st[i].n_chunks;
    }
    for (int i = 0; i < b.n_bias; ++i) bias_blocks += (b.bias[i].n + 1023) / 1024;
    for (int i = 0; i < b.n_gain; ++i) {
        int max_out = 1;
        for (int l = 0; l < b.gain[i].n; ++l) max_out = b.gain[i].out[l] > max_out ? b.gain[i].out[l] : max_out;
        gain_blocks += b.gain[i].n * ((max_out + 15) / 16);
    }
    for (int i = 0; i < b.n_bgain; ++i) bgain_blocks += b.bgain[i].n;
    hipLaunchKernelGGL(refresh_gather_kernel, dim3((unsigned)(chunks + bias_blocks + 1)), dim3(1024), 0, s, b);
    hipLaunchKernelGGL(refresh_convert_kernel, dim3((unsigned)(chunks + gain_blocks + bgain_blocks + 1)), dim3(1024), 0, s, b);
    return hipGetLastError();
}

}  // namespace nerf
