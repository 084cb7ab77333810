// dW = dY^T X of the training step without LDS staging, several layers per launch (declared in nerf_internal.h). Its own translation unit: the 256 accumulators of a wave live in the AGPR half of the register file,
// so this file is compiled WITHOUT -amdgpu-mfma-vgpr-form (build.py), unlike the other training kernels.
#include "nerf_internal.h"

#include <cstdlib>

namespace nerf {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// ---------------------------------------------------------------------------------------------
// The same product without LDS: both operands of v_mfma_f32_32x32x2_f32 want, per lane, ONE value of a row of the
// [point][feature] arrays - lane (i, k) takes feature i of point k - so a wave reads its fragments straight from
// global memory, 16 bytes per lane: the four features 4i..4i+3 of a point are the fragments of FOUR interleaved
// 32-row tiles (tile t = features 4i + t), and a half-wave reads 512 contiguous bytes. No staging, no transposition,
// no barrier, no loader waves: a workgroup is 2 x 2 waves of 128 rows x 32 NT columns each (64 NT accumulator
// registers per lane: 256 at NT = 4, hence one wave per SIMD), a k-step is two points = 4 NT MFMAs per wave fed by two
// loads, and the loads of kTnDepth k-steps are in flight, in kTnDepth register sets the loop is unrolled over (hipcc
// issues a group's loads at its head: a whole group of MFMAs, 8 192 matrix-pipe cycles at NT = 4, for them to arrive).
// Every operand byte is read from HBM once per job (the staged kernel read dY once per 128 columns).
//   NT = 4: a job is exactly 256 columns                                     (hidden-width inputs)
//   NT = 1: a job is at most 64 columns, lanes beyond n_end re-read the last (gamma(x) / gamma(d) columns)
// blockIdx.y picks the job (GradBatch): the weight gradients of a whole network share a launch, see nerf_internal.h.
// Rows: Mo must be a multiple of 128 (128 or 256); the wave row of an absent upper half exits at once.
// ---------------------------------------------------------------------------------------------
constexpr int kTnDepth = 8;
// (NT = 1 at depth 16 / 32: 123 / 130 us per launch against 128 - its four MFMAs per two points are 80 % of the fp32 pipe)
template <int NT>
constexpr int tn_depth() { return kTnDepth; }
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));   // 16 bytes at dword alignment (columns 63.. of a concat row)

template <int NT>
struct TnCols {
    float v[NT];
};
template <int NT>
__device__ __forceinline__ TnCols<NT> tn_load_cols(const float* p) {
    TnCols<NT> r;
    if constexpr (NT == 4) {
        const f32x4u q = *(const f32x4u*)p;
        r.v[0] = q[0]; r.v[1] = q[1]; r.v[2] = q[2]; r.v[3] = q[3];
    } else {
        r.v[0] = *p;
    }
    return r;
}

template <int NT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void grad_batch_kernel(const GradBatch b) {
    const GradJob& g = b.job[blockIdx.y];
    const int n_begin = g.n_begin, n_end = g.n_end, width = n_end - n_begin;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, k = lane >> 5, i = lane & 31;
    const int wm = wave >> 1, wn = wave & 1;
    const int m_base = 128 * wm;
    if (m_base >= g.Mo) return;                 // (no barrier anywhere below)
    const int slice = blockIdx.x;
    const int c_base = n_begin + 32 * NT * wn;   // this wave's first column
    if (c_base >= n_end) return;
    int col = c_base + NT * i;
    if (col > n_end - NT) col = n_end - NT;     // NT = 1 only: keeps the load inside the row; those outputs are not stored
    const int64_t p_begin = (int64_t)slice * b.pts_per_slice;
    int64_t p_end = p_begin + b.pts_per_slice;
    if (p_end > b.P) p_end = b.P;
    const int64_t n_pts = p_end > p_begin ? p_end - p_begin : 0;
    const int n_steps = (int)(n_pts / 2);       // whole k-steps (two points); an odd last point goes through the tail

    f32x16 acc[4][NT];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < NT; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][c][r] = 0.0f;
    float asum[4] = {0.0f, 0.0f, 0.0f, 0.0f};

    // A (dY) row-major, or blocked by 32 points (GradJob::blocked bit 0: what the fp16-pair backward-data kernel writes): this
    // lane's 16 bytes of point p - features m_base + 4 i .. + 3 - then sit in piece (m_base + 4 i) / 8 of p's group
    const bool a_blk = (g.blocked & 1) != 0;
    const int64_t a_group = 32 * (int64_t)g.Mo;
    const int a_lane = ((m_base + 4 * i) >> 3) * 256 + (i & 1) * 4 + k * 8;
    auto a_at = [&](int64_t p) -> const float* {      // p: the k-step's first point (even); this lane's point is p + k
        return a_blk ? g.A + (p >> 5) * a_group + (int)(p & 31) * 8 + a_lane : g.A + (p + k) * g.lda + m_base + 4 * i;
    };
    int64_t pp = p_begin;                               // point of the next k-step to be loaded
    const float* pb = g.B + (p_begin + k) * g.ldb + col;
    const int64_t sb = 2 * (int64_t)g.ldb;

    auto step = [&](const f32x4u& a, const TnCols<NT>& b) {
#pragma unroll
        for (int tm = 0; tm < 4; ++tm) {
            // (opaque: as plain C++ hipcc gathers the eight k-steps' additions at the end of a group and keeps a copy of
            // every A register for them - sixteen v_mov_b64 and their loads' waits in front of the group's first MFMA)
            asm volatile("v_add_f32 %0, %0, %1" : "+v"(asum[tm]) : "v"(a[tm]));
#pragma unroll
            for (int tn = 0; tn < NT; ++tn) acc[tm][tn] = mfma32(a[tm], b.v[tn], acc[tm][tn]);
        }
    };

    constexpr int kDepth = tn_depth<NT>();
    f32x4u ra[kDepth];
    TnCols<NT> rb[kDepth];
    int s = 0;
    if (n_steps >= kDepth) {
#pragma unroll
        for (int j = 0; j < kDepth; ++j) {
            ra[j] = *(const f32x4u*)a_at(pp);
            rb[j] = tn_load_cols<NT>(pb);
            pp += 2;
            pb += sb;
        }
        for (; s + 2 * kDepth <= n_steps; s += kDepth) {
#pragma unroll
            for (int j = 0; j < kDepth; ++j) {
                // pinned: the k-step's MFMAs, then its two loads and their address arithmetic - issued while the last of
                // those MFMAs runs. Left alone hipcc hoists the loads, the bias sums and a copy of every operand register
                // of all eight k-steps in front of the group's 128 MFMAs, ~110 instructions with an empty matrix pipe.
                step(ra[j], rb[j]);
                __builtin_amdgcn_sched_barrier(0);
                ra[j] = *(const f32x4u*)a_at(pp);      // k-step s + kDepth + j, into the registers just consumed
                rb[j] = tn_load_cols<NT>(pb);
                pp += 2;
                pb += sb;
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int j = 0; j < kDepth; ++j) step(ra[j], rb[j]);
        s += kDepth;
    }
    // the tail: fewer than kTnDepth whole k-steps and possibly a single last point, loads predicated per lane
    for (int64_t p = p_begin + 2 * (int64_t)s; p < p_end; p += 2) {
        f32x4u a = {0.0f, 0.0f, 0.0f, 0.0f};
        TnCols<NT> b;
#pragma unroll
        for (int tn = 0; tn < NT; ++tn) b.v[tn] = 0.0f;
        if (p + k < p_end) {
            a = *(const f32x4u*)a_at(p);
            b = tn_load_cols<NT>(g.B + (p + k) * g.ldb + col);
        }
        step(a, b);
    }

    // acc[tm][tn][r] at lane (j = i, h = k): row m_base + 4 (r&3 + 8 (r>>2) + 4 h) + tm, column c_base + NT j + tn
    float* part = g.part + (int64_t)slice * g.Mo * width - n_begin;
#pragma unroll
    for (int tm = 0; tm < 4; ++tm)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m_base + 4 * ((r & 3) + 8 * (r >> 2) + 4 * k) + tm;
            float* dst = part + (int64_t)m * width + c_base + NT * i;
            if constexpr (NT == 4) {
                const f32x4u v = {acc[tm][0][r], acc[tm][1][r], acc[tm][2][r], acc[tm][3][r]};
                *(f32x4u*)dst = v;              // NT = 4 blocks lie wholly inside [n_begin, n_end)
            } else {
                if (c_base + i < n_end) *dst = acc[tm][0][r];
            }
        }
    if (g.db && wn == 0) {
#pragma unroll
        for (int tm = 0; tm < 4; ++tm) {
            const float t = asum[tm] + __shfl_xor(asum[tm], 32);
            if (k == 0) g.dbp[(int64_t)slice * g.Mo + m_base + 4 * i + tm] = t;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// The wide jobs on the fp16 matrix pipe ("fp16-pair" arithmetic, as mlp_kernel_h2.hip: every operand as hi = rn16(v),
// lo = rn16(v - hi), a product as lo*hi + hi*lo + hi*hi in three v_mfma_f32_32x32x16_f16, fp32 accumulate). The
// contraction runs over POINTS, so each operand takes ONE power-of-two scale per tensor (largest |value| -> [2^13, 2^14);
// the maxima come from the backward-data kernel, which writes dY and reads X anyway), undone once on the accumulators.
// v_mfma_f32_32x32x16_f16 wants eight points per lane and feature where the fp32 instruction wanted one: lane (i, kh)
// loads its 16 bytes (four features of one point) for the eight points 8 kh .. 8 kh + 7 of a 16-point step, and the
// split is register-local - feature t of those eight points is fragment t. 48 MFMAs (1 536 matrix-pipe cycles) per step
// and wave against ~200 vector instructions; two register sets, the next step's sixteen loads issued before this step's
// arithmetic.
// ---------------------------------------------------------------------------------------------
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void pair_split(float a0, float a1, unsigned& hi, unsigned& lo) {
    asm("v_cvt_pk_f16_f32 %0, %2, %3\n\t"
        "v_fma_mixlo_f16 %1, %0, -1.0, %2 op_sel_hi:[1,0,0]\n\t"
        "s_nop 0\n\t"
        "v_fma_mixhi_f16 %1, %0, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
        : "=&v"(hi), "=&v"(lo)
        : "v"(a0), "v"(a1));
}
__device__ __forceinline__ f32x16 mfma16h(const u32x4& a, const u32x4& b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, a), __builtin_bit_cast(h16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ int pair_scale_exponent(const unsigned* bits) {
    const float m = __uint_as_float(*bits);
    if (!(m > 0.0f) || !(m < __builtin_inff())) return 0;
    const int e = 14 - __builtin_amdgcn_frexp_expf(m);
    return e < -100 ? -100 : (e > 100 ? 100 : e);
}

struct PairSet {
    f32x4u a[8], b[8];
};

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void grad_batch_pair_kernel(const GradBatch b) {
    const GradJob& g = b.job[blockIdx.y];
    const int n_begin = g.n_begin, n_end = g.n_end, width = n_end - n_begin;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, kh = lane >> 5, i = lane & 31;
    const int wm = wave >> 1, wn = wave & 1;
    const int m_base = 128 * wm;
    if (m_base >= g.Mo) return;                 // (no barrier anywhere below)
    const int slice = blockIdx.x;
    const int c_base = n_begin + 128 * wn;       // this wave's first column
    if (c_base >= n_end) return;
    const int col = c_base + 4 * i;
    const int64_t p_begin = (int64_t)slice * b.pts_per_slice;
    int64_t p_end = p_begin + b.pts_per_slice;
    if (p_end > b.P) p_end = b.P;
    const int64_t n_pts = p_end > p_begin ? p_end - p_begin : 0;
    const int n_steps = (int)(n_pts / 16);

    const int ea = pair_scale_exponent(g.a_max), eb = pair_scale_exponent(g.b_max);
    const float sa = __builtin_ldexpf(1.0f, ea), sb = __builtin_ldexpf(1.0f, eb);

    f32x16 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][c][r] = 0.0f;
    float asum[4] = {0.0f, 0.0f, 0.0f, 0.0f};

    const float* pa = g.A + (p_begin + 8 * kh) * g.lda + m_base + 4 * i;
    const float* pb = g.B + (p_begin + 8 * kh) * g.ldb + col;
    const int64_t step_a = 16 * (int64_t)g.lda, step_b = 16 * (int64_t)g.ldb;

    auto load = [&](PairSet& r) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            r.a[j] = *(const f32x4u*)(pa + j * (int64_t)g.lda);
            r.b[j] = *(const f32x4u*)(pb + j * (int64_t)g.ldb);
        }
        pa += step_a;
        pb += step_b;
    };
    auto step = [&](const PairSet& r) {
        u32x4 ahi[4], alo[4], bhi[4], blo[4];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float a0 = r.a[2 * q][t], a1 = r.a[2 * q + 1][t];
                asum[t] += a0 + a1;
                unsigned hi, lo;
                pair_split(a0 * sa, a1 * sa, hi, lo);
                ahi[t][q] = hi;
                alo[t][q] = lo;
                pair_split(r.b[2 * q][t] * sb, r.b[2 * q + 1][t] * sb, hi, lo);
                bhi[t][q] = hi;
                blo[t][q] = lo;
            }
#pragma unroll
        for (int tm = 0; tm < 4; ++tm)
#pragma unroll
            for (int tn = 0; tn < 4; ++tn) {
                acc[tm][tn] = mfma16h(alo[tm], bhi[tn], acc[tm][tn]);
                acc[tm][tn] = mfma16h(ahi[tm], blo[tn], acc[tm][tn]);
                acc[tm][tn] = mfma16h(ahi[tm], bhi[tn], acc[tm][tn]);
            }
    };

    PairSet r0, r1;
    int s = 0;
    if (n_steps > 0) load(r0);
    for (; s + 2 <= n_steps; s += 2) {
        load(r1);
        step(r0);
        if (s + 2 < n_steps) load(r0);
        step(r1);
    }
    if (s < n_steps) {
        step(r0);
        ++s;
    }
    // the tail: fewer than sixteen points, loads predicated per lane and point
    const int64_t p_tail = p_begin + 16 * (int64_t)s;
    if (p_tail < p_end) {
        PairSet r;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int64_t p = p_tail + 8 * kh + j;
            const f32x4u z = {0.0f, 0.0f, 0.0f, 0.0f};
            r.a[j] = z;
            r.b[j] = z;
            if (p < p_end) {
                r.a[j] = *(const f32x4u*)(g.A + p * g.lda + m_base + 4 * i);
                r.b[j] = *(const f32x4u*)(g.B + p * g.ldb + col);
            }
        }
        step(r);
    }

    // acc[tm][tn][r] at lane (j = i, h = kh): row m_base + 4 (r&3 + 8 (r>>2) + 4 h) + tm, column c_base + 4 j + tn
    const float descale = __builtin_ldexpf(1.0f, -(ea + eb));
    float* part = g.part + (int64_t)slice * g.Mo * width - n_begin;
#pragma unroll
    for (int tm = 0; tm < 4; ++tm)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m_base + 4 * ((r & 3) + 8 * (r >> 2) + 4 * kh) + tm;
            float* dst = part + (int64_t)m * width + c_base + 4 * i;
            const f32x4u v = {acc[tm][0][r] * descale, acc[tm][1][r] * descale, acc[tm][2][r] * descale, acc[tm][3][r] * descale};
            *(f32x4u*)dst = v;
        }
    if (g.db && wn == 0) {
#pragma unroll
        for (int tm = 0; tm < 4; ++tm) {
            const float t = asum[tm] + __shfl_xor(asum[tm], 32);
            if (kh == 0) g.dbp[(int64_t)slice * g.Mo + m_base + 4 * i + tm] = t;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// The same kernel with its operands prefetched THROUGH LDS (global_load_lds_dwordx4: no destination registers, so the
// depth of the prefetch is no longer what the register file has left next to 256 accumulators). A wave owns kDmaSlots
// slots of 16 KiB - the sixteen 1 KiB loads of a step, each lane's 16 bytes at lane * 16 of its load's KiB, i.e. exactly
// what the lane reads back (no transposition, no sharing, no barrier) - and keeps kDmaSlots steps in flight: 32 KiB per
// wave, 128 KiB per CU, twice what two register sets held. Loads return in order, so "step s has landed" is
// vmcnt(16 x the steps issued after it).
// ---------------------------------------------------------------------------------------------
// BLK (GradJob::blocked = 3): both operands are blocked by 32 points (MlpStore::blocked), i.e. cut into KiB pieces of [32
// points][8 features]. A 16-point step then wants HALF of sixteen pieces per operand, and a load instruction fetches two such
// halves - lanes 0-31 the 512 bytes of piece j, lanes 32-63 those of piece j + 8 - which is as coalesced as the two rows of a
// row-major load. The pieces land verbatim ([16 points][2 half-waves][4 features] each), and the lane that computes with
// features 4 i .. 4 i + 3 reads its eight points back from piece i / 2 at a stride of 32 bytes. So that the 16 lanes a
// ds_read_b128 serves per cycle ({0-3, 12-15, 20-27}, ... : MI355X_MICROARCH.md, LDS) meet 16 different 16-byte bank slots,
// image j starts at j x (1024 + 32): the eight pieces j of a group then differ by 32 bytes modulo the 256-byte bank row, the
// two half-waves h by 16 (SQ_LDS_BANK_CONFLICT stays 0). Same loads per step, same bytes in flight, same registers after the
// read-back as the row-major form - only the addresses differ.
// (the operands' loads with the nt bit - they are read once - measured no faster: 299.9 against 301.1 it/s, profiles/r04_ab_notes.txt)
constexpr int kDmaSlots = 2;
constexpr int kDmaImage = 1024 + 32;           // one load instruction's KiB in LDS (BLK: padded, see above)
constexpr int kDmaRiderOff = 16 * kDmaImage;   // behind a step's sixteen images: its sixteen y values (GradJob::y), one 256-byte load
constexpr int kDmaSlotBytes = 16 * kDmaImage + 256;
constexpr size_t kPairDmaLds = (size_t)4 * kDmaSlots * kDmaSlotBytes;

template <int OFF>
__device__ __forceinline__ void dma_read(f32x4u& q, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(q) : "v"(addr), "n"(OFF) : "memory");
}
// point J of this lane's eight: row-major images hold one point (two, for the two half-waves) each, blocked ones eight features
// of sixteen points each (`addr` then points at this lane's piece, half-wave and point octet)
template <int J, bool BLK>
__device__ __forceinline__ void dma_read_set(PairSet& r, unsigned addr) {
    if constexpr (J < 8) {
        dma_read<BLK ? J * 32 : J * kDmaImage>(r.a[J], addr);
        dma_read<8 * kDmaImage + (BLK ? J * 32 : J * kDmaImage)>(r.b[J], addr);
        dma_read_set<J + 1, BLK>(r, addr);
    }
}

template <int U>
struct UnitTag {
    static constexpr int value = U;
};
template <int LO, int HI, class F>
__device__ __forceinline__ void unit_for(F&& f) {
    if constexpr (LO < HI) {
        f(UnitTag<LO>{});
        unit_for<LO + 1, HI>(f);
    }
}

// A step's operands split into fp16 (hi, lo) fragments: what its 48 MFMAs read
struct PairConv {
    u32x4 ahi[4], alo[4], bhi[4], blo[4];
};
// two pairs at once - one of each operand: two independent chains interleaved, so that no instruction reads what the one before it
// wrote and the two half-register writes of a pair's low halves are an instruction apart (what pair_split buys with an s_nop)
__device__ __forceinline__ void pair_split2(float a0, float a1, float b0, float b1, unsigned& ahi, unsigned& alo, unsigned& bhi,
                                            unsigned& blo) {
    asm("v_cvt_pk_f16_f32 %0, %4, %5\n\t"
        "v_cvt_pk_f16_f32 %2, %6, %7\n\t"
        "v_fma_mixlo_f16 %1, %0, -1.0, %4 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixlo_f16 %3, %2, -1.0, %6 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %1, %0, -1.0, %5 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %3, %2, -1.0, %7 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
        : "=&v"(ahi), "=&v"(alo), "=&v"(bhi), "=&v"(blo)
        : "v"(a0), "v"(a1), "v"(b0), "v"(b1));
}
typedef float f32x2u __attribute__((ext_vector_type(2)));
// a quad times a power of two, as two packed multiplies (hipcc chose scalar ones for two thirds of them)
__device__ __forceinline__ void scale_quad(f32x4u& x, float s) {
    f32x2u lo = {x[0], x[1]}, hi = {x[2], x[3]};
    const f32x2u ss = {s, s};
    asm("v_pk_mul_f32 %0, %0, %1" : "+v"(lo) : "v"(ss));
    asm("v_pk_mul_f32 %0, %0, %1" : "+v"(hi) : "v"(ss));
    x = f32x4u{lo[0], lo[1], hi[0], hi[1]};
}
// unit u of the sixteen a step's split is made of: the point pair (2 q, 2 q + 1), q = u / 4, feature t = u % 4, both operands.
// A pair's first unit (t = 0) adds its two points' dY to the bias sums (all four features: two packed adds each way, in the
// order the scalar loop added them) and scales both operands' quads IN PLACE (packed multiplies: powers of two); the other
// three find them scaled. The kernel issues one instruction per ~7 cycles with its one wave per SIMD - it is bound by their
// NUMBER (profiles/r04_ab_notes.txt) - and this is 9 per pair of values where the scalar form was 15.
__device__ __forceinline__ void pair_convert_unit(PairConv& c, PairSet& r, int u, float sa, float sb, f32x2u (&asum)[2], bool sums) {
    const int q = u >> 2, t = u & 3;
#ifndef NERF_EXP_DW_NOCONV
    if (t == 0) {
        f32x4u &x0 = r.a[2 * q], &x1 = r.a[2 * q + 1];
        // (all waves add, though only those of the first column half store: a branch per unit costs the ones that do more than
        // four instructions cost the others; volatile: hipcc otherwise gathers a whole loop body's sums behind its last MFMA, with
        // copies of the unscaled values)
        (void)sums;
        f32x2u lo, hi;
        asm volatile("v_pk_add_f32 %2, %4, %5\n\t"
                     "v_pk_add_f32 %3, %6, %7\n\t"
                     "v_pk_add_f32 %0, %0, %2\n\t"
                     "v_pk_add_f32 %1, %1, %3"
                     : "+v"(asum[0]), "+v"(asum[1]), "=&v"(lo), "=&v"(hi)
                     : "v"(f32x2u{x0[0], x0[1]}), "v"(f32x2u{x1[0], x1[1]}), "v"(f32x2u{x0[2], x0[3]}), "v"(f32x2u{x1[2], x1[3]}));
        scale_quad(x0, sa);
        scale_quad(x1, sa);
        scale_quad(r.b[2 * q], sb);
        scale_quad(r.b[2 * q + 1], sb);
    }
#endif
    unsigned ahi, alo, bhi, blo;
#ifdef NERF_EXP_DW_NOCONV      // ... without the (hi, lo) split ...
    ahi = __float_as_uint(r.a[2 * q][t]); alo = __float_as_uint(r.a[2 * q + 1][t]);
    bhi = __float_as_uint(r.b[2 * q][t]); blo = __float_as_uint(r.b[2 * q + 1][t]);
#else
    pair_split2(r.a[2 * q][t], r.a[2 * q + 1][t], r.b[2 * q][t], r.b[2 * q + 1][t], ahi, alo, bhi, blo);
#endif
    c.ahi[t][q] = ahi;
    c.alo[t][q] = alo;
    c.bhi[t][q] = bhi;
    c.blo[t][q] = blo;
}
// accumulator tile g = 4 tm + tn of the wave's sixteen: its three products, smallest first
__device__ __forceinline__ void pair_mma_tile(f32x16 (&acc)[4][4], const PairConv& c, int g) {
    const int tm = g >> 2, tn = g & 3;
#ifdef NERF_EXP_DW_NOMMA      // timing experiments (profiles/r04_ab_notes.txt): the step without its MFMAs ...
    asm volatile("" ::"v"(c.alo[tm]), "v"(c.bhi[tn]), "v"(c.ahi[tm]), "v"(c.blo[tn]));
    return;
#endif
    acc[tm][tn] = mfma16h(c.alo[tm], c.bhi[tn], acc[tm][tn]);
    acc[tm][tn] = mfma16h(c.ahi[tm], c.blo[tn], acc[tm][tn]);
    acc[tm][tn] = mfma16h(c.ahi[tm], c.bhi[tn], acc[tm][tn]);
}

template <bool BLK>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void grad_batch_pair_dma_kernel(const GradBatch b) {
    extern __shared__ __attribute__((aligned(16))) char dma_ring[];
    const GradJob& g = b.job[blockIdx.y];
    const int n_begin = g.n_begin, n_end = g.n_end, width = n_end - n_begin;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), kh = lane >> 5, i = lane & 31;
    const int wm = wave >> 1, wn = wave & 1;
    const int m_base = 128 * wm;
    if (m_base >= g.Mo) return;                 // (no barrier anywhere below)
    const int slice = blockIdx.x;
    const int c_base = n_begin + 128 * wn;       // this wave's first column
    if (c_base >= n_end) return;
    const int col = c_base + 4 * i;
#ifdef NERF_EXP_DW_SAMESLICE      // timing experiment (profiles/r04_ab_notes.txt): every workgroup reads slice 0's points - out of L2
    const int64_t p_begin = 0;
    int64_t p_end = b.pts_per_slice;
#else
    const int64_t p_begin = (int64_t)slice * b.pts_per_slice;
    int64_t p_end = p_begin + b.pts_per_slice;
#endif
    if (p_end > b.P) p_end = b.P;
    const int64_t n_pts = p_end > p_begin ? p_end - p_begin : 0;
    const int n_steps = (int)(n_pts / 16);

    const int ea = pair_scale_exponent(g.a_max), eb = pair_scale_exponent(g.b_max);
    const float sa = __builtin_ldexpf(1.0f, ea), sb = __builtin_ldexpf(1.0f, eb);

    f32x16 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][c][r] = 0.0f;
    f32x2u asum[2] = {{0.0f, 0.0f}, {0.0f, 0.0f}};      // bias sums of the wave's four row tiles (only the waves that store them add)
    const bool sums = g.db != nullptr && wn == 0;      // (wave-uniform)
    // the rider row (GradJob::y): the waves of the first row half hold X of their 128 columns anyway
    const bool has_y = g.y != nullptr && wm == 0;      // (wave-uniform)
    float ysum[4] = {0.0f, 0.0f, 0.0f, 0.0f}, ybias = 0.0f;
    const float* py = has_y ? g.y + (p_begin + (lane & 15)) * g.ldy : nullptr;
    const int64_t step_y = 16 * (int64_t)g.ldy;

    // row-major: load j of a step = points j (lanes 0-31) and 8 + j (lanes 32-63), features m_base + 4 i .. + 3.
    // BLK: load j = piece j (lanes 0-31) and piece 8 + j (lanes 32-63) of the wave's sixteen, the 16-point half the step is in;
    //      lane l < 32 fetches bytes 16 l .. of that half: (point l / 2, half-wave l % 2).
    const int b_feat0 = c_base - (BLK ? g.b_first : 0);      // BLK: B's feature index of this wave's first column
    const float* pa = BLK ? g.A + (p_begin >> 5) * (32 * (int64_t)g.Mo) + ((p_begin >> 4) & 1) * 128 + ((m_base >> 3) + 8 * kh) * 256 + 4 * i
                          : g.A + (p_begin + 8 * kh) * g.lda + m_base + 4 * i;
    const float* pb = BLK ? g.B + (p_begin >> 5) * (32 * (int64_t)256) + ((p_begin >> 4) & 1) * 128 + ((b_feat0 >> 3) + 8 * kh) * 256 + 4 * i
                          : g.B + (p_begin + 8 * kh) * g.ldb + col;
    const int64_t step_a = 16 * (int64_t)g.lda, step_b = 16 * (int64_t)g.ldb;
    const int64_t load_a = BLK ? 256 : (int64_t)g.lda, load_b = BLK ? 256 : (int64_t)g.ldb;      // from load j to load j + 1
    // BLK: a step moves on to the other half of the group (+ 128 floats) or to the first half of the next one
    const int64_t group_a = 32 * (int64_t)g.Mo - 128, group_b = 32 * (int64_t)256 - 128;
    int half = (int)((p_begin >> 4) & 1);
    char* my = dma_ring + (size_t)wave * kDmaSlots * kDmaSlotBytes;
    const unsigned my_lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)my;
    // where this lane reads back from: lane x 16 of every image, or (BLK) piece i / 2 = image (i / 2) % 8, its upper 512 bytes for
    // pieces 8..15, this lane's point octet (8 kh) and half-wave (i % 2)
    const unsigned my_addr = BLK ? my_lds + ((i >> 1) & 7) * kDmaImage + (i >> 4) * 512 + kh * 256 + (i & 1) * 16 : my_lds + lane * 16;

    // BLK: the addresses of a step's loads differ by the lane's constant part (one VGPR for both operands), a wave-uniform base per
    // operand that moves from step to step (scalar registers; + 4 KiB so that the eight loads' j KiB fit the signed immediate)
    // and the immediate - no vector instruction per load, where a pointer per lane cost a 64-bit add each. The immediate moves
    // the LDS address as well: M0 = the image's place minus it. (Every LDS-DMA of this instantiation is such a statement: hipcc
    // never owns M0 here.)
    const unsigned v_lane = (unsigned)(kh * 8192 + i * 16);
    const char* sa_base = nullptr;
    const char* sb_base = nullptr;
    if constexpr (BLK) {
        sa_base = (const char*)(g.A + (p_begin >> 5) * (32 * (int64_t)g.Mo) + ((p_begin >> 4) & 1) * 128 + (m_base >> 3) * 256) + 4096;
        sb_base = (const char*)(g.B + (p_begin >> 5) * (32 * (int64_t)256) + ((p_begin >> 4) & 1) * 128 + (b_feat0 >> 3) * 256) + 4096;
    }
    const unsigned v_lane_y = (unsigned)((lane & 15) * g.ldy * 4);
    const char* sy_base = has_y ? (const char*)(g.y + p_begin * g.ldy) : nullptr;
    // load u of the sixteen of the step at (pa, pb) into `slot`: a's eight in the order the old loop issued them pairwise with b's
    auto issue_one = [&](int slot, auto U) {
        constexpr int u = decltype(U)::value, j = u >> 1;
        char* base = my + slot * kDmaSlotBytes;
#ifdef NERF_EXP_DW_NOLOAD      // ... without its loads (the LDS images stay what they were)
        return;
#endif
#ifdef NERF_EXP_DW_NODUP       // ... with only the half of its loads no other wave of the workgroup issues too (timing of a shared ring)
        if (((u & 1) == 0 && (j >> 2) != wn) || ((u & 1) == 1 && (j >> 2) != wm)) return;
#endif
        if constexpr (BLK) {
            const unsigned m0 = my_lds + slot * kDmaSlotBytes + ((u & 1) ? 8 + j : j) * kDmaImage - (j * 1024 - 4096);
            asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1 offset:%3"
                         : : "v"(v_lane), "s"((u & 1) ? sb_base : sa_base), "s"(m0), "n"(j * 1024 - 4096) : "memory");
            return;
        }
        if ((u & 1) == 0)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(pa + j * load_a),
                                             (__attribute__((address_space(3))) void*)(base + j * kDmaImage), 16, 0, 0);
        else
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(pb + j * load_b),
                                             (__attribute__((address_space(3))) void*)(base + 8 * kDmaImage + j * kDmaImage), 16, 0, 0);
    };
    auto issue_done = [&](int slot) {      // behind the sixteen: the rider's load, the pointers on to the next step
#ifdef NERF_EXP_DW_NOLOAD
        if (false)
#else
        if (has_y)      // lane l fetches y of point (l & 15) of the step: LDS [16 floats] x 4 copies
#endif
        {
            if constexpr (BLK) {
                asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dword %0, %1"
                             : : "v"(v_lane_y), "s"(sy_base), "s"(my_lds + slot * kDmaSlotBytes + kDmaRiderOff) : "memory");
                sy_base += 4 * step_y;
            } else {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)py,
                                                 (__attribute__((address_space(3))) void*)(my + slot * kDmaSlotBytes + kDmaRiderOff), 4, 0, 0);
                py += step_y;
            }
        }
        if constexpr (BLK) {
            sa_base += 4 * (half ? group_a : 128);
            sb_base += 4 * (half ? group_b : 128);
            half ^= 1;
        } else {
            pa += step_a;
            pb += step_b;
        }
    };
    auto issue = [&](int slot) {      // the sixteen loads of the step at (pa, pb) into `slot`
        unit_for<0, 16>([&](auto U) { issue_one(slot, U); });
        issue_done(slot);
    };
    auto ride = [&](const PairSet& r, const f32x4u& y0, const f32x4u& y1) {      // y of this lane's eight points
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        f32x2 s01 = {ysum[0], ysum[1]}, s23 = {ysum[2], ysum[3]};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float yj = j < 4 ? y0[j] : y1[j - 4];
            const f32x2 yy = {yj, yj};
            // (v_pk_fma_f32: two columns per instruction)
            s01 = __builtin_elementwise_fma(yy, f32x2{r.b[j][0], r.b[j][1]}, s01);
            s23 = __builtin_elementwise_fma(yy, f32x2{r.b[j][2], r.b[j][3]}, s23);
            ybias += yj;
        }
        ysum[0] = s01[0]; ysum[1] = s01[1]; ysum[2] = s23[0]; ysum[3] = s23[1];
    };
    auto step = [&](const PairSet& r) {
        u32x4 ahi[4], alo[4], bhi[4], blo[4];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float a0 = r.a[2 * q][t], a1 = r.a[2 * q + 1][t];
                asum[t >> 1][t & 1] += a0 + a1;
                unsigned hi, lo;
                pair_split(a0 * sa, a1 * sa, hi, lo);
                ahi[t][q] = hi;
                alo[t][q] = lo;
                pair_split(r.b[2 * q][t] * sb, r.b[2 * q + 1][t] * sb, hi, lo);
                bhi[t][q] = hi;
                blo[t][q] = lo;
            }
#pragma unroll
        for (int tm = 0; tm < 4; ++tm)
#pragma unroll
            for (int tn = 0; tn < 4; ++tn) {
                acc[tm][tn] = mfma16h(alo[tm], bhi[tn], acc[tm][tn]);
                acc[tm][tn] = mfma16h(ahi[tm], blo[tn], acc[tm][tn]);
                acc[tm][tn] = mfma16h(ahi[tm], bhi[tn], acc[tm][tn]);
            }
    };

    // A step = 48 MFMAs (1 536 matrix-pipe cycles) + its preparation: sixteen reads back from LDS, sixteen loads of the step two
    // ahead into the slot just read, ~240 vector instructions of scaling and (hi, lo) splitting. One after the other - what hipcc
    // makes of a loop body written that way, with one wave per SIMD and nobody else to issue - a step took ~4 000 cycles and the
    // matrix pipe was busy 38 % of them: the kernel ran at 4.8 TB/s of the 6.0 the chip reads at (profiles/microbench/
    // hbm_read_rate.hip); pipelined it runs at 5.3 (profiles/r04_train_pmc_summary.md). So the loop is software-pipelined by hand: while step s's MFMAs run from one register set (PairConv), step
    // s + 1 is awaited, read back and split into the other, a slice of that work behind every accumulator tile's three MFMAs,
    // the slices fenced (sched_barrier) so that they stay where they are put. Same products into the same accumulators in the
    // same order: bit-identical sums.
    // wait: step k has landed when at most the loads of the step issued after it are outstanding
    auto await = [&](bool newer_in_flight) {
#ifdef NERF_EXP_DW_NOLOAD
        return;
#endif
        if (!newer_in_flight) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef NERF_EXP_DW_NODUP
        else if (has_y) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
#else
        else if (has_y) asm volatile("s_waitcnt vmcnt(17)" ::: "memory");      // (a step of a wave with the rider is 17 loads)
        else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
#endif
    };
    auto read_back = [&](PairSet& r, f32x4u& y0, f32x4u& y1, int slot) {
        dma_read_set<0, BLK>(r, my_addr + slot * kDmaSlotBytes);
        if (has_y) {
            const unsigned ya = my_lds + slot * kDmaSlotBytes + kh * 32;
            dma_read<kDmaRiderOff>(y0, ya);
            dma_read<kDmaRiderOff + 16>(y1, ya);
        }
    };
    // one step's preparation in the open (the first step, the last three): `more` = the step two ahead exists
    auto prepare = [&](PairConv& c, int slot, bool newer_in_flight, bool more) {
        __builtin_amdgcn_sched_barrier(0);
        await(newer_in_flight);
        PairSet r;
        f32x4u y0 = {0.0f, 0.0f, 0.0f, 0.0f}, y1 = y0;
        read_back(r, y0, y1, slot);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (read back: the slot may be overwritten)
        __builtin_amdgcn_sched_barrier(0);
        if (more) issue(slot);
        __builtin_amdgcn_sched_barrier(0);
        if (has_y) ride(r, y0, y1);
#pragma unroll
        for (int u = 0; u < 16; ++u) pair_convert_unit(c, r, u, sa, sb, asum, sums);
    };
    auto mma_all = [&](const PairConv& c) {
#pragma unroll
        for (int g = 0; g < 16; ++g) pair_mma_tile(acc, c, g);
    };
    // the steady state: step s's MFMAs from `cur`; step s + 1 (in `slot`, step s + 2 in flight behind it) into `nxt`; the loads
    // of step s + 3 into `slot`
    auto overlap = [&](const PairConv& cur, PairConv& nxt, int slot) {
        __builtin_amdgcn_sched_barrier(0);
        await(true);
        PairSet r;
        f32x4u y0 = {0.0f, 0.0f, 0.0f, 0.0f}, y1 = y0;
        read_back(r, y0, y1, slot);
        pair_mma_tile(acc, cur, 0);      // (two tiles' MFMAs cover the reads' latency)
        pair_mma_tile(acc, cur, 1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#ifdef NERF_EXP_DW_EARLY      // timing experiment: the loads of step s + 3 all at once, as soon as the slot is free
        issue(slot);
        __builtin_amdgcn_sched_barrier(0);
#endif
        if (has_y) ride(r, y0, y1);
        unit_for<2, 16>([&](auto G) {
            constexpr int g = decltype(G)::value;
            pair_mma_tile(acc, cur, g);
            // sixteen units of splitting and sixteen loads behind fourteen tiles: two each behind the first two
            constexpr int u0 = g < 4 ? 2 * (g - 2) : g;
            pair_convert_unit(nxt, r, u0, sa, sb, asum, sums);
#ifndef NERF_EXP_DW_EARLY
            issue_one(slot, UnitTag<u0>{});
#endif
            if constexpr (g < 4) {
                pair_convert_unit(nxt, r, u0 + 1, sa, sb, asum, sums);
#ifndef NERF_EXP_DW_EARLY
                issue_one(slot, UnitTag<u0 + 1>{});
#endif
            }
#ifndef NERF_EXP_DW_EARLY
            if constexpr (g == 15) issue_done(slot);
#endif
            __builtin_amdgcn_sched_barrier(0);
        });
    };

    if (n_steps > 0) {
        PairConv c0, c1;
        issue(0);
        if (n_steps > 1) issue(1);
        prepare(c0, 0, n_steps > 1, n_steps > 2);
        int s = 0;
        // here: step s split in c0, steps s + 1 and s + 2 in flight (slots 1 and 0); a pass through the body issues s + 3 and s + 4
        for (; s + 4 < n_steps; s += 2) {
            overlap(c0, c1, 1);
            overlap(c1, c0, 0);
        }
        // the last four steps or fewer, one thing after the other
        mma_all(c0);
        if (s + 1 < n_steps) {
            prepare(c1, 1, s + 2 < n_steps, s + 3 < n_steps);
            mma_all(c1);
        }
        if (s + 2 < n_steps) {
            prepare(c0, 0, s + 3 < n_steps, false);
            mma_all(c0);
        }
        if (s + 3 < n_steps) {
            prepare(c1, 1, false, false);
            mma_all(c1);
        }
    }
    // the tail: fewer than sixteen points, loads predicated per lane and point
    const int64_t p_tail = p_begin + 16 * (int64_t)n_steps;
    if (p_tail < p_end) {
        PairSet r;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int64_t p = p_tail + 8 * kh + j;
            const f32x4u z = {0.0f, 0.0f, 0.0f, 0.0f};
            r.a[j] = z;
            r.b[j] = z;
            if (p < p_end) {
                if constexpr (BLK) {      // piece (feature / 8) of p's group, 32 bytes per point, 16 per half-wave
                    r.a[j] = *(const f32x4u*)(g.A + (p >> 5) * (32 * (int64_t)g.Mo) + ((m_base + 4 * i) >> 3) * 256 + (int)(p & 31) * 8 + (i & 1) * 4);
                    r.b[j] = *(const f32x4u*)(g.B + (p >> 5) * (32 * (int64_t)256) + ((b_feat0 + 4 * i) >> 3) * 256 + (int)(p & 31) * 8 + (i & 1) * 4);
                } else {
                    r.a[j] = *(const f32x4u*)(g.A + p * g.lda + m_base + 4 * i);
                    r.b[j] = *(const f32x4u*)(g.B + p * g.ldb + col);
                }
            }
        }
        if (has_y) {
            f32x4u y0 = {0.0f, 0.0f, 0.0f, 0.0f}, y1 = y0;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int64_t p = p_tail + 8 * kh + j;
                const float v = p < p_end ? g.y[p * g.ldy] : 0.0f;
                if (j < 4) y0[j] = v;
                else y1[j - 4] = v;
            }
            ride(r, y0, y1);
        }
        step(r);
    }

    const float descale = __builtin_ldexpf(1.0f, -(ea + eb));
    float* part = g.part + (int64_t)slice * g.Mo * width - n_begin;
#pragma unroll
    for (int tm = 0; tm < 4; ++tm)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m_base + 4 * ((r & 3) + 8 * (r >> 2) + 4 * kh) + tm;
            float* dst = part + (int64_t)m * width + c_base + 4 * i;
            const f32x4u v = {acc[tm][0][r] * descale, acc[tm][1][r] * descale, acc[tm][2][r] * descale, acc[tm][3][r] * descale};
            *(f32x4u*)dst = v;
        }
    if (g.db && wn == 0) {
#pragma unroll
        for (int tm = 0; tm < 4; ++tm) {
            const float t = asum[tm >> 1][tm & 1] + __shfl_xor(asum[tm >> 1][tm & 1], 32);
            if (kh == 0) g.dbp[(int64_t)slice * g.Mo + m_base + 4 * i + tm] = t;
        }
    }
    if (has_y) {      // the two half-waves hold the two halves of every step's points
        f32x4u v;
#pragma unroll
        for (int t = 0; t < 4; ++t) v[t] = ysum[t] + __shfl_xor(ysum[t], 32);
        if (kh == 0) *(f32x4u*)(g.y_part + (int64_t)slice * width + (c_base - n_begin) + 4 * i) = v;
        const float tb = ybias + __shfl_xor(ybias, 32);
        if (wn == 0 && lane == 0) g.y_dbp[slice] = tb;
    }
}

// ---------------------------------------------------------------------------------------------
// The gamma(x) / gamma(d) columns' weight gradients on the fp16 pipe (blocked passes; round 4). dW[:, columns of the encoded
// input] = dY^T gamma for layer 0, the skip layer and the view layer reads dY - a kilobyte per point - a second time for 63
// (or 27) columns of X: on the fp32 matrix pipe (grad_batch_kernel<1>) that was 0.30 ms per iteration at 80 % of the pipe, and
// 0.37 once dY was blocked (its register loads became 32-byte gathers). Here a WAVE is a slice: 128 rows x 64 columns (128
// accumulators), dY through the padded LDS images of grad_batch_pair_dma_kernel<true> (eight coalesced KiB per 16-point
// step), gamma - row-major [points, 64], zero-padded - as four more LDS-DMA KiB, two steps in flight, no barrier; per step 24
// v_mfma_f32_32x32x16_f16 against 12 KiB of operands: the kernel is bound by reading dY once more, not by arithmetic. The waves of
// a workgroup are 2 row halves x 2 slices (256 rows) or 4 slices (128 rows); slices are summed by grad_batch_reduce_kernel.
// Arithmetic: the exact (hi, lo) split of both operands, one power-of-two scale per tensor (dY: the maxima the backward-data
// kernel tracks; gamma(x): the range the forward kernel measures on its encoded inputs; gamma(d): 1), three products per term.
// ---------------------------------------------------------------------------------------------
constexpr int kNpSlots = 2;
constexpr int kNpCols = 64;                                   // X columns per job (row stride of the gamma buffers)
constexpr int kNpSlotBytes = 8 * kDmaImage + 16 * kNpCols * 4;   // a step's eight dY images + its [16 points][64 columns] of X
constexpr size_t kNarrowPairLds = (size_t)4 * kNpSlots * kNpSlotBytes;

template <int OFF>
__device__ __forceinline__ void np_read32(float& v, unsigned addr) {
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=&v"(v) : "v"(addr), "n"(OFF) : "memory");
}
template <int J>
__device__ __forceinline__ void np_read_set(f32x4u (&a)[8], float (&x)[2][8], unsigned a_addr, unsigned x_addr) {
    if constexpr (J < 8) {
        dma_read<J * 32>(a[J], a_addr);
        np_read32<8 * kDmaImage + J * kNpCols * 4>(x[0][J], x_addr);
        np_read32<8 * kDmaImage + J * kNpCols * 4 + 128>(x[1][J], x_addr);
        np_read_set<J + 1>(a, x, a_addr, x_addr);
    }
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void grad_batch_narrow_pair_kernel(const GradBatch b) {
    extern __shared__ __attribute__((aligned(16))) char np_ring[];
    const GradJob& g = b.job[blockIdx.y];
    const int width = g.n_end - g.n_begin;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), kh = lane >> 5, i = lane & 31;
    const int wph = g.Mo > 128 ? 2 : 4;                    // waves per row half: each takes a share of the workgroup's points
    const int wm = wave / wph, ws = wave - wm * wph;
    const int slice = blockIdx.x;                           // a workgroup is a slice; its waves' sums are added up in LDS at the end
    if (slice >= b.n_slices) return;                        // (uniform over the workgroup)
    const int m_base = 128 * wm;
    const int64_t sub = b.pts_per_slice / wph;              // (a multiple of 32: launch_grad_batch_narrow_pair)
    const int64_t p_begin = (int64_t)slice * b.pts_per_slice + ws * sub;
    int64_t p_end = p_begin + sub;
    if (p_end > b.P) p_end = b.P;
    const int64_t n_pts = p_end > p_begin ? p_end - p_begin : 0;
    const int n_steps = (int)(n_pts / 16);

    const int ea = pair_scale_exponent(g.a_max), eb = pair_scale_exponent(g.b_max);
    const float sa = __builtin_ldexpf(1.0f, ea), sb = __builtin_ldexpf(1.0f, eb);

    f32x16 acc[4][2];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][c][r] = 0.0f;
    float asum[4] = {0.0f, 0.0f, 0.0f, 0.0f};

    // dY blocked by 32 points: load j = piece j (lanes 0-31) and piece 8 + j (lanes 32-63) of this row half's sixteen, the
    // 16-point half the step is in (grad_batch_pair_dma_kernel<true>); X: [16 points][64 floats] = four contiguous KiB
    const float* pa = g.A + (p_begin >> 5) * (32 * (int64_t)g.Mo) + ((p_begin >> 4) & 1) * 128 + ((m_base >> 3) + 8 * kh) * 256 + 4 * i;
    const float* px = g.B + p_begin * kNpCols + 4 * lane;
    const int64_t group_a = 32 * (int64_t)g.Mo - 128;
    int half = (int)((p_begin >> 4) & 1);
    char* my = np_ring + (size_t)wave * kNpSlots * kNpSlotBytes;
    const unsigned my_lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)my;
    const unsigned a_addr = my_lds + ((i >> 1) & 7) * kDmaImage + (i >> 4) * 512 + kh * 256 + (i & 1) * 16;
    const unsigned x_addr = my_lds + kh * (8 * kNpCols * 4) + i * 4;      // points 8 kh .., column i (and 32 + i)

    auto issue = [&](int slot) {
        char* base = my + slot * kNpSlotBytes;
#pragma unroll
        for (int j = 0; j < 8; ++j)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(pa + j * 256),
                                             (__attribute__((address_space(3))) void*)(base + j * kDmaImage), 16, 0, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(px + j * 256),
                                             (__attribute__((address_space(3))) void*)(base + 8 * kDmaImage + j * 1024), 16, 0, 0);
        pa += half ? group_a : 128;
        half ^= 1;
        px += 16 * kNpCols;
    };
    auto step = [&](const f32x4u (&ra)[8], const float (&rx)[2][8]) {
        u32x4 ahi[4], alo[4], bhi[2], blo[2];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float a0 = ra[2 * q][t], a1 = ra[2 * q + 1][t];
                asum[t] += a0 + a1;
                unsigned hi, lo;
                pair_split(a0 * sa, a1 * sa, hi, lo);
                ahi[t][q] = hi;
                alo[t][q] = lo;
            }
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                unsigned hi, lo;
                pair_split(rx[c][2 * q] * sb, rx[c][2 * q + 1] * sb, hi, lo);
                bhi[c][q] = hi;
                blo[c][q] = lo;
            }
        }
#pragma unroll
        for (int tm = 0; tm < 4; ++tm)
#pragma unroll
            for (int tn = 0; tn < 2; ++tn) {
                acc[tm][tn] = mfma16h(alo[tm], bhi[tn], acc[tm][tn]);
                acc[tm][tn] = mfma16h(ahi[tm], blo[tn], acc[tm][tn]);
                acc[tm][tn] = mfma16h(ahi[tm], bhi[tn], acc[tm][tn]);
            }
    };

    for (int t = 0; t < kNpSlots && t < n_steps; ++t) issue(t);
    int slot = 0;
    for (int s = 0; s < n_steps; ++s) {
        __builtin_amdgcn_sched_barrier(0);
        // step s has landed when at most the twelve loads of the step issued after it are outstanding
        if (s + 1 >= n_steps) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        f32x4u ra[8];
        float rx[2][8];
        np_read_set<0>(ra, rx, a_addr + slot * kNpSlotBytes, x_addr + slot * kNpSlotBytes);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (read back: the slot may be overwritten)
        __builtin_amdgcn_sched_barrier(0);
        if (s + kNpSlots < n_steps) issue(slot);
        __builtin_amdgcn_sched_barrier(0);
        step(ra, rx);
        slot = slot + 1 == kNpSlots ? 0 : slot + 1;
    }
    // the tail: fewer than sixteen points, loads predicated per lane and point
    const int64_t p_tail = p_begin + 16 * (int64_t)n_steps;
    if (p_tail < p_end) {
        f32x4u ra[8];
        float rx[2][8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int64_t p = p_tail + 8 * kh + j;
            const f32x4u z = {0.0f, 0.0f, 0.0f, 0.0f};
            ra[j] = z;
            rx[0][j] = rx[1][j] = 0.0f;
            if (p < p_end) {
                ra[j] = *(const f32x4u*)(g.A + (p >> 5) * (32 * (int64_t)g.Mo) + ((m_base + 4 * i) >> 3) * 256 + (int)(p & 31) * 8 + (i & 1) * 4);
                rx[0][j] = g.B[p * kNpCols + i];
                rx[1][j] = g.B[p * kNpCols + 32 + i];
            }
        }
        step(ra, rx);
    }

    // ---- the waves of a row half add their sums up (a partial per workgroup, not per wave: the pass over the partials is
    //      latency-bound on their number): the upper half of them hands its registers over through the ring's memory, twice
    //      for four waves. Fixed order, so the result does not depend on timing. ----
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    for (int span = wph >> 1; span >= 1; span >>= 1) {
        __syncthreads();
        float* box = (float*)np_ring + (size_t)(wm * (wph >> 1) + (ws - span)) * (132 * 64);      // 128 sums + 4 bias sums per lane
        if (ws >= span && ws < 2 * span) {
#pragma unroll
            for (int tm = 0; tm < 4; ++tm) {
#pragma unroll
                for (int tn = 0; tn < 2; ++tn)
#pragma unroll
                    for (int r = 0; r < 16; ++r) box[((tm * 2 + tn) * 16 + r) * 64 + lane] = acc[tm][tn][r];
                box[(128 + tm) * 64 + lane] = asum[tm];
            }
        }
        __syncthreads();
        if (ws < span) {
            const float* from = (const float*)np_ring + (size_t)(wm * (wph >> 1) + ws) * (132 * 64);
#pragma unroll
            for (int tm = 0; tm < 4; ++tm) {
#pragma unroll
                for (int tn = 0; tn < 2; ++tn)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[tm][tn][r] += from[((tm * 2 + tn) * 16 + r) * 64 + lane];
                asum[tm] += from[(128 + tm) * 64 + lane];
            }
        }
    }
    if (ws != 0) return;
    // acc[tm][tn][r] at lane (n = i, h = kh): row m_base + 4 (r&3 + 8 (r>>2) + 4 h) + tm, column 32 tn + n
    const float descale = __builtin_ldexpf(1.0f, -(ea + eb));
    float* part = g.part + (int64_t)slice * g.Mo * width;
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) {
        const int c = 32 * tn + i;
        if (c < width) {
#pragma unroll
            for (int tm = 0; tm < 4; ++tm)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m_base + 4 * ((r & 3) + 8 * (r >> 2) + 4 * kh) + tm;
                    part[(int64_t)m * width + c] = acc[tm][tn][r] * descale;
                }
        }
    }
    if (g.db) {
#pragma unroll
        for (int tm = 0; tm < 4; ++tm) {
            const float t = asum[tm] + __shfl_xor(asum[tm], 32);
            if (kh == 0) g.dbp[(int64_t)slice * g.Mo + m_base + 4 * i + tm] = t;
        }
    }
}

// part[s][m][c] summed over the slices in order (deterministic) into dW[m][n_begin + c]; thread = four consecutive elements
// of one job (eight 16-byte loads in flight), the bias gradients behind them. A job whose operands were in the units of the
// row-equalised network scales its sums by 2^(row_exp[m] - col_exp[n]) here (GradJob): the plain parameters' gradient.
__global__ __launch_bounds__(256) void grad_batch_reduce_kernel(const GradBatch b) {
    const GradJob& g = b.job[blockIdx.y];
    const int width = g.n_end - g.n_begin;
    const int64_t n_w = (int64_t)g.Mo * width;
    const int64_t quads = (n_w + 3) / 4;
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t < quads) {
        const int64_t e0 = 4 * t;
        const bool whole = e0 + 4 <= n_w && (n_w & 3) == 0;
        float sum[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        int k = 0;
        if (whole) {
            for (; k + 8 <= b.n_slices; k += 8) {
                f32x4u v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = *(const f32x4u*)(g.part + (int64_t)(k + u) * n_w + e0);
#pragma unroll
                for (int u = 0; u < 8; ++u)
#pragma unroll
                    for (int c = 0; c < 4; ++c) sum[c] += v[u][c];
            }
        }
        for (; k < b.n_slices; ++k)
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (e0 + c < n_w) sum[c] += g.part[(int64_t)k * n_w + e0 + c];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int64_t idx = e0 + c;
            if (idx < n_w) {
                const int m = (int)(idx / width), n = g.n_begin + (int)(idx % width);
                const float v = __builtin_ldexpf(sum[c], g.ex.of(m, n));
                float* w = g.dW + (int64_t)m * g.ldw + n;
                *w = b.accumulate ? *w + v : v;   // second pass through a shared network: .grad accumulates (nerf.ipynb:1270)
            }
        }
    } else if (g.db && t - quads < g.Mo) {
        const int m = (int)(t - quads);
        float sacc = 0.0f;
        for (int k = 0; k < b.n_slices; ++k) sacc += g.dbp[(int64_t)k * g.Mo + m];
        sacc = __builtin_ldexpf(sacc, g.ex.of_row(m));
        g.db[m] = b.accumulate ? g.db[m] + sacc : sacc;
    }
}

bool grad_pair_takes_riders() {      // NERF_TRAIN_DW_DMA=0 keeps the operands' prefetch in registers (A/B; that kernel takes no rider)
    static const bool through_lds = [] {
        const char* e = getenv("NERF_TRAIN_DW_DMA");
        return !(e && *e == '0');
    }();
    return through_lds;
}

static hipError_t launch_grad_batch_impl(GradBatch& b, bool wide, float* part, size_t part_floats, float* dbp, size_t dbp_floats,
                                         hipStream_t s, bool pair, const GradRider* rider);

hipError_t launch_grad_batch(GradBatch& b, bool wide, float* part, size_t part_floats, float* dbp, size_t dbp_floats,
                             hipStream_t s, bool pair) {
    return launch_grad_batch_impl(b, wide, part, part_floats, dbp, dbp_floats, s, pair, nullptr);
}

// the narrow jobs of a blocked pass on the fp16 pipe (grad_batch_narrow_pair_kernel): a workgroup is a slice
hipError_t launch_grad_batch_narrow_pair(GradBatch& b, float* part, size_t part_floats, float* dbp, size_t dbp_floats, hipStream_t s) {
    if (b.n <= 0) return hipSuccess;
    if (b.n > kMaxGradJobs || b.n_slices <= 0 || (b.pts_per_slice & 127)) return hipErrorInvalidValue;      // (four shares of whole groups of 32)
    size_t used = 0, used_db = 0;
    int64_t max_threads = 0;
    int max_blocks = 1;
    for (int j = 0; j < b.n; ++j) {
        GradJob& g = b.job[j];
        const int width = g.n_end - g.n_begin;
        if (g.Mo % 128 != 0 || g.Mo > 256 || width <= 0 || width > kNpCols || g.ldb != kNpCols || g.blocked != 1 || !g.a_max ||
            !g.b_max || (reinterpret_cast<uintptr_t>(g.B) & 15))
            return hipErrorInvalidValue;
        g.part = part + used;
        g.dbp = dbp + used_db;
        used += (size_t)b.n_slices * g.Mo * width;
        if (g.db) used_db += (size_t)b.n_slices * g.Mo;
        const int64_t th = ((int64_t)g.Mo * width + 3) / 4 + g.Mo;
        max_threads = th > max_threads ? th : max_threads;
        max_blocks = b.n_slices;
    }
    if (used > part_floats || used_db > dbp_floats) return hipErrorInvalidValue;
    static bool raised[64] = {};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 64 && !raised[dev]) {
        e = hipFuncSetAttribute((const void*)grad_batch_narrow_pair_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kNarrowPairLds);
        if (e != hipSuccess) return e;
        raised[dev] = true;
    }
    hipLaunchKernelGGL(grad_batch_narrow_pair_kernel, dim3((unsigned)max_blocks, (unsigned)b.n), dim3(256), kNarrowPairLds, s, b);
    hipLaunchKernelGGL(grad_batch_reduce_kernel, dim3((unsigned)((max_threads + 255) / 256), (unsigned)b.n), dim3(256), 0, s, b);
    return hipGetLastError();
}
hipError_t launch_grad_batch_with_rider(GradBatch& b, const GradRider& r, float* part, size_t part_floats, float* dbp,
                                        size_t dbp_floats, hipStream_t s) {
    return launch_grad_batch_impl(b, true, part, part_floats, dbp, dbp_floats, s, true, &r);
}

static hipError_t launch_grad_batch_impl(GradBatch& b, bool wide, float* part, size_t part_floats, float* dbp, size_t dbp_floats,
                             hipStream_t s, bool pair, const GradRider* rider) {
    if (b.n <= 0) return hipSuccess;
    if (b.n > kMaxGradJobs || b.n_slices <= 0) return hipErrorInvalidValue;
    if (rider && (!grad_pair_takes_riders() || rider->job < 0 || rider->job >= b.n || b.n >= kMaxGradJobs || !b.job[rider->job].y))
        return hipErrorInvalidValue;
    size_t used = 0, used_db = 0;
    int64_t max_threads = 0;
    for (int j = 0; j < b.n; ++j) {
        GradJob& g = b.job[j];
        const int width = g.n_end - g.n_begin;
        if (g.Mo % 128 != 0 || g.Mo > 256 || width <= 0 || (wide ? width != 256 : width > 64)) return hipErrorInvalidValue;
        if (pair && (!wide || !g.a_max || !g.b_max || (g.lda & 3) || (g.ldb & 3))) return hipErrorInvalidValue;
        g.part = part + used;
        g.dbp = dbp + used_db;
        used += (size_t)b.n_slices * g.Mo * width;
        if (g.db) used_db += (size_t)b.n_slices * g.Mo;      // (only the job that owns the bias gradient writes bias partials)
        const int64_t th = ((int64_t)g.Mo * width + 3) / 4 + g.Mo;
        max_threads = th > max_threads ? th : max_threads;
    }
    int n_all = b.n;
    if (rider) {      // the rider's partial sums, and a reduce-only entry behind the real jobs that adds them up
        GradJob& g = b.job[rider->job];
        const int width = g.n_end - g.n_begin;
        g.y_part = part + used;
        used += (size_t)b.n_slices * width;
        g.y_dbp = dbp + used_db;
        used_db += (size_t)b.n_slices;
        GradJob& x = b.job[n_all++];
        x = GradJob{};
        x.Mo = 1;
        x.n_begin = 0;
        x.n_end = width;
        x.dW = rider->dW;
        x.ldw = width;
        x.db = rider->db;
        x.part = g.y_part;
        x.dbp = g.y_dbp;
        x.ex = rider->ex;
    }
    if (used > part_floats || used_db > dbp_floats) return hipErrorInvalidValue;
    const dim3 grid((unsigned)b.n_slices, (unsigned)b.n);
    if (pair && grad_pair_takes_riders()) {
        static bool raised[64] = {};
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return e;
        if (dev >= 0 && dev < 64 && !raised[dev]) {
            e = hipFuncSetAttribute((const void*)grad_batch_pair_dma_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kPairDmaLds);
            if (e == hipSuccess)
                e = hipFuncSetAttribute((const void*)grad_batch_pair_dma_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kPairDmaLds);
            if (e != hipSuccess) return e;
            raised[dev] = true;
        }
        // a batch is blocked by 32 points as a whole (every job's A and B) or not at all
        int n_blk = 0;
        for (int j = 0; j < b.n; ++j) n_blk += b.job[j].blocked == 3 ? 1 : 0;
        if (n_blk != 0 && n_blk != b.n) return hipErrorInvalidValue;
        if (n_blk) hipLaunchKernelGGL(grad_batch_pair_dma_kernel<true>, grid, dim3(256), kPairDmaLds, s, b);
        else hipLaunchKernelGGL(grad_batch_pair_dma_kernel<false>, grid, dim3(256), kPairDmaLds, s, b);
    } else {
        for (int j = 0; j < b.n; ++j)      // (only the LDS-prefetch kernel reads blocked X; the fp32 kernels read blocked dY)
            if ((b.job[j].blocked & 2) || (pair && b.job[j].blocked)) return hipErrorInvalidValue;
        if (pair) hipLaunchKernelGGL(grad_batch_pair_kernel, grid, dim3(256), 0, s, b);
        else if (wide) hipLaunchKernelGGL(grad_batch_kernel<4>, grid, dim3(256), 0, s, b);
        else hipLaunchKernelGGL(grad_batch_kernel<1>, grid, dim3(256), 0, s, b);
    }
    hipLaunchKernelGGL(grad_batch_reduce_kernel, dim3((unsigned)((max_threads + 255) / 256), (unsigned)n_all), dim3(256), 0, s, b);
    return hipGetLastError();
}

}  // namespace nerf
