// streaming.hip -- fused multi-input streaming merges (HBM-bound elementwise kernels).
//
// One kernel template covers cast / weighted_sum / n_average / add_difference /
// subtract / scale for one key (a flat tensor) or for a whole arena of keys
// (the batched form: the arithmetic does not depend on the key, so the arena is
// swept as ONE flat range and the key is looked up only on the rare non-finite path).
//
// Roofline: HBM.  Algorithmic bytes per element = sum(input element sizes) + output
// element size (weighted_sum/add_difference 6 B, n_average N=8 18 B
// with bf16 in and out).  Every input is read exactly once with 16-byte-per-lane
// coalesced loads, the fp32 arithmetic, the finite checks of omit_non_finite and of
// check_finite_output and the output cast are fused into the same pass.
#include <type_traits>
#include "common.h"

namespace {

enum : int { OP_CAST = 0, OP_LERP, OP_NAVG, OP_ADDCMUL, OP_SUB, OP_SCALE, OP_LERP_T, OP_ADDCMUL_T, OP_ADDSUB };

struct EwParams {
    const void *in[MECHA_MAX_MODELS];
    int n_in;
    float alpha;          // LERP: w; ADDCMUL/SCALE: alpha
    float alpha_m1;       // LERP: w - 1 (fp32)
    int small;            // LERP: |w| < 0.5
    const float *alpha_t; // LERP_T / ADDCMUL_T: one fp32 weight per element (a full alpha tensor, broadcast by the caller)
    float w[MECHA_MAX_MODELS];     // NAVG: fp32(1/i) for model index i-1
    float wm1[MECHA_MAX_MODELS];   // NAVG: w - 1
};

// Compute 8 outputs from the models' 8-element vectors.  NAVG streams models in
// groups of 8 so that up to 8 independent 16-byte loads are in flight per lane.
template <int OP, int IN_DT, bool VEC>
__device__ __forceinline__ void compute8(const EwParams &p, int64_t i, int valid, float (&r)[8], bool &bad_in) {
    auto ld = [&](const void *base, float (&v)[8]) {
        if constexpr (VEC) load8<IN_DT, true>(base, i, v);          // every input element is read exactly once
        else {
#pragma unroll
            for (int j = 0; j < 8; j++) v[j] = (j < valid) ? load1<IN_DT>(base, i + j) : 0.0f;
        }
    };
    if constexpr (OP == OP_NAVG && VEC && IN_DT != MECHA_F32) {
        // 16-bit storage: up to 8 loads in flight as raw words (4 VGPRs each), converted as they are consumed
        for (int g = 0; g < p.n_in; g += 8) {
            u32x4 raw[8];
#pragma unroll
            for (int m = 0; m < 8; m++)
                if (g + m < p.n_in) raw[m] = load_raw16<true>(p.in[g + m], i);
#pragma unroll
            for (int m = 0; m < 8; m++) {
                if (g + m < p.n_in) {
                    const int idx = g + m;
                    const float w = p.w[idx], wm1 = p.wm1[idx];
                    const bool small = idx >= 2;
                    float v[8];
                    unpack16<IN_DT>(raw[m], v);
#pragma unroll
                    for (int j = 0; j < 8; j++) {
                        bad_in |= nonfinite_f32(v[j]);
                        r[j] = (idx == 0) ? v[j] : lerp_torch(r[j], v[j], w, wm1, small);
                    }
                }
            }
        }
    } else if constexpr (OP == OP_NAVG) {
        for (int g = 0; g < p.n_in; g += 8) {
            float v[8][8];
#pragma unroll
            for (int m = 0; m < 8; m++)
                if (g + m < p.n_in) ld(p.in[g + m], v[m]);
#pragma unroll
            for (int m = 0; m < 8; m++) {
                if (g + m < p.n_in) {
                    const int idx = g + m;
                    const float w = p.w[idx], wm1 = p.wm1[idx];
                    const bool small = idx >= 2;          // 1/i < 0.5 for i >= 3 (Lerp.h weight test)
#pragma unroll
                    for (int j = 0; j < 8; j++) {
                        bad_in |= nonfinite_f32(v[m][j]);
                        r[j] = (idx == 0) ? v[m][j] : lerp_torch(r[j], v[m][j], w, wm1, small);
                    }
                }
            }
        }
    } else if constexpr (OP == OP_CAST || OP == OP_SCALE) {
        float a[8];
        ld(p.in[0], a);
#pragma unroll
        for (int j = 0; j < 8; j++) {
            bad_in |= nonfinite_f32(a[j]);
            r[j] = (OP == OP_CAST) ? a[j] : a[j] * p.alpha;
        }
    } else if constexpr (OP == OP_LERP_T || OP == OP_ADDCMUL_T) {
        // a weight per element (linear.py:23 / :76 with alpha.numel() > 1): the same single fma as the scalar forms, torch's
        // vector kernels blend coefficient and base per element (cpu/LerpKernel.cpp lerp_vec; addcmul = fma(b, alpha, a))
        float a[8], b[8], w[8];
        ld(p.in[0], a);
        ld(p.in[1], b);
        if constexpr (VEC) load8<MECHA_F32, true>(p.alpha_t, i, w);
        else {
#pragma unroll
            for (int j = 0; j < 8; j++) w[j] = (j < valid) ? p.alpha_t[i + j] : 0.0f;
        }
#pragma unroll
        for (int j = 0; j < 8; j++) {
            bad_in |= nonfinite_f32(a[j]) | nonfinite_f32(b[j]);          // the weights are hyper-parameters: no omit_non_finite on them
            if constexpr (OP == OP_LERP_T) r[j] = lerp_torch(a[j], b[j], w[j], w[j] - 1.0f, __builtin_fabsf(w[j]) < 0.5f);
            else r[j] = __builtin_fmaf(b[j], w[j], a[j]);
        }
    } else if constexpr (OP == OP_ADDSUB) {
        // add_difference(a, subtract(b, c), alpha) (wrappers.py:9-39 with `c`): the delta b - c (linear.py:101) never leaves registers
        float a[8], b[8], c[8];
        ld(p.in[0], a);
        ld(p.in[1], b);
        ld(p.in[2], c);
#pragma unroll
        for (int j = 0; j < 8; j++) {
            bad_in |= nonfinite_f32(a[j]) | nonfinite_f32(b[j]) | nonfinite_f32(c[j]);
            r[j] = __builtin_fmaf(b[j] - c[j], p.alpha, a[j]);                       // a.addcmul(b - c, alpha)  (linear.py:76)
        }
    } else {
        float a[8], b[8];
        ld(p.in[0], a);
        ld(p.in[1], b);
#pragma unroll
        for (int j = 0; j < 8; j++) {
            bad_in |= nonfinite_f32(a[j]) | nonfinite_f32(b[j]);
            if constexpr (OP == OP_LERP) r[j] = lerp_torch(a[j], b[j], p.alpha, p.alpha_m1, p.small != 0);
            else if constexpr (OP == OP_ADDCMUL) r[j] = __builtin_fmaf(b[j], p.alpha, a[j]);   // a.addcmul(b, alpha)
            else r[j] = a[j] - b[j];
        }
    }
}

template <int OP, int IN_DT, int OUT_DT, bool VEC, bool HAS_OUT>
__global__ __launch_bounds__(MECHA_BLOCK) void ew_kernel(EwParams p, void *out, int64_t total,
                                                          const mecha_segment *segs, int n_segs, uint32_t *flags) {
    // FLAT grid: one 16-byte vector position per thread, one block per 256 of them, no loop.  On gfx9 loads and stores share
    // the one in-order vmcnt counter, so inside a loop every wait for the next loads also waits for the previous stores'
    // acknowledgements; a wave that stores and ends does not.  Measured on the SDXL x4 footprint (tools/stream_bench3.hip,
    // profiles/r02_stream_bench3.txt): 6.0 TB/s for 4 reads : 1 write and 6.4 TB/s for a copy, against 5.1-5.2 TB/s for every
    // looped shape (chunk per block, grid-stride, 1-4 vectors in flight, grids of 1024-16384).
    const int64_t n_vec = (total + 7) >> 3;
    const int64_t v = (int64_t)blockIdx.x * MECHA_BLOCK + threadIdx.x;
    if (v < n_vec) {
        const int64_t i = v << 3;
        const int valid = (total - i >= 8) ? 8 : (int)(total - i);
        float r[8];
        bool bad_in = false;
        if (VEC && valid == 8) compute8<OP, IN_DT, true>(p, i, 8, r, bad_in);
        else compute8<OP, IN_DT, false>(p, i, valid, r, bad_in);
        bool bad_out = false;
#pragma unroll
        for (int j = 0; j < 8; j++) bad_out |= (j < valid) && nonfinite_f32(r[j]);
        if constexpr (HAS_OUT) {
            if (VEC && valid == 8) store8<OUT_DT, true>(out, i, r);
            else {
#pragma unroll
                for (int j = 0; j < 8; j++)
                    if (j < valid) store1<OUT_DT>(out, i + j, r[j]);
            }
        }
        if ((bad_in | bad_out) && flags != nullptr) {
            // rare path.  Per key: one flag word.  Batched: the flag word of the key that owns the vector;
            // a vector whose tail lies in the gap after its key is re-evaluated on the key's elements only.
            uint32_t *word = flags;
            if (n_segs > 0) {
                const int sidx = find_segment(segs, n_segs, i);
                if (sidx < 0) return;
                word = flags + sidx;
                const int64_t seg_end = segs[sidx].offset + segs[sidx].numel;
                if (seg_end - i < valid) {
                    const int v2 = (int)(seg_end - i);
                    bad_in = false;
                    compute8<OP, IN_DT, false>(p, i, v2, r, bad_in);
                    bad_out = false;
#pragma unroll
                    for (int j = 0; j < 8; j++) bad_out |= (j < v2) && nonfinite_f32(r[j]);
                }
            }
            const uint32_t bits = (bad_in ? MECHA_FLAG_INPUT_NONFINITE : 0u) | (bad_out ? MECHA_FLAG_OUTPUT_NONFINITE : 0u);
            if (bits) atomicOr(word, bits);
        }
    }
}

template <int OP>
int launch_ew(EwParams &p, void *out, int64_t total, int in_dt, int out_dt, const mecha_segment *segs, int n_segs,
              uint32_t *flags, void *stream) {
    MECHA_REQUIRE(total >= 0, "numel must be >= 0 (got %lld)", (long long)total);
    MECHA_REQUIRE(mecha_valid_dtype(in_dt) && mecha_valid_dtype(out_dt), "unknown dtype code (in %d, out %d)", in_dt, out_dt);
    if (total == 0) return MECHA_OK;
    bool vec = (out == nullptr) || mecha_aligned16(out);
    for (int m = 0; m < p.n_in; m++) {
        MECHA_REQUIRE(p.in[m] != nullptr, "input %d is NULL", m);
        vec = vec && mecha_aligned16(p.in[m]);
    }
    if constexpr (OP == OP_LERP_T || OP == OP_ADDCMUL_T) {
        MECHA_REQUIRE(p.alpha_t != nullptr, "alpha tensor is NULL");
        vec = vec && ((reinterpret_cast<uintptr_t>(p.alpha_t) & 31u) == 0);          // two 16-byte loads per 8 weights
    }
    const int64_t n_vec = (total + 7) >> 3;
    const int64_t blocks = (n_vec + MECHA_BLOCK - 1) / MECHA_BLOCK;
    // HIP rejects a launch whose gridDim.x * blockDim.x reaches 2^32: at most 2^24 - 1 blocks of 256 threads (3.4e10 elements)
    MECHA_REQUIRE(blocks * MECHA_BLOCK < (1ll << 32), "numel %lld is beyond one launch (2^24 - 1 blocks of 2048 elements)", (long long)total);
    dim3 grid((unsigned)blocks), block(MECHA_BLOCK);
    hipStream_t s = static_cast<hipStream_t>(stream);
    MECHA_DISPATCH_DTYPES(in_dt, out_dt, {
        if (out == nullptr) {
            if constexpr (OP == OP_CAST) {                      // finite check only (omit_non_finite)
                if (vec) MECHA_LAUNCH(MK_STREAMING, (ew_kernel<OP, IN_DT, OUT_DT, true, false>), grid, block, 0, s, p, out, total, segs, n_segs, flags);
                else MECHA_LAUNCH(MK_STREAMING, (ew_kernel<OP, IN_DT, OUT_DT, false, false>), grid, block, 0, s, p, out, total, segs, n_segs, flags);
            }
        } else {
            if (vec) MECHA_LAUNCH(MK_STREAMING, (ew_kernel<OP, IN_DT, OUT_DT, true, true>), grid, block, 0, s, p, out, total, segs, n_segs, flags);
            else MECHA_LAUNCH(MK_STREAMING, (ew_kernel<OP, IN_DT, OUT_DT, false, true>), grid, block, 0, s, p, out, total, segs, n_segs, flags);
        }
    });
    MECHA_CHECK_HIP(hipGetLastError());
    return MECHA_OK;
}

void set_lerp(EwParams &p, float alpha) {
    p.alpha = alpha;
    p.alpha_m1 = alpha - 1.0f;
    p.small = (alpha < 0.5f && alpha > -0.5f) ? 1 : 0;
}

int weighted_sum_impl(const void *a, const void *b, void *out, int64_t total, int in_dt, int out_dt, float alpha,
                      const mecha_segment *segs, int n_segs, uint32_t *flags, void *stream) {
    MECHA_REQUIRE(out != nullptr || total == 0, "out is NULL");
    EwParams p{};
    if (mecha_isclose(alpha, 0.0)) { p.n_in = 1; p.in[0] = a; return launch_ew<OP_CAST>(p, out, total, in_dt, out_dt, segs, n_segs, flags, stream); }
    if (mecha_isclose(alpha, 1.0)) { p.n_in = 1; p.in[0] = b; return launch_ew<OP_CAST>(p, out, total, in_dt, out_dt, segs, n_segs, flags, stream); }
    p.n_in = 2; p.in[0] = a; p.in[1] = b;
    set_lerp(p, alpha);
    return launch_ew<OP_LERP>(p, out, total, in_dt, out_dt, segs, n_segs, flags, stream);
}

int n_average_impl(const void *const *models, int n_models, void *out, int64_t total, int in_dt, int out_dt,
                   const mecha_segment *segs, int n_segs, uint32_t *flags, void *stream) {
    MECHA_REQUIRE(out != nullptr || total == 0, "out is NULL");
    MECHA_REQUIRE(models != nullptr && n_models >= 1 && n_models <= MECHA_MAX_MODELS,
                  "n_average needs 1..%d models (got %d)", MECHA_MAX_MODELS, n_models);
    EwParams p{};
    p.n_in = n_models;
    for (int m = 0; m < n_models; m++) {
        p.in[m] = models[m];
        p.w[m] = (float)(1.0 / (double)(m + 1));      // python double 1/i narrowed to fp32 (linear.py:39)
        p.wm1[m] = p.w[m] - 1.0f;
    }
    return launch_ew<OP_NAVG>(p, out, total, in_dt, out_dt, segs, n_segs, flags, stream);
}

int add_difference_impl(const void *a, const void *b, void *out, int64_t total, int in_dt, int out_dt, float alpha,
                        const mecha_segment *segs, int n_segs, uint32_t *flags, void *stream) {
    MECHA_REQUIRE(out != nullptr || total == 0, "out is NULL");
    EwParams p{};
    if (mecha_isclose(alpha, 0.0)) { p.n_in = 1; p.in[0] = a; return launch_ew<OP_CAST>(p, out, total, in_dt, out_dt, segs, n_segs, flags, stream); }
    p.n_in = 2; p.in[0] = a; p.in[1] = b; p.alpha = alpha;
    return launch_ew<OP_ADDCMUL>(p, out, total, in_dt, out_dt, segs, n_segs, flags, stream);
}

}  // namespace

extern "C" {

int mecha_cast(const void *a, void *out, int64_t numel, int in_dtype, int out_dtype, uint32_t *flags, void *stream) {
    EwParams p{};
    p.n_in = 1; p.in[0] = a;
    return launch_ew<OP_CAST>(p, out, numel, in_dtype, out_dtype, nullptr, 0, flags, stream);
}

int mecha_weighted_sum(const void *a, const void *b, void *out, int64_t numel, int in_dtype, int out_dtype, float alpha,
                       uint32_t *flags, void *stream) {
    return weighted_sum_impl(a, b, out, numel, in_dtype, out_dtype, alpha, nullptr, 0, flags, stream);
}

int mecha_n_average(const void *const *models, int n_models, void *out, int64_t numel, int in_dtype, int out_dtype,
                    uint32_t *flags, void *stream) {
    return n_average_impl(models, n_models, out, numel, in_dtype, out_dtype, nullptr, 0, flags, stream);
}

int mecha_add_difference(const void *a, const void *b, void *out, int64_t numel, int in_dtype, int out_dtype, float alpha,
                         uint32_t *flags, void *stream) {
    return add_difference_impl(a, b, out, numel, in_dtype, out_dtype, alpha, nullptr, 0, flags, stream);
}

int mecha_weighted_sum_tensor(const void *a, const void *b, const float *alpha, void *out, int64_t numel, int in_dtype, int out_dtype,
                              uint32_t *flags, void *stream) {
    MECHA_REQUIRE(out != nullptr || numel == 0, "out is NULL");
    EwParams p{};
    p.n_in = 2; p.in[0] = a; p.in[1] = b; p.alpha_t = alpha;
    return launch_ew<OP_LERP_T>(p, out, numel, in_dtype, out_dtype, nullptr, 0, flags, stream);
}

int mecha_add_difference_tensor(const void *a, const void *b, const float *alpha, void *out, int64_t numel, int in_dtype, int out_dtype,
                                uint32_t *flags, void *stream) {
    MECHA_REQUIRE(out != nullptr || numel == 0, "out is NULL");
    EwParams p{};
    p.n_in = 2; p.in[0] = a; p.in[1] = b; p.alpha_t = alpha;
    return launch_ew<OP_ADDCMUL_T>(p, out, numel, in_dtype, out_dtype, nullptr, 0, flags, stream);
}

int mecha_subtract(const void *a, const void *b, void *out, int64_t numel, int in_dtype, int out_dtype, uint32_t *flags,
                   void *stream) {
    MECHA_REQUIRE(out != nullptr || numel == 0, "out is NULL");
    EwParams p{};
    p.n_in = 2; p.in[0] = a; p.in[1] = b;
    return launch_ew<OP_SUB>(p, out, numel, in_dtype, out_dtype, nullptr, 0, flags, stream);
}

int mecha_scale(const void *a, void *out, int64_t numel, int in_dtype, int out_dtype, float alpha, uint32_t *flags,
                void *stream) {
    MECHA_REQUIRE(out != nullptr || numel == 0, "out is NULL");
    EwParams p{};
    p.n_in = 1; p.in[0] = a; p.alpha = alpha;
    if (mecha_isclose(alpha, 1.0)) return launch_ew<OP_CAST>(p, out, numel, in_dtype, out_dtype, nullptr, 0, flags, stream);
    return launch_ew<OP_SCALE>(p, out, numel, in_dtype, out_dtype, nullptr, 0, flags, stream);
}

int mecha_batched_weighted_sum(const void *a_arena, const void *b_arena, void *out_arena, const mecha_segment *segments,
                               int n_segments, int64_t total_elems, int in_dtype, int out_dtype, float alpha,
                               uint32_t *seg_flags, void *stream) {
    MECHA_REQUIRE(n_segments > 0 && segments != nullptr, "batched call needs a device segment table");
    return weighted_sum_impl(a_arena, b_arena, out_arena, total_elems, in_dtype, out_dtype, alpha, segments, n_segments,
                             seg_flags, stream);
}

int mecha_batched_n_average(const void *const *arenas, int n_models, void *out_arena, const mecha_segment *segments,
                            int n_segments, int64_t total_elems, int in_dtype, int out_dtype, uint32_t *seg_flags,
                            void *stream) {
    MECHA_REQUIRE(n_segments > 0 && segments != nullptr, "batched call needs a device segment table");
    return n_average_impl(arenas, n_models, out_arena, total_elems, in_dtype, out_dtype, segments, n_segments, seg_flags,
                          stream);
}

int mecha_batched_add_difference(const void *a_arena, const void *b_arena, void *out_arena, const mecha_segment *segments,
                                 int n_segments, int64_t total_elems, int in_dtype, int out_dtype, float alpha,
                                 uint32_t *seg_flags, void *stream) {
    MECHA_REQUIRE(n_segments > 0 && segments != nullptr, "batched call needs a device segment table");
    return add_difference_impl(a_arena, b_arena, out_arena, total_elems, in_dtype, out_dtype, alpha, segments, n_segments,
                               seg_flags, stream);
}

int mecha_add_difference_sub(const void *a, const void *b, const void *c, void *out, int64_t numel, int in_dtype, int out_dtype,
                             float alpha, uint32_t *flags, void *stream) {
    MECHA_REQUIRE(out != nullptr || numel == 0, "out is NULL");
    MECHA_REQUIRE(!mecha_isclose(alpha, 0.0), "alpha == 0: add_difference returns `a` untouched (linear.py:72-73); use mecha_cast");
    EwParams p{};
    p.n_in = 3; p.in[0] = a; p.in[1] = b; p.in[2] = c; p.alpha = alpha;
    return launch_ew<OP_ADDSUB>(p, out, numel, in_dtype, out_dtype, nullptr, 0, flags, stream);
}

int mecha_batched_add_difference_sub(const void *a_arena, const void *b_arena, const void *c_arena, void *out_arena,
                                     const mecha_segment *segments, int n_segments, int64_t total_elems, int in_dtype,
                                     int out_dtype, float alpha, uint32_t *seg_flags, void *stream) {
    MECHA_REQUIRE(n_segments > 0 && segments != nullptr, "batched call needs a device segment table");
    MECHA_REQUIRE(out_arena != nullptr || total_elems == 0, "out is NULL");
    MECHA_REQUIRE(!mecha_isclose(alpha, 0.0), "alpha == 0: add_difference returns `a` untouched (linear.py:72-73); use the cast entry point");
    EwParams p{};
    p.n_in = 3; p.in[0] = a_arena; p.in[1] = b_arena; p.in[2] = c_arena; p.alpha = alpha;
    return launch_ew<OP_ADDSUB>(p, out_arena, total_elems, in_dtype, out_dtype, segments, n_segments, seg_flags, stream);
}

}  // extern "C"
