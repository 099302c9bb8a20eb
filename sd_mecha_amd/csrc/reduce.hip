// reduce.hip -- merges that need whole-tensor reductions: slerp, perpendicular_component, add_perpendicular.
//
// Reference: extensions/builtin/merge_methods/linear.py:43-61 (slerp), :104-113 (perpendicular_component),
// wrappers.py:42-64 (add_perpendicular = add_difference(a, perpendicular_component(a - c, b - c), alpha)).
//
// Three launches per key, two passes over the inputs (the reference makes 6-8):
//   1. reduce : per-block partial sums of a.a, b.b, a.b in fp64 -> workspace
//   2. finish : one block folds the partials and derives the scalars (norms, omega, sin weights, projection)
//   3. apply  : the elementwise formula + output cast + finite checks; raises a device flag if any result is NaN
//   4. fixup  : reads that flag; only if set rewrites the output with the reference's NaN fallback
//               (slerp: plain lerp, linear.py:59-60; perpendicular: zeros, linear.py:111-112)
// torch reduces in fp32 in an order that depends on its thread count; the sums here are fp64, rounded once.
// Roofline: HBM; algorithmic bytes = one read of each input + one write, implementation
// traffic = two reads of each input + one write.
#include <type_traits>
#include "common.h"

namespace {

enum : int { RM_SLERP = 0, RM_PERP = 1, RM_ADDPERP = 2 };

constexpr int RED_MAX_BLOCKS = 1024;

struct RedScalars {            // written by finish, read by apply / fixup
    float na, nb;              // norms
    float sa, sb, so, nl;      // slerp: sin((1-alpha) omega), sin(alpha omega), sin(omega), lerp(|a|, |b|, alpha)
    float proj;                // perpendicular: sum((a/|a|) * (b/|a|))
    uint32_t any_nan;          // set by apply
};

struct RedWorkspace {
    double partial[RED_MAX_BLOCKS][3];
    RedScalars sc;
};

struct RedParams {
    const void *a, *b, *c;
    void *out;
    int64_t n;
    float alpha, alpha_m1;
    int small;
    RedWorkspace *ws;
    uint32_t *flags;
};

template <int IN_DT, bool VEC>
__device__ __forceinline__ void ld8(const void *p, int64_t i, int valid, float (&v)[8]) {
    if constexpr (VEC) load8<IN_DT>(p, i, v);
    else {
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = (j < valid) ? load1<IN_DT>(p, i + j) : 0.0f;
    }
}

// x = first operand, y = second operand of the reduction, as fp32 (ADDPERP: a - c and b - c, linear.py:101)
template <int MODE, int IN_DT, bool VEC>
__device__ __forceinline__ void operands(const RedParams &p, int64_t i, int valid, float (&x)[8], float (&y)[8], float (&a0)[8], bool &bad_in) {
    ld8<IN_DT, VEC>(p.a, i, valid, x);
    ld8<IN_DT, VEC>(p.b, i, valid, y);
#pragma unroll
    for (int j = 0; j < 8; j++) bad_in |= nonfinite_f32(x[j]) | nonfinite_f32(y[j]);
    if constexpr (MODE == RM_ADDPERP) {
        float c[8];
        ld8<IN_DT, VEC>(p.c, i, valid, c);
#pragma unroll
        for (int j = 0; j < 8; j++) {
            bad_in |= nonfinite_f32(c[j]);
            a0[j] = x[j];
            x[j] = x[j] - c[j];
            y[j] = y[j] - c[j];
        }
    }
}

template <int MODE, int IN_DT, bool VEC>
__global__ __launch_bounds__(MECHA_BLOCK) void reduce_kernel(RedParams p) {
    double sxx = 0.0, syy = 0.0, sxy = 0.0;
    bool bad_in = false;
    const int64_t n_vec = (p.n + 7) >> 3;
    const int64_t stride = (int64_t)gridDim.x * MECHA_BLOCK;
    for (int64_t v = (int64_t)blockIdx.x * MECHA_BLOCK + threadIdx.x; v < n_vec; v += stride) {
        const int64_t i = v << 3;
        const int valid = (p.n - i >= 8) ? 8 : (int)(p.n - i);
        float x[8], y[8], a0[8];
        if (VEC && valid == 8) operands<MODE, IN_DT, true>(p, i, 8, x, y, a0, bad_in);
        else operands<MODE, IN_DT, false>(p, i, valid, x, y, a0, bad_in);
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const double dx = x[j], dy = y[j];
            sxx = __builtin_fma(dx, dx, sxx);
            syy = __builtin_fma(dy, dy, syy);
            sxy = __builtin_fma(dx, dy, sxy);
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        sxx += __shfl_down(sxx, off, 64);
        syy += __shfl_down(syy, off, 64);
        sxy += __shfl_down(sxy, off, 64);
    }
    __shared__ double wsum[MECHA_BLOCK / 64][3];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { wsum[wave][0] = sxx; wsum[wave][1] = syy; wsum[wave][2] = sxy; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double t0 = 0, t1 = 0, t2 = 0;
        for (int w = 0; w < MECHA_BLOCK / 64; w++) { t0 += wsum[w][0]; t1 += wsum[w][1]; t2 += wsum[w][2]; }
        p.ws->partial[blockIdx.x][0] = t0;
        p.ws->partial[blockIdx.x][1] = t1;
        p.ws->partial[blockIdx.x][2] = t2;
    }
    if (bad_in && p.flags != nullptr) atomicOr(p.flags, MECHA_FLAG_INPUT_NONFINITE);
}

template <int MODE>
__global__ __launch_bounds__(MECHA_BLOCK) void finish_kernel(RedParams p, int n_blocks) {
    double s0 = 0, s1 = 0, s2 = 0;
    for (int i = threadIdx.x; i < n_blocks; i += MECHA_BLOCK) { s0 += p.ws->partial[i][0]; s1 += p.ws->partial[i][1]; s2 += p.ws->partial[i][2]; }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { s0 += __shfl_down(s0, off, 64); s1 += __shfl_down(s1, off, 64); s2 += __shfl_down(s2, off, 64); }
    __shared__ double wsum[MECHA_BLOCK / 64][3];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { wsum[wave][0] = s0; wsum[wave][1] = s1; wsum[wave][2] = s2; }
    __syncthreads();
    if (threadIdx.x != 0) return;
    double sxx = 0, syy = 0, sxy = 0;
    for (int w = 0; w < MECHA_BLOCK / 64; w++) { sxx += wsum[w][0]; syy += wsum[w][1]; sxy += wsum[w][2]; }
    RedScalars sc{};
    sc.na = (float)__builtin_sqrt(sxx);                                     // a.norm()
    sc.nb = (float)__builtin_sqrt(syy);
    if constexpr (MODE == RM_SLERP) {
        float ab = (float)(sxy / ((double)sc.na * (double)sc.nb));            // (a_n * b_n).sum()
        ab = (ab < -1.0f) ? -1.0f : ((ab > 1.0f) ? 1.0f : ab);                // .clamp(-1, 1); NaN passes through
        const float omega = acosf(ab);
        sc.sa = sinf((1.0f - p.alpha) * omega);
        sc.sb = sinf(p.alpha * omega);
        sc.so = sinf(omega);
        sc.nl = lerp_torch(sc.na, sc.nb, p.alpha, p.alpha_m1, p.small != 0);  // torch.lerp(a.norm(), b.norm(), alpha)
    } else {
        sc.proj = (float)(sxy / ((double)sc.na * (double)sc.na));             // sum((a/|a|) * (b/|a|))
    }
    sc.any_nan = 0;
    p.ws->sc = sc;
}

template <int MODE, int IN_DT, int OUT_DT, bool VEC>
__global__ __launch_bounds__(MECHA_BLOCK) void apply_kernel(RedParams p) {
    const RedScalars sc = p.ws->sc;
    bool any_nan = false, bad_out = false;
    const int64_t n_vec = (p.n + 7) >> 3;
    const int64_t stride = (int64_t)gridDim.x * MECHA_BLOCK;
    for (int64_t v = (int64_t)blockIdx.x * MECHA_BLOCK + threadIdx.x; v < n_vec; v += stride) {
        const int64_t i = v << 3;
        const int valid = (p.n - i >= 8) ? 8 : (int)(p.n - i);
        const bool vec = VEC && valid == 8;
        float x[8], y[8], a0[8], r[8];
        bool bad_in = false;
        if (vec) operands<MODE, IN_DT, true>(p, i, 8, x, y, a0, bad_in);
        else operands<MODE, IN_DT, false>(p, i, valid, x, y, a0, bad_in);
#pragma unroll
        for (int j = 0; j < 8; j++) {
            float q;
            if constexpr (MODE == RM_SLERP) {
                const float an = x[j] / sc.na, bn = y[j] / sc.nb;              // linear.py:49-50
                q = (an * sc.sa + bn * sc.sb) / sc.so;                         // linear.py:55-57
                q = q * sc.nl;                                                 // linear.py:58
            } else {
                q = y[j] - x[j] * sc.proj;                                     // linear.py:110
            }
            if (j < valid) any_nan |= (q != q);
            if constexpr (MODE == RM_ADDPERP) q = __builtin_fmaf(q, p.alpha, a0[j]);   // add_difference(a, perp, alpha)
            r[j] = q;
            bad_out |= (j < valid) && nonfinite_f32(q);
        }
        if (vec) store8<OUT_DT>(p.out, i, r);
        else {
#pragma unroll
            for (int j = 0; j < 8; j++)
                if (j < valid) store1<OUT_DT>(p.out, i + j, r[j]);
        }
    }
    if (any_nan) atomicOr(&p.ws->sc.any_nan, 1u);
    if (bad_out && p.flags != nullptr) atomicOr(p.flags, MECHA_FLAG_OUTPUT_NONFINITE);
}

// Runs always, does work only if apply saw a NaN (the reference's `if res.isnan().any()` branches).
template <int MODE, int IN_DT, int OUT_DT, bool VEC>
__global__ __launch_bounds__(MECHA_BLOCK) void fixup_kernel(RedParams p) {
    if (p.ws->sc.any_nan == 0) return;
    const int64_t n_vec = (p.n + 7) >> 3;
    const int64_t stride = (int64_t)gridDim.x * MECHA_BLOCK;
    bool bad_out = false;
    for (int64_t v = (int64_t)blockIdx.x * MECHA_BLOCK + threadIdx.x; v < n_vec; v += stride) {
        const int64_t i = v << 3;
        const int valid = (p.n - i >= 8) ? 8 : (int)(p.n - i);
        const bool vec = VEC && valid == 8;
        float r[8];
        if constexpr (MODE == RM_SLERP) {
            float a[8], b[8];
            if (vec) { ld8<IN_DT, true>(p.a, i, 8, a); ld8<IN_DT, true>(p.b, i, 8, b); }
            else { ld8<IN_DT, false>(p.a, i, valid, a); ld8<IN_DT, false>(p.b, i, valid, b); }
#pragma unroll
            for (int j = 0; j < 8; j++) r[j] = lerp_torch(a[j], b[j], p.alpha, p.alpha_m1, p.small != 0);   // linear.py:60
        } else if constexpr (MODE == RM_PERP) {
#pragma unroll
            for (int j = 0; j < 8; j++) r[j] = 0.0f;                                                      // linear.py:112
        } else {
            float a[8];
            if (vec) ld8<IN_DT, true>(p.a, i, 8, a); else ld8<IN_DT, false>(p.a, i, valid, a);
#pragma unroll
            for (int j = 0; j < 8; j++) r[j] = __builtin_fmaf(0.0f, p.alpha, a[j]);                       // a.addcmul(zeros, alpha)
        }
#pragma unroll
        for (int j = 0; j < 8; j++) bad_out |= (j < valid) && nonfinite_f32(r[j]);
        if (vec) store8<OUT_DT>(p.out, i, r);
        else {
#pragma unroll
            for (int j = 0; j < 8; j++)
                if (j < valid) store1<OUT_DT>(p.out, i + j, r[j]);
        }
    }
    // the output flag raised by apply described the discarded result; the fallback output is what counts,
    // but a flag word can only be OR-ed into: a NaN fallback leaves the (warning-only) output bit set.
    if (bad_out && p.flags != nullptr) atomicOr(p.flags, MECHA_FLAG_OUTPUT_NONFINITE);
}

template <int MODE>
int run_reduce(const void *a, const void *b, const void *c, void *out, int64_t n, int in_dt, int out_dt, float alpha,
               void *workspace, size_t workspace_bytes, uint32_t *flags, void *stream) {
    MECHA_REQUIRE(n >= 0, "numel must be >= 0");
    MECHA_REQUIRE(n == 0 || (a != nullptr && b != nullptr && out != nullptr && (MODE != RM_ADDPERP || c != nullptr)), "NULL tensor pointer");
    MECHA_REQUIRE(mecha_valid_dtype(in_dt) && mecha_valid_dtype(out_dt), "unknown dtype code (in %d, out %d)", in_dt, out_dt);
    MECHA_REQUIRE(workspace != nullptr && workspace_bytes >= sizeof(RedWorkspace) && mecha_aligned16(workspace),
                  "reduce workspace too small or unaligned: need %zu bytes, got %zu", sizeof(RedWorkspace), workspace_bytes);
    if (n == 0) return MECHA_OK;
    const bool vec = mecha_aligned16(a) && mecha_aligned16(b) && mecha_aligned16(out) && (c == nullptr || mecha_aligned16(c));
    RedParams p{};
    p.a = a; p.b = b; p.c = c; p.out = out; p.n = n; p.alpha = alpha; p.alpha_m1 = alpha - 1.0f;
    p.small = (alpha < 0.5f && alpha > -0.5f) ? 1 : 0;
    p.ws = static_cast<RedWorkspace *>(workspace);
    p.flags = flags;
    const int64_t n_vec = (n + 7) >> 3;
    int64_t blocks = (n_vec + MECHA_BLOCK - 1) / MECHA_BLOCK;
    const int red_blocks = (int)(blocks < RED_MAX_BLOCKS ? blocks : RED_MAX_BLOCKS);
    const unsigned ew_blocks = (unsigned)(blocks < MECHA_MAX_GRID ? blocks : MECHA_MAX_GRID);
    hipStream_t s = static_cast<hipStream_t>(stream);
    MECHA_DISPATCH_DTYPES(in_dt, out_dt, {
        if (vec) {
            MECHA_LAUNCH(MK_RED_REDUCE, (reduce_kernel<MODE, IN_DT, true>), dim3(red_blocks), dim3(MECHA_BLOCK), 0, s, p);
            MECHA_LAUNCH(MK_RED_FINISH, (finish_kernel<MODE>), dim3(1), dim3(MECHA_BLOCK), 0, s, p, red_blocks);
            MECHA_LAUNCH(MK_RED_APPLY, (apply_kernel<MODE, IN_DT, OUT_DT, true>), dim3(ew_blocks), dim3(MECHA_BLOCK), 0, s, p);
            MECHA_LAUNCH(MK_RED_FIXUP, (fixup_kernel<MODE, IN_DT, OUT_DT, true>), dim3(ew_blocks), dim3(MECHA_BLOCK), 0, s, p);
        } else {
            MECHA_LAUNCH(MK_RED_REDUCE, (reduce_kernel<MODE, IN_DT, false>), dim3(red_blocks), dim3(MECHA_BLOCK), 0, s, p);
            MECHA_LAUNCH(MK_RED_FINISH, (finish_kernel<MODE>), dim3(1), dim3(MECHA_BLOCK), 0, s, p, red_blocks);
            MECHA_LAUNCH(MK_RED_APPLY, (apply_kernel<MODE, IN_DT, OUT_DT, false>), dim3(ew_blocks), dim3(MECHA_BLOCK), 0, s, p);
            MECHA_LAUNCH(MK_RED_FIXUP, (fixup_kernel<MODE, IN_DT, OUT_DT, false>), dim3(ew_blocks), dim3(MECHA_BLOCK), 0, s, p);
        }
    });
    MECHA_CHECK_HIP(hipGetLastError());
    return MECHA_OK;
}

}  // namespace

extern "C" {

size_t mecha_reduce_workspace_bytes(int64_t numel) {
    (void)numel;
    return sizeof(RedWorkspace);
}

int mecha_slerp(const void *a, const void *b, void *out, int64_t numel, int in_dtype, int out_dtype, float alpha,
                void *workspace, size_t workspace_bytes, uint32_t *flags, void *stream) {
    return run_reduce<RM_SLERP>(a, b, nullptr, out, numel, in_dtype, out_dtype, alpha, workspace, workspace_bytes, flags, stream);
}

int mecha_perpendicular_component(const void *a, const void *b, void *out, int64_t numel, int in_dtype, int out_dtype,
                                  void *workspace, size_t workspace_bytes, uint32_t *flags, void *stream) {
    return run_reduce<RM_PERP>(a, b, nullptr, out, numel, in_dtype, out_dtype, 0.0f, workspace, workspace_bytes, flags, stream);
}

int mecha_add_perpendicular(const void *a, const void *b, const void *c, void *out, int64_t numel, int in_dtype,
                            int out_dtype, float alpha, void *workspace, size_t workspace_bytes, uint32_t *flags, void *stream) {
    return run_reduce<RM_ADDPERP>(a, b, c, out, numel, in_dtype, out_dtype, alpha, workspace, workspace_bytes, flags, stream);
}

}  // extern "C"
