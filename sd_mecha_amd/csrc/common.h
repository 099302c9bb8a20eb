// common.h -- shared device helpers for libmecha_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "../../include/mecha_hip.h"

// ---------------------------------------------------------------- errors
void mecha_set_error(const char *fmt, ...);   // abi.cpp; thread-local message

#define MECHA_CHECK_HIP(expr)                                                          \
    do {                                                                               \
        hipError_t _e = (expr);                                                        \
        if (_e != hipSuccess) {                                                        \
            mecha_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return MECHA_ERR_HIP;                                                      \
        }                                                                              \
    } while (0)

#define MECHA_REQUIRE(cond, ...)                                                       \
    do {                                                                               \
        if (!(cond)) {                                                                 \
            mecha_set_error(__VA_ARGS__);                                              \
            return MECHA_ERR_ARG;                                                      \
        }                                                                              \
    } while (0)

static inline bool mecha_valid_dtype(int dt) { return dt == MECHA_F32 || dt == MECHA_F16 || dt == MECHA_BF16; }
static inline size_t mecha_dtype_size(int dt) { return dt == MECHA_F32 ? 4 : 2; }
// python's math.isclose(a, b) with the defaults the reference relies on: rel_tol = 1e-9, abs_tol = 0
// (linear.py:18-21, :72, ties.py:26).
static inline bool mecha_isclose(double a, double b) {
    if (a == b) return true;
    const double diff = a > b ? a - b : b - a, aa = a < 0 ? -a : a, ab = b < 0 ? -b : b;
    return diff <= 1e-9 * (aa > ab ? aa : ab);
}
static inline bool mecha_aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// ---------------------------------------------------------------- per-kernel timers
// Off by default.  When enabled (mecha_profile_enable) every launch is bracketed by HIP events on its own
// stream; mecha_profile_read sums the elapsed times per kernel family.  bench.py uses this for `roofline`.
enum MechaKernelId : int {
    MK_STREAMING = 0, MK_TIES_INDEX, MK_TIES_HIST1, MK_TIES_HIST2, MK_TIES_HIST3, MK_TIES_SCAN, MK_TIES_MERGE,
    MK_RED_REDUCE, MK_RED_FINISH, MK_RED_APPLY, MK_RED_FIXUP, MK_LORA, MK_ROT_GEMM, MK_ROT_NORMS,
    MK_TIES_SAMPLE, MK_TIES_COUNT, MK_TIES_RESOLVE, MK_TIES_FIXUP, MK_COUNT
};
extern bool g_mecha_profiling;
void mecha_prof_begin(hipStream_t s, int id);
void mecha_prof_end(hipStream_t s, int id);
#define MECHA_LAUNCH(id, kernel, grid, block, lds, stream, ...)                        \
    do {                                                                               \
        if (g_mecha_profiling) mecha_prof_begin(stream, id);                           \
        hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__);             \
        if (g_mecha_profiling) mecha_prof_end(stream, id);                             \
    } while (0)

// ---------------------------------------------------------------- geometry
// 256 CUs; memory-bound grids are capped at 8 blocks of 256 threads per CU and
// stride over the rest (cdna_hip_programming.md Guideline 11).
constexpr int MECHA_BLOCK = 256;
constexpr int MECHA_MAX_GRID = 256 * 8;
constexpr int MECHA_VEC = 8;                       // elements per lane per vector access (16 B of bf16/fp16)

// ---------------------------------------------------------------- conversions
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float bf16_bits_to_f32(uint32_t h) { return __uint_as_float(h << 16); }
__device__ __forceinline__ float f16_bits_to_f32(uint32_t h) {
    _Float16 x = __builtin_bit_cast(_Float16, (uint16_t)h);
    return (float)x;                                // v_cvt_f32_f16, exact
}
__device__ __forceinline__ uint32_t f32_to_bf16_bits(float f) {
    __bf16 h = (__bf16)f;                           // v_cvt_pk_bf16_f32: round-to-nearest-even, NaN stays NaN
    return (uint32_t)__builtin_bit_cast(uint16_t, h);
}
__device__ __forceinline__ uint32_t f32_to_f16_bits(float f) {
    _Float16 h = (_Float16)f;                       // v_cvt_f16_f32, round-to-nearest-even
    return (uint32_t)__builtin_bit_cast(uint16_t, h);
}
__device__ __forceinline__ bool nonfinite_f32(float f) { return (__float_as_uint(f) & 0x7f800000u) == 0x7f800000u; }

// Load 8 consecutive elements starting at element index i (i % 8 == 0, base 16-B aligned) as fp32.
// NT = non-temporal: for data a kernel touches exactly once (measured on the streaming merge shape, tools/stream_bench.hip:
// contiguous chunk per block + 2 vectors in flight + nt = 5.1-5.4 TB/s against 4.0-4.7 TB/s for a grid-stride loop).
template <int DT, bool NT = false>
__device__ __forceinline__ void load8(const void *base, int64_t i, float (&v)[8]) {
    if constexpr (DT == MECHA_F32) {
        const f32x4 *p = reinterpret_cast<const f32x4 *>(reinterpret_cast<const float *>(base) + i);
        f32x4 a, b;
        if constexpr (NT) { a = __builtin_nontemporal_load(p); b = __builtin_nontemporal_load(p + 1); }
        else { a = p[0]; b = p[1]; }
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else {
        const u32x4 *q = reinterpret_cast<const u32x4 *>(reinterpret_cast<const uint16_t *>(base) + i);
        u32x4 r;
        if constexpr (NT) r = __builtin_nontemporal_load(q);
        else r = *q;
        uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if constexpr (DT == MECHA_BF16) {
                v[2 * j] = __uint_as_float(w[j] << 16);
                v[2 * j + 1] = __uint_as_float(w[j] & 0xffff0000u);
            } else {
                v[2 * j] = f16_bits_to_f32(w[j] & 0xffffu);
                v[2 * j + 1] = f16_bits_to_f32(w[j] >> 16);
            }
        }
    }
}
// The same for 16-bit storage in two steps, so that a kernel can keep many loads in flight as raw words
// (4 VGPRs per vector instead of 8) and convert when it consumes them.
template <bool NT>
__device__ __forceinline__ u32x4 load_raw16(const void *base, int64_t i) {
    const u32x4 *q = reinterpret_cast<const u32x4 *>(reinterpret_cast<const uint16_t *>(base) + i);
    if constexpr (NT) return __builtin_nontemporal_load(q);
    else return *q;
}
template <int DT>
__device__ __forceinline__ void unpack16(u32x4 r, float (&v)[8]) {
    static_assert(DT != MECHA_F32, "16-bit storage only");
    uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int j = 0; j < 4; j++) {
        if constexpr (DT == MECHA_BF16) {
            v[2 * j] = __uint_as_float(w[j] << 16);
            v[2 * j + 1] = __uint_as_float(w[j] & 0xffff0000u);
        } else {
            v[2 * j] = f16_bits_to_f32(w[j] & 0xffffu);
            v[2 * j + 1] = f16_bits_to_f32(w[j] >> 16);
        }
    }
}
// Scalar element load (tails and unaligned tensors).
template <int DT>
__device__ __forceinline__ float load1(const void *base, int64_t i) {
    if constexpr (DT == MECHA_F32) return reinterpret_cast<const float *>(base)[i];
    else if constexpr (DT == MECHA_BF16) return bf16_bits_to_f32(reinterpret_cast<const uint16_t *>(base)[i]);
    else return f16_bits_to_f32(reinterpret_cast<const uint16_t *>(base)[i]);
}
template <int DT, bool NT = false>
__device__ __forceinline__ void store8(void *base, int64_t i, const float (&v)[8]) {
    if constexpr (DT == MECHA_F32) {
        f32x4 *p = reinterpret_cast<f32x4 *>(reinterpret_cast<float *>(base) + i);
        f32x4 a = {v[0], v[1], v[2], v[3]}, b = {v[4], v[5], v[6], v[7]};
        if constexpr (NT) { __builtin_nontemporal_store(a, p); __builtin_nontemporal_store(b, p + 1); }
        else { p[0] = a; p[1] = b; }
    } else {
        uint32_t w[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if constexpr (DT == MECHA_BF16) w[j] = f32_to_bf16_bits(v[2 * j]) | (f32_to_bf16_bits(v[2 * j + 1]) << 16);
            else w[j] = f32_to_f16_bits(v[2 * j]) | (f32_to_f16_bits(v[2 * j + 1]) << 16);
        }
        u32x4 r = {w[0], w[1], w[2], w[3]};
        u32x4 *q = reinterpret_cast<u32x4 *>(reinterpret_cast<uint16_t *>(base) + i);
        if constexpr (NT) __builtin_nontemporal_store(r, q);
        else *q = r;
    }
}
template <int DT>
__device__ __forceinline__ void store1(void *base, int64_t i, float v) {
    if constexpr (DT == MECHA_F32) reinterpret_cast<float *>(base)[i] = v;
    else if constexpr (DT == MECHA_BF16) reinterpret_cast<uint16_t *>(base)[i] = (uint16_t)f32_to_bf16_bits(v);
    else reinterpret_cast<uint16_t *>(base)[i] = (uint16_t)f32_to_f16_bits(v);
}

// torch.lerp with a scalar weight as torch's CPU vector kernel evaluates it
// (ATen/native/Lerp.h:12-34, cpu/LerpKernel.cpp lerp_vec): ONE fma.
__device__ __forceinline__ float lerp_torch(float a, float b, float w, float wm1, bool small) {
    float d = b - a;
    return small ? __builtin_fmaf(w, d, a) : __builtin_fmaf(wm1, d, b);
}

// Dispatch a (in_dtype, out_dtype) pair to template arguments.
#define MECHA_DISPATCH_DTYPES(in_dt, out_dt, ...)                                        \
    do {                                                                                 \
        auto _body = [&](auto _in, auto _out) {                                          \
            constexpr int IN_DT = decltype(_in)::value;                                  \
            constexpr int OUT_DT = decltype(_out)::value;                                \
            __VA_ARGS__;                                                                 \
        };                                                                               \
        auto _outer = [&](auto _in) {                                                    \
            switch (out_dt) {                                                            \
                case MECHA_F32: _body(_in, std::integral_constant<int, MECHA_F32>{}); break;  \
                case MECHA_F16: _body(_in, std::integral_constant<int, MECHA_F16>{}); break;  \
                default: _body(_in, std::integral_constant<int, MECHA_BF16>{}); break;   \
            }                                                                            \
        };                                                                               \
        switch (in_dt) {                                                                 \
            case MECHA_F32: _outer(std::integral_constant<int, MECHA_F32>{}); break;     \
            case MECHA_F16: _outer(std::integral_constant<int, MECHA_F16>{}); break;     \
            default: _outer(std::integral_constant<int, MECHA_BF16>{}); break;           \
        }                                                                                \
    } while (0)

// Batched mode: index of the segment that contains element `elem`, or -1 if it lies in a gap
// (binary search on the sorted segment offsets).  Called on the rare non-finite path only.
__device__ __forceinline__ int find_segment(const mecha_segment *segs, int n_segs, int64_t elem) {
    int lo = 0, hi = n_segs - 1;
    while (lo < hi) {                              // last segment with offset <= elem
        int mid = (lo + hi + 1) >> 1;
        if (segs[mid].offset <= elem) lo = mid; else hi = mid - 1;
    }
    return (elem >= segs[lo].offset && elem < segs[lo].offset + segs[lo].numel) ? lo : -1;
}
