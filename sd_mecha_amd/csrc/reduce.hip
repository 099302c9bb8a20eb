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
    uint32_t bad_out;          // batched form: apply saw a non-finite result (reported only if the key is not rewritten by fixup)
    float rna, rnb, rso;       // batched form: RN(1 / na), RN(1 / nb), RN(1 / so) for the three-instruction division
    uint32_t fast_div;         // batched form: 1 = that division is safe for this key's divisors
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


// ------------------------------------------------------------------------------------------------ batched (segmented) form
// Every key of an arena in ONE launch sequence (replaces one future per key, merging.py:223-229, for recipes
// slerp(model, model) and add_perpendicular-shaped graphs).  The reductions are per KEY (linear.py:49-52, :109-110 reduce over
// the whole tensor), so the arena is cut into tiles of BRED_TILE elements that never straddle a key:
//   1. tiles   : tile_start[] (prefix of tiles per key) and one 16-byte record per tile, built on the device from the
//                segment table
//   2. reduce  : FLAT grid, one block per tile, one 16-byte vector position per thread: fp64 partial sums of x.x, y.y, x.y
//                per tile -> workspace (no atomics: a key's partials are summed in tile order, so results do not depend on
//                the order blocks run in)
//   3. finish  : one wave per key folds its tiles' partials and derives the key's scalars
//   4. apply   : flat grid again; a block reads its key's scalars, applies the formula, casts, checks; a key whose result
//                holds a NaN is noted in its scalars and in one global word
//   5. fixup   : a small grid that exits at once unless that word is set; then it rewrites the keys that were noted with the
//                reference's NaN fallback (linear.py:59-60, :111-112)
// Roofline: HBM; algorithmic bytes = one read of each input + one write, implementation = two reads + one write.
constexpr int BRED_TILE = MECHA_BLOCK * 8;           // 2048 elements: one vector position per thread

struct BredTile { long long elem; int seg; int valid; };   // first element inside the arena, key, elements in the tile

struct BredGlobals { uint32_t n_nan_keys; uint32_t pad[3]; };

struct BredParams {
    const void *a, *b, *c;
    void *out;
    const mecha_segment *segs;
    int n_segs;
    const long long *tile_start;        // [n_segs + 1]
    const BredTile *tiles;
    long long n_tiles;
    double *partial;                    // [n_tiles][3]
    RedScalars *sc;                     // [n_segs]
    BredGlobals *globals;
    int *nan_keys;                      // [n_segs] keys whose result held a NaN, in the order the apply pass met them
    float alpha, alpha_m1;
    int small;
    int vec_ok;
    uint32_t *flags;                    // [n_segs] or nullptr
};

__global__ __launch_bounds__(1024) void bred_tile_index_kernel(const mecha_segment *segs, int n_segs, long long *tile_start) {
    __shared__ long long part[1024];
    const int per = (n_segs + 1023) / 1024;
    const int lo = (threadIdx.x * per < n_segs) ? threadIdx.x * per : n_segs;
    const int hi = (lo + per < n_segs) ? lo + per : n_segs;
    long long sum = 0;
    for (int i = lo; i < hi; i++) sum += (segs[i].numel + BRED_TILE - 1) / BRED_TILE;
    part[threadIdx.x] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        long long run = 0;
        for (int i = 0; i < 1024; i++) { long long v = part[i]; part[i] = run; run += v; }
        tile_start[n_segs] = run;
    }
    __syncthreads();
    long long run = part[threadIdx.x];
    for (int i = lo; i < hi; i++) { tile_start[i] = run; run += (segs[i].numel + BRED_TILE - 1) / BRED_TILE; }
}

__global__ __launch_bounds__(MECHA_BLOCK) void bred_tile_records_kernel(const mecha_segment *segs, const long long *tile_start, BredTile *rec) {
    const int s = blockIdx.x;
    const mecha_segment seg = segs[s];
    const long long first = tile_start[s], n = tile_start[s + 1] - first;
    for (long long i = threadIdx.x; i < n; i += MECHA_BLOCK) {
        const long long left = seg.numel - i * BRED_TILE;
        rec[first + i] = BredTile{seg.offset + i * BRED_TILE, s, (int)(left < BRED_TILE ? left : BRED_TILE)};
    }
}

// operands of one vector position of a tile: x, y (ADDPERP: a - c, b - c) and a itself
template <int MODE, int IN_DT>
__device__ __forceinline__ void bred_operands(const BredParams &p, int64_t e, int valid, bool vec, float (&x)[8], float (&y)[8], float (&a0)[8], bool &bad_in) {
    float c[8];
    if (vec) {
        load8<IN_DT, true>(p.a, e, x);
        load8<IN_DT, true>(p.b, e, y);
        if constexpr (MODE == RM_ADDPERP) load8<IN_DT, true>(p.c, e, c);
    } else {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            x[j] = (j < valid) ? load1<IN_DT>(p.a, e + j) : 0.0f;
            y[j] = (j < valid) ? load1<IN_DT>(p.b, e + j) : 0.0f;
            if constexpr (MODE == RM_ADDPERP) c[j] = (j < valid) ? load1<IN_DT>(p.c, e + j) : 0.0f;
        }
    }
    // fma(v, 0, poison) stays 0 while every input is finite and turns NaN after an inf or a NaN (two values per v_pk_fma_f32)
    f32x2 poison = {0.0f, 0.0f};
    const f32x2 zero2 = {0.0f, 0.0f};
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
        poison = __builtin_elementwise_fma(f32x2{x[j], x[j + 1]}, zero2, poison);
        poison = __builtin_elementwise_fma(f32x2{y[j], y[j + 1]}, zero2, poison);
        if constexpr (MODE == RM_ADDPERP) poison = __builtin_elementwise_fma(f32x2{c[j], c[j + 1]}, zero2, poison);
    }
    bad_in |= (poison.x != poison.x) | (poison.y != poison.y);
    if constexpr (MODE == RM_ADDPERP) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            a0[j] = x[j];
            x[j] = x[j] - c[j];                       // subtract (linear.py:101)
            y[j] = y[j] - c[j];
        }
    }
}

// The reduce pass as a walk: a block owns a contiguous run of `chunk` tiles and every lane keeps fp64 partial sums for the key it
// is in; they are folded across the wave and written ONCE per (block, key) -- into the slots of the key's first tile inside the
// block's run -- instead of once per tile.  (A fold per tile costs 36 ds_bpermute per wave for three doubles: the flat
// one-block-per-tile form ran at one tile per 1.7 ns whatever it read -- 3.0 ms for 13.9 GB as for 20.8 GB -- bound by the LDS
// crossbar, not by HBM.)  The loads of tile t + 1 are issued before tile t is consumed (two register slots of raw words,
// converted when consumed).  Which slots hold a partial follows from `chunk` alone, so the finish pass finds them without a
// list, and the sums do not depend on the order the blocks run in.
constexpr int BRED_WAVES = MECHA_BLOCK / 64;      // partial slots per tile (one per wave)

template <int IN_DT> struct BredRaw { u32x4 w[IN_DT == MECHA_F32 ? 2 : 1]; };
template <int IN_DT>
__device__ __forceinline__ BredRaw<IN_DT> bred_load_raw(const void *ptr, int64_t e, int valid, bool vec) {
    BredRaw<IN_DT> r;
    if (vec) {
        if constexpr (IN_DT == MECHA_F32) {
            const u32x4 *q = reinterpret_cast<const u32x4 *>(reinterpret_cast<const float *>(ptr) + e);
            r.w[0] = __builtin_nontemporal_load(q);
            r.w[1] = __builtin_nontemporal_load(q + 1);
        } else {
            r.w[0] = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(reinterpret_cast<const uint16_t *>(ptr) + e));
        }
    } else {
        uint32_t t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < 8; j++) {
            if (j < valid) {
                if constexpr (IN_DT == MECHA_F32) t[j] = reinterpret_cast<const uint32_t *>(ptr)[e + j];
                else t[j >> 1] |= (uint32_t)reinterpret_cast<const uint16_t *>(ptr)[e + j] << (16 * (j & 1));
            }
        }
        r.w[0] = u32x4{t[0], t[1], t[2], t[3]};
        if constexpr (IN_DT == MECHA_F32) r.w[1] = u32x4{t[4], t[5], t[6], t[7]};
    }
    return r;
}
template <int IN_DT>
__device__ __forceinline__ void bred_unpack(const BredRaw<IN_DT> &r, float (&v)[8]) {
    if constexpr (IN_DT == MECHA_F32) {
        v[0] = __uint_as_float(r.w[0].x); v[1] = __uint_as_float(r.w[0].y); v[2] = __uint_as_float(r.w[0].z); v[3] = __uint_as_float(r.w[0].w);
        v[4] = __uint_as_float(r.w[1].x); v[5] = __uint_as_float(r.w[1].y); v[6] = __uint_as_float(r.w[1].z); v[7] = __uint_as_float(r.w[1].w);
    } else {
        unpack16<IN_DT>(r.w[0], v);
    }
}

template <int MODE, int IN_DT>
__global__ __launch_bounds__(MECHA_BLOCK) void bred_reduce_kernel(BredParams p, long long chunk) {
    const long long t0 = (long long)blockIdx.x * chunk;
    const long long t1 = (t0 + chunk < p.n_tiles) ? t0 + chunk : p.n_tiles;
    if (t0 >= t1) return;
    const int local = threadIdx.x * 8;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int NIN = MODE == RM_ADDPERP ? 3 : 2;
    BredTile rec_u[2];
    int valid_u[2];
    BredRaw<IN_DT> raw_u[2][NIN];
    auto issue = [&](long long t, auto slot_c) {
        constexpr int slot = decltype(slot_c)::value;
        const BredTile rec = p.tiles[t];
        const int valid = rec.valid - local >= 8 ? 8 : (rec.valid - local > 0 ? rec.valid - local : 0);
        rec_u[slot] = rec;
        valid_u[slot] = valid;
        if (valid == 0) return;
        const bool vec = p.vec_ok && valid == 8;
        const int64_t e = rec.elem + local;
        raw_u[slot][0] = bred_load_raw<IN_DT>(p.a, e, valid, vec);
        raw_u[slot][1] = bred_load_raw<IN_DT>(p.b, e, valid, vec);
        if constexpr (MODE == RM_ADDPERP) raw_u[slot][2] = bred_load_raw<IN_DT>(p.c, e, valid, vec);
    };
    double axx = 0.0, ayy = 0.0, axy = 0.0;          // this lane's sums over the current key's tiles of this block
    f32x2 poison = {0.0f, 0.0f};                     // fma(v, 0, poison): 0 while every input is finite, NaN after an inf / NaN
    int cur_seg = -1;
    long long run_first = t0;                        // the tile whose slots receive the current key's partials
    auto flush = [&]() {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            axx += __shfl_down(axx, off, 64);
            ayy += __shfl_down(ayy, off, 64);
            axy += __shfl_down(axy, off, 64);
        }
        if (lane == 0) {
            double *dst = p.partial + (run_first * BRED_WAVES + wave) * 3;
            dst[0] = axx; dst[1] = ayy; dst[2] = axy;
        }
        if (((poison.x != poison.x) | (poison.y != poison.y)) && p.flags != nullptr) atomicOr(p.flags + cur_seg, MECHA_FLAG_INPUT_NONFINITE);
        axx = ayy = axy = 0.0;
        poison = f32x2{0.0f, 0.0f};
    };
    auto consume = [&](long long t, auto slot_c) {
        constexpr int slot = decltype(slot_c)::value;
        const int seg = rec_u[slot].seg;             // block-uniform
        if (seg != cur_seg) {
            if (cur_seg >= 0) flush();
            cur_seg = seg;
            run_first = t;
        }
        if (valid_u[slot] == 0) return;
        float x[8], y[8];
        bred_unpack<IN_DT>(raw_u[slot][0], x);
        bred_unpack<IN_DT>(raw_u[slot][1], y);
        const f32x2 zero2 = {0.0f, 0.0f};
#pragma unroll
        for (int j = 0; j < 8; j += 2) {
            poison = __builtin_elementwise_fma(f32x2{x[j], x[j + 1]}, zero2, poison);
            poison = __builtin_elementwise_fma(f32x2{y[j], y[j + 1]}, zero2, poison);
        }
        if constexpr (MODE == RM_ADDPERP) {
            float c[8];
            bred_unpack<IN_DT>(raw_u[slot][2], c);
#pragma unroll
            for (int j = 0; j < 8; j += 2) poison = __builtin_elementwise_fma(f32x2{c[j], c[j + 1]}, zero2, poison);
#pragma unroll
            for (int j = 0; j < 8; j++) { x[j] = x[j] - c[j]; y[j] = y[j] - c[j]; }          // subtract (linear.py:101)
        }
        // a lane's 8 elements in fp32, in index order (products of 16-bit values are exact); everything above that in fp64.  The fp32
        // part rounds at 2^-24 of those 8 elements' own sum; torch's reduction is an fp32 cascade throughout.
        float fxx = 0.0f, fyy = 0.0f, fxy = 0.0f;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            fxx = __builtin_fmaf(x[j], x[j], fxx);
            fyy = __builtin_fmaf(y[j], y[j], fyy);
            fxy = __builtin_fmaf(x[j], y[j], fxy);
        }
        axx += (double)fxx; ayy += (double)fyy; axy += (double)fxy;
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    issue(t0, S0{});
    for (long long t = t0; t < t1; t += 2) {
        if (t + 1 < t1) issue(t + 1, S1{});
        consume(t, S0{});
        if (t + 2 < t1) issue(t + 2, S0{});
        if (t + 1 < t1) consume(t + 1, S1{});
    }
    flush();
}

// The flat form of the same pass (one block per tile, every wave writes its own partial; chunk = 1 for the finish pass): with
// THREE input streams a tile carries enough bytes per fold that the walk's looped loads lose more (4.0 ms) than the folds cost
// (3.3 ms for the 20.8 GB of the SDXL x3 job = 6.3 TB/s); with two streams it is the other way round (3.0 against 2.3 ms).
template <int MODE, int IN_DT>
__global__ __launch_bounds__(MECHA_BLOCK) void bred_reduce_flat_kernel(BredParams p) {
    const BredTile rec = p.tiles[blockIdx.x];
    const int local = threadIdx.x * 8;
    const int valid = rec.valid - local >= 8 ? 8 : (rec.valid - local > 0 ? rec.valid - local : 0);
    float fxx = 0.0f, fyy = 0.0f, fxy = 0.0f;
    bool bad_in = false;
    if (valid > 0) {
        float x[8], y[8], a0[8];
        bred_operands<MODE, IN_DT>(p, rec.elem + local, valid, p.vec_ok && valid == 8, x, y, a0, bad_in);
#pragma unroll
        for (int j = 0; j < 8; j++) {
            fxx = __builtin_fmaf(x[j], x[j], fxx);
            fyy = __builtin_fmaf(y[j], y[j], fyy);
            fxy = __builtin_fmaf(x[j], y[j], fxy);
        }
    }
    double sxx = fxx, syy = fyy, sxy = fxy;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        sxx += __shfl_down(sxx, off, 64);
        syy += __shfl_down(syy, off, 64);
        sxy += __shfl_down(sxy, off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        double *dst = p.partial + ((long long)blockIdx.x * BRED_WAVES + (threadIdx.x >> 6)) * 3;
        dst[0] = sxx; dst[1] = syy; dst[2] = sxy;
    }
    if (bad_in && p.flags != nullptr) atomicOr(p.flags + rec.seg, MECHA_FLAG_INPUT_NONFINITE);
}

template <int MODE>
__global__ __launch_bounds__(MECHA_BLOCK) void bred_finish_kernel(BredParams p, long long chunk) {
    const int s = blockIdx.x;
    // the key's partials sit at its first tile and at every multiple of `chunk` inside it (where another block took over)
    const long long first = p.tile_start[s], end = p.tile_start[s + 1];
    const long long k0 = first / chunk + 1;                                   // first block boundary strictly after `first`
    const long long k1 = (end - 1) / chunk;                                   // last block boundary before `end`
    const long long n_runs = 1 + (k1 >= k0 ? k1 - k0 + 1 : 0);
    double s0 = 0, s1 = 0, s2 = 0;
    for (long long i = threadIdx.x; i < n_runs * BRED_WAVES; i += MECHA_BLOCK) {
        const long long run = i / BRED_WAVES;
        const long long t = run == 0 ? first : (k0 + run - 1) * chunk;
        const double *src = p.partial + (t * BRED_WAVES + i % BRED_WAVES) * 3;
        s0 += src[0]; s1 += src[1]; s2 += src[2];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { s0 += __shfl_down(s0, off, 64); s1 += __shfl_down(s1, off, 64); s2 += __shfl_down(s2, off, 64); }
    __shared__ double wsum[MECHA_BLOCK / 64][3];
    if ((threadIdx.x & 63) == 0) { wsum[threadIdx.x >> 6][0] = s0; wsum[threadIdx.x >> 6][1] = s1; wsum[threadIdx.x >> 6][2] = s2; }
    __syncthreads();
    if (threadIdx.x != 0) return;
    s0 = s1 = s2 = 0;
    for (int w = 0; w < MECHA_BLOCK / 64; w++) { s0 += wsum[w][0]; s1 += wsum[w][1]; s2 += wsum[w][2]; }
    RedScalars sc{};
    sc.na = (float)__builtin_sqrt(s0);
    sc.nb = (float)__builtin_sqrt(s1);
    if constexpr (MODE == RM_SLERP) {
        float ab = (float)(s2 / ((double)sc.na * (double)sc.nb));
        ab = (ab < -1.0f) ? -1.0f : ((ab > 1.0f) ? 1.0f : ab);
        const float omega = acosf(ab);
        sc.sa = sinf((1.0f - p.alpha) * omega);
        sc.sb = sinf(p.alpha * omega);
        sc.so = sinf(omega);
        sc.nl = lerp_torch(sc.na, sc.nb, p.alpha, p.alpha_m1, p.small != 0);
    } else {
        sc.proj = (float)(s2 / ((double)sc.na * (double)sc.na));
    }
    sc.any_nan = 0;
    sc.bad_out = 0;
    // x / d as q0 = RN(x y), q = RN(q0 + RN(x - d q0) y) with y = RN(1 / d): the correctly rounded quotient (Markstein) unless d's
    // significand is all ones or an intermediate leaves the normal range; divisors outside [2^-60, 2^60] keep the IEEE division
    auto safe = [](float d) {
        const uint32_t b = __float_as_uint(d) & 0x7fffffffu;
        return b >= 0x21800000u && b <= 0x5d800000u && (b & 0x7fffffu) != 0x7fffffu;
    };
    sc.rna = 1.0f / sc.na; sc.rnb = 1.0f / sc.nb; sc.rso = 1.0f / sc.so;
    sc.fast_div = (MODE == RM_SLERP && safe(sc.na) && safe(sc.nb) && safe(sc.so)) ? 1u : 0u;
    p.sc[s] = sc;
}

__device__ __forceinline__ float bred_div(float x, float d, float rd) {
    const float q0 = x * rd;
    return __builtin_fmaf(__builtin_fmaf(-d, q0, x), rd, q0);
}

template <int MODE, bool FAST>
__device__ __forceinline__ float bred_formula(const RedScalars &sc, float x, float y) {
    if constexpr (MODE == RM_SLERP) {
        // three IEEE divisions per element cost ~35 of the ~47 VALU instructions per element this pass can afford at HBM speed
        if constexpr (FAST) {
            const float an = bred_div(x, sc.na, sc.rna), bn = bred_div(y, sc.nb, sc.rnb);
            const float q = bred_div(an * sc.sa + bn * sc.sb, sc.so, sc.rso);
            return q * sc.nl;
        }
        const float an = x / sc.na, bn = y / sc.nb;              // linear.py:49-50
        float q = (an * sc.sa + bn * sc.sb) / sc.so;             // linear.py:55-57
        return q * sc.nl;                                        // linear.py:58
    } else {
        return y - x * sc.proj;                                  // linear.py:110
    }
}

template <int MODE, int IN_DT, int OUT_DT>
__global__ __launch_bounds__(MECHA_BLOCK) void bred_apply_kernel(BredParams p) {
    const BredTile rec = p.tiles[blockIdx.x];
    const int local = threadIdx.x * 8;
    const int valid = rec.valid - local >= 8 ? 8 : (rec.valid - local > 0 ? rec.valid - local : 0);
    if (valid == 0) return;
    const int64_t e = rec.elem + local;
    const bool vec = p.vec_ok && valid == 8;
    float x[8], y[8], a0[8], r[8];
    bool bad_in = false, any_nan = false, bad_out = false;
    bred_operands<MODE, IN_DT>(p, e, valid, vec, x, y, a0, bad_in);     // the loads are issued before the key's scalars are needed
    const RedScalars sc = p.sc[rec.seg];
    bool redo = false;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        float q = sc.fast_div ? bred_formula<MODE, true>(sc, x[j], y[j]) : bred_formula<MODE, false>(sc, x[j], y[j]);
        // non-zero operands below 2^-40 could leave the normal range on the way (the divisors lie in [2^-60, 2^60]): IEEE division then
        if constexpr (MODE == RM_SLERP)
            redo |= (((__float_as_uint(x[j]) & 0x7fffffffu) - 1u) < 0x2b800000u - 1u) | (((__float_as_uint(y[j]) & 0x7fffffffu) - 1u) < 0x2b800000u - 1u);
        r[j] = q;
    }
    if (redo && sc.fast_div) {
#pragma unroll
        for (int j = 0; j < 8; j++) r[j] = bred_formula<MODE, false>(sc, x[j], y[j]);
    }
#pragma unroll
    for (int j = 0; j < 8; j++) {
        float q = r[j];
        if (j < valid) any_nan |= (q != q);
        if constexpr (MODE == RM_ADDPERP) q = __builtin_fmaf(q, p.alpha, a0[j]);      // add_difference(a, perp, alpha)  (linear.py:76)
        r[j] = q;
        bad_out |= (j < valid) && nonfinite_f32(q);
    }
    if (vec) store8<OUT_DT, true>(p.out, e, r);
    else {
#pragma unroll
        for (int j = 0; j < 8; j++)
            if (j < valid) store1<OUT_DT>(p.out, e + j, r[j]);
    }
    if (any_nan && atomicOr(&p.sc[rec.seg].any_nan, 1u) == 0u) p.nan_keys[atomicAdd(&p.globals->n_nan_keys, 1u)] = rec.seg;
    // a key with a NaN is rewritten as a whole by fixup: what its discarded result looked like must not reach the flag word
    // (check_finite_output sees the method's RETURN value, merging.py:297-308).  bred_flags_kernel sorts that out per key.
    if (bad_out) atomicOr(&p.sc[rec.seg].bad_out, 1u);
}

__global__ __launch_bounds__(MECHA_BLOCK) void bred_flags_kernel(BredParams p) {
    const int s = blockIdx.x * MECHA_BLOCK + threadIdx.x;
    if (s >= p.n_segs || p.flags == nullptr) return;
    if (p.sc[s].bad_out != 0 && p.sc[s].any_nan == 0) atomicOr(p.flags + s, MECHA_FLAG_OUTPUT_NONFINITE);
}

// The reference's `if res.isnan().any()` branches, for the keys the apply pass noted (rare: an all-zero tensor, identical inputs).
template <int MODE, int IN_DT, int OUT_DT>
__global__ __launch_bounds__(MECHA_BLOCK) void bred_fixup_kernel(BredParams p) {
    const uint32_t n_list = p.globals->n_nan_keys;
    for (uint32_t li = 0; li < n_list; li++) {
    const int s = p.nan_keys[li];
    for (long long t = p.tile_start[s] + blockIdx.x; t < p.tile_start[s + 1]; t += gridDim.x) {
        const BredTile rec = p.tiles[t];
        const int local = threadIdx.x * 8;
        const int valid = rec.valid - local >= 8 ? 8 : (rec.valid - local > 0 ? rec.valid - local : 0);
        if (valid == 0) continue;
        const int64_t e = rec.elem + local;
        const bool vec = p.vec_ok && valid == 8;
        float a[8], b[8], r[8];
        auto ld = [&](const void *ptr, float (&v)[8]) {
            if (vec) load8<IN_DT, false>(ptr, e, v);
            else {
#pragma unroll
                for (int j = 0; j < 8; j++) v[j] = (j < valid) ? load1<IN_DT>(ptr, e + j) : 0.0f;
            }
        };
        if constexpr (MODE == RM_SLERP) {
            ld(p.a, a); ld(p.b, b);
#pragma unroll
            for (int j = 0; j < 8; j++) r[j] = lerp_torch(a[j], b[j], p.alpha, p.alpha_m1, p.small != 0);   // linear.py:60
        } else if constexpr (MODE == RM_PERP) {
#pragma unroll
            for (int j = 0; j < 8; j++) r[j] = 0.0f;                                                      // linear.py:112
        } else {
            ld(p.a, a);
#pragma unroll
            for (int j = 0; j < 8; j++) r[j] = __builtin_fmaf(0.0f, p.alpha, a[j]);                       // a.addcmul(zeros, alpha)
        }
        bool bad_out = false;
#pragma unroll
        for (int j = 0; j < 8; j++) bad_out |= (j < valid) && nonfinite_f32(r[j]);
        if (vec) store8<OUT_DT, false>(p.out, e, r);
        else {
#pragma unroll
            for (int j = 0; j < 8; j++)
                if (j < valid) store1<OUT_DT>(p.out, e + j, r[j]);
        }
        if (bad_out && p.flags != nullptr) atomicOr(p.flags + rec.seg, MECHA_FLAG_OUTPUT_NONFINITE);
    }
    }
}

size_t bred_align(size_t v) { return (v + 255) / 256 * 256; }

struct BredLayout { size_t tile_start_off, tiles_off, partial_off, sc_off, globals_off, nan_keys_off, total; size_t max_tiles; };
BredLayout bred_layout(int n_segs, int64_t total_elems) {
    BredLayout w;
    w.max_tiles = (size_t)(total_elems > 0 ? total_elems : 0) / BRED_TILE + (size_t)n_segs;
    size_t off = 0;
    w.tile_start_off = off; off = bred_align(off + sizeof(long long) * ((size_t)n_segs + 1));
    w.tiles_off = off;      off = bred_align(off + sizeof(BredTile) * w.max_tiles);
    w.partial_off = off;    off = bred_align(off + sizeof(double) * 3 * BRED_WAVES * w.max_tiles);
    w.sc_off = off;         off = bred_align(off + sizeof(RedScalars) * (size_t)n_segs);
    w.globals_off = off;    off = bred_align(off + sizeof(BredGlobals));
    w.nan_keys_off = off;   off = bred_align(off + sizeof(int) * (size_t)n_segs);
    w.total = off;
    return w;
}

template <int MODE>
int run_batched_reduce(const void *a, const void *b, const void *c, void *out, const mecha_segment *segs_dev, const mecha_segment *segs_host,
                       int n_segs, int64_t total_elems, int in_dt, int out_dt, float alpha, void *workspace, size_t workspace_bytes,
                       uint32_t *flags, void *stream) {
    MECHA_REQUIRE(a != nullptr && b != nullptr && out != nullptr && (MODE != RM_ADDPERP || c != nullptr), "NULL arena pointer");
    MECHA_REQUIRE(segs_dev != nullptr && segs_host != nullptr && n_segs >= 1, "batched call needs host and device segment tables");
    MECHA_REQUIRE(mecha_valid_dtype(in_dt) && mecha_valid_dtype(out_dt), "unknown dtype code (in %d, out %d)", in_dt, out_dt);
    long long n_tiles = 0;
    int64_t prev_end = 0;
    for (int s = 0; s < n_segs; s++) {
        MECHA_REQUIRE(segs_host[s].numel > 0 && (segs_host[s].offset % 8) == 0 && segs_host[s].offset >= prev_end,
                      "segment %d: offsets must be multiples of 8, sorted and non-overlapping, numel > 0", s);
        prev_end = segs_host[s].offset + segs_host[s].numel;
        n_tiles += (segs_host[s].numel + BRED_TILE - 1) / BRED_TILE;
    }
    MECHA_REQUIRE(prev_end <= total_elems, "segments end at element %lld, beyond total_elems = %lld", (long long)prev_end, (long long)total_elems);
    const BredLayout w = bred_layout(n_segs, total_elems);
    MECHA_REQUIRE((size_t)n_tiles <= w.max_tiles, "segments hold more tiles than the workspace was sized for");
    MECHA_REQUIRE(workspace != nullptr && workspace_bytes >= w.total && mecha_aligned16(workspace),
                  "reduce workspace too small or unaligned: need %zu bytes, got %zu", w.total, workspace_bytes);
    MECHA_REQUIRE(n_tiles * MECHA_BLOCK < (1ll << 32), "too many tiles for one launch (2^24 - 1 tiles of 2048 elements)");
    char *ws = static_cast<char *>(workspace);
    BredParams p{};
    p.a = a; p.b = b; p.c = c; p.out = out; p.segs = segs_dev; p.n_segs = n_segs;
    p.tile_start = reinterpret_cast<const long long *>(ws + w.tile_start_off);
    p.tiles = reinterpret_cast<const BredTile *>(ws + w.tiles_off);
    p.n_tiles = n_tiles;
    p.partial = reinterpret_cast<double *>(ws + w.partial_off);
    p.sc = reinterpret_cast<RedScalars *>(ws + w.sc_off);
    p.globals = reinterpret_cast<BredGlobals *>(ws + w.globals_off);
    p.nan_keys = reinterpret_cast<int *>(ws + w.nan_keys_off);
    p.alpha = alpha; p.alpha_m1 = alpha - 1.0f; p.small = (alpha < 0.5f && alpha > -0.5f) ? 1 : 0;
    p.vec_ok = (mecha_aligned16(a) && mecha_aligned16(b) && mecha_aligned16(out) && (c == nullptr || mecha_aligned16(c))) ? 1 : 0;
    p.flags = flags;
    hipStream_t s = static_cast<hipStream_t>(stream);
    MECHA_CHECK_HIP(hipMemsetAsync(ws + w.globals_off, 0, sizeof(BredGlobals), s));
    MECHA_LAUNCH(MK_TIES_INDEX, bred_tile_index_kernel, dim3(1), dim3(1024), 0, s, segs_dev, n_segs, reinterpret_cast<long long *>(ws + w.tile_start_off));
    MECHA_LAUNCH(MK_TIES_INDEX, bred_tile_records_kernel, dim3((unsigned)n_segs), dim3(MECHA_BLOCK), 0, s, segs_dev, p.tile_start,
                 reinterpret_cast<BredTile *>(ws + w.tiles_off));
    const unsigned grid = (unsigned)n_tiles;
    // the reduce pass: 8 resident blocks per CU, each with a contiguous run of tiles
    const long long chunk = MODE == RM_ADDPERP ? 1 : (n_tiles + MECHA_MAX_GRID - 1) / MECHA_MAX_GRID;
    const unsigned red_grid = (unsigned)((n_tiles + chunk - 1) / chunk);
    const unsigned fix_grid = (unsigned)(n_tiles < 2048 ? n_tiles : 2048);
    MECHA_DISPATCH_DTYPES(in_dt, out_dt, {
        if constexpr (MODE == RM_ADDPERP) MECHA_LAUNCH(MK_RED_REDUCE, (bred_reduce_flat_kernel<MODE, IN_DT>), dim3(grid), dim3(MECHA_BLOCK), 0, s, p);
        else MECHA_LAUNCH(MK_RED_REDUCE, (bred_reduce_kernel<MODE, IN_DT>), dim3(red_grid), dim3(MECHA_BLOCK), 0, s, p, chunk);
        MECHA_LAUNCH(MK_RED_FINISH, (bred_finish_kernel<MODE>), dim3((unsigned)n_segs), dim3(MECHA_BLOCK), 0, s, p, chunk);
        MECHA_LAUNCH(MK_RED_APPLY, (bred_apply_kernel<MODE, IN_DT, OUT_DT>), dim3(grid), dim3(MECHA_BLOCK), 0, s, p);
        MECHA_LAUNCH(MK_RED_FIXUP, (bred_fixup_kernel<MODE, IN_DT, OUT_DT>), dim3(fix_grid), dim3(MECHA_BLOCK), 0, s, p);
        MECHA_LAUNCH(MK_RED_FIXUP, bred_flags_kernel, dim3((unsigned)((n_segs + MECHA_BLOCK - 1) / MECHA_BLOCK)), dim3(MECHA_BLOCK), 0, s, p);
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

size_t mecha_batched_reduce_workspace_bytes(int n_segments, int64_t total_elems) {
    return bred_layout(n_segments < 1 ? 1 : n_segments, total_elems).total;
}

int mecha_batched_slerp(const void *a_arena, const void *b_arena, void *out_arena, const mecha_segment *segments,
                        const mecha_segment *segments_host, int n_segments, int64_t total_elems, int in_dtype, int out_dtype,
                        float alpha, void *workspace, size_t workspace_bytes, uint32_t *seg_flags, void *stream) {
    return run_batched_reduce<RM_SLERP>(a_arena, b_arena, nullptr, out_arena, segments, segments_host, n_segments, total_elems, in_dtype,
                                        out_dtype, alpha, workspace, workspace_bytes, seg_flags, stream);
}

int mecha_batched_perpendicular_component(const void *a_arena, const void *b_arena, void *out_arena, const mecha_segment *segments,
                                          const mecha_segment *segments_host, int n_segments, int64_t total_elems, int in_dtype,
                                          int out_dtype, void *workspace, size_t workspace_bytes, uint32_t *seg_flags, void *stream) {
    return run_batched_reduce<RM_PERP>(a_arena, b_arena, nullptr, out_arena, segments, segments_host, n_segments, total_elems, in_dtype,
                                       out_dtype, 0.0f, workspace, workspace_bytes, seg_flags, stream);
}

int mecha_batched_add_perpendicular(const void *a_arena, const void *b_arena, const void *c_arena, void *out_arena,
                                    const mecha_segment *segments, const mecha_segment *segments_host, int n_segments,
                                    int64_t total_elems, int in_dtype, int out_dtype, float alpha, void *workspace,
                                    size_t workspace_bytes, uint32_t *seg_flags, void *stream) {
    MECHA_REQUIRE(!mecha_isclose(alpha, 0.0), "alpha == 0: add_difference returns `a` untouched (linear.py:72-73); use the cast entry point");
    return run_batched_reduce<RM_ADDPERP>(a_arena, b_arena, c_arena, out_arena, segments, segments_host, n_segments, total_elems, in_dtype,
                                          out_dtype, alpha, workspace, workspace_bytes, seg_flags, stream);
}

}  // extern "C"
