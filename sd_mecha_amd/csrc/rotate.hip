// rotate.hip -- orthogonal Procrustes alignment of one key (reference svd.py:10-86 at alignment = 1, alpha = 0).
//
//   a_2d, b_2d : the key reshaped to n x p (svd.py:49-53)
//   a_mean = a_2d.mean(0) * centralization, b_mean likewise; Ac = a_2d - a_mean, Bc = b_2d - b_mean   (svd.py:67-70)
//   n >= p :  M = Ac^T Bc  (p x p);  M = U S V^T;  R = U V^T;  out = Ac R + b_mean                    (svd.py:165-169, :84-85)
//   n <  p :  b.mH = Qb T with Qb an orthonormal basis of the row space of Bc (the reference takes a QR, svd.py:161; any
//             orthonormal factorisation gives the same transform);  N = Ac^T T^T (p x n) = U S W^T;  V = Qb W;
//             out = Ac - (Ac V) V^T + (Ac U) V^T + b_mean                                             (svd.py:160-163, :181)
//   lerp(a_mean, b_mean, alignment = 1) = b_mean exactly (Lerp.h: weight >= 0.5 branch, fma(0, d, b)).
//
// The transform only ever uses U V^T, never U, S or V on their own: U V^T is the orthogonal POLAR FACTOR of the matrix, and
// the polar factor is reachable with matrix products alone (Newton-Schulz, polar_newton_schulz below).  So there is no SVD
// here: every step is a GEMM on the matrix cores.  (A one-sided Jacobi SVD, the first version of this file, spent > 99 % of
// a key's time in ~25 000 tiny dependent launches per 1280 x 1280 key.)  The same iteration gives the orthonormal basis Qb
// of the wide case (polar factor of Bc) and U W^T (polar factor of N); with Z = polar(N^T):  U V^T = Z^T Qb^T.
//
// Kernels
//   rot_colsum / rot_means / rot_center : column means (fp64 accumulation) and the centred fp32 copies Ac, Bc
//   gemm_kernel   : C[m][n] = sum_k A(k, m) B(k, n) with arbitrary element strides, on v_mfma_f32_32x32x2_f32 (exact fp32
//                   fma chains); split-K slabs + an in-order reduce for skinny outputs (conv 3x3 keys: p = 9, n ~ 1.6 M);
//                   fused epilogue (a X + b C update, minus / plus a matrix, per-column bias, cast to the output dtype, flag)
//   rot_gram_norms: |G|_F^2 and |G - I|_F^2 of the Gram matrix, the convergence measure the host reads once per iteration
// The SVD's sign/order ambiguity does not exist for the polar factor; parity with the reference is <= 1e-3 relative
// Frobenius error (wide keys with centring: on the component the reference determines, DESIGN.md "rotate: reference
// indeterminacy").  Roofline: MFMA (f32).  Flops, with q = min(n, p), c = max over the polar solves' column count:
// 2 n p^2 for each of the A^T B contraction and the final application, 4 q^2 c per Newton-Schulz iteration.
#include <stdlib.h>
#include <type_traits>
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// A launch covers `count` keys of ONE 2-D shape.  Every key has the same private workspace layout, `ws_bytes` apart, so a
// workspace pointer of key g is the pointer of key 0 plus g * ws_bytes; the tensors themselves sit in the per-model arenas at
// offs[g] elements (nullptr: one key at offset 0).  `skip` (nullable): per-key device words; a block whose key's word is
// non-zero exits at once -- how the keys of a batch that have converged leave the later polar steps without the host
// learning about it.
struct Batch {
    int count;
    int64_t ws_bytes;
    const int64_t *offs;
    const uint32_t *skip;
};
// Slots of a key's block of 8 doubles (`scal`): 0-1 Gram norms, 2 sum of squares, 3 trace (truncate_rank), 4 the done word of a
// single key, 5 the bit patterns of max |a| and max |b| (two uint32), 6 the image scales t_a, t_b (two floats: powers of two that
// bring the centred matrices' entries to <= 2, so that their split-fp16 images cannot overflow whatever the weights' range), 7 the
// float 1 / t_a.
constexpr int ROT_SCAL_AMAX = 5, ROT_SCAL_T = 6, ROT_SCAL_INV_TA = 7;
template <typename T> __device__ __forceinline__ T *key_ptr(T *p, const Batch &bt, int key) {
    return p == nullptr ? nullptr : reinterpret_cast<T *>(reinterpret_cast<char *>(const_cast<typename std::remove_const<T>::type *>(p)) + (int64_t)key * bt.ws_bytes);
}
__device__ __forceinline__ int64_t key_off(const Batch &bt, int key) { return bt.offs != nullptr ? bt.offs[key] : 0; }

// ------------------------------------------------------------------------------------------------ centring
__device__ __forceinline__ float ld_any(int dt, const void *p, int64_t i) {
    return (dt == MECHA_BF16) ? load1<MECHA_BF16>(p, i) : (dt == MECHA_F16) ? load1<MECHA_F16>(p, i) : load1<MECHA_F32>(p, i);
}

// partial column sums: grid (ceil(p/64), chunks); block 256 = 4 row lanes x 64 columns
__global__ __launch_bounds__(256) void rot_colsum_kernel(const void *a, const void *b, int dt, int64_t n, int64_t p, int chunks,
                                                         double *part /* [2][chunks][p] */, uint32_t *flags, Batch bt) {
    const int key = blockIdx.z;
    const int64_t base = key_off(bt, key);
    part = key_ptr(part, bt, key);
    if (flags != nullptr) flags += key;
    const int c = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int64_t col = (int64_t)blockIdx.x * 64 + c;
    const int64_t rows_per = (n + chunks - 1) / chunks;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per, r1 = (r0 + rows_per < n) ? r0 + rows_per : n;
    double sa = 0.0, sb = 0.0;
    bool bad = false;
    if (col < p) {
        for (int64_t r = r0 + rl; r < r1; r += 4) {
            const float x = ld_any(dt, a, base + r * p + col), y = ld_any(dt, b, base + r * p + col);
            bad |= nonfinite_f32(x) | nonfinite_f32(y);
            sa += (double)x; sb += (double)y;
        }
    }
    __shared__ double sh[2][4][64];
    sh[0][rl][c] = sa; sh[1][rl][c] = sb;
    __syncthreads();
    if (rl == 0 && col < p) {
        part[((int64_t)0 * chunks + blockIdx.y) * p + col] = sh[0][0][c] + sh[0][1][c] + sh[0][2][c] + sh[0][3][c];
        part[((int64_t)1 * chunks + blockIdx.y) * p + col] = sh[1][0][c] + sh[1][1][c] + sh[1][2][c] + sh[1][3][c];
    }
    if (bad && flags != nullptr) atomicOr(flags, MECHA_FLAG_INPUT_NONFINITE);
}

// The same for p % 8 == 0 and 16-byte-aligned keys (every key of an arena), with 16-byte loads: a lane owns 8 consecutive
// columns, a wave a row of 512, the block's four waves take every fourth row of the chunk.  The same pass evaluates
// torch.allclose(a.half(), b.half()) (svd.py:46) when `not_close` is given -- both read a and b once, whole.
// grid (ceil(p / 512), chunks, keys).
template <int DT>
__global__ __launch_bounds__(256) void rot_colsum8_kernel(const void *a, const void *b, int64_t n, int64_t p, int chunks, double *part /* [2][chunks][p] */,
                                                          uint32_t *flags, uint32_t *not_close, double *scal, Batch bt) {
    __shared__ double sh[2][3][512];
    const int key = blockIdx.z;
    const int64_t base = key_off(bt, key);
    part = key_ptr(part, bt, key);
    uint32_t *amax = reinterpret_cast<uint32_t *>(key_ptr(scal, bt, key) + ROT_SCAL_AMAX);      // bit patterns of max |a|, max |b| (zeroed by the host)
    float ma = 0.0f, mb = 0.0f;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t col = (int64_t)blockIdx.x * 512 + 8 * lane;
    const int64_t rows_per = (n + chunks - 1) / chunks;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per, r1 = (r0 + rows_per < n) ? r0 + rows_per : n;
    double sa[8], sb[8];
#pragma unroll
    for (int j = 0; j < 8; j++) { sa[j] = 0.0; sb[j] = 0.0; }
    bool bad = false, differ = false;
    if (col < p) {
        for (int64_t r = r0 + wave; r < r1; r += 4) {
            float x[8], y[8];
            load8<DT>(a, base + r * p + col, x);
            load8<DT>(b, base + r * p + col, y);
#pragma unroll
            for (int j = 0; j < 8; j++) {
                bad |= nonfinite_f32(x[j]) | nonfinite_f32(y[j]);
                sa[j] += (double)x[j]; sb[j] += (double)y[j];
                ma = __builtin_fmaxf(ma, __builtin_fabsf(x[j])); mb = __builtin_fmaxf(mb, __builtin_fabsf(y[j]));
                const float xh = (float)(_Float16)x[j], yh = (float)(_Float16)y[j];
                differ |= !((xh == yh) || (__builtin_fabsf(xh - yh) <= 1e-8f + 1e-5f * __builtin_fabsf(yh)));
            }
        }
    }
    if (wave > 0) {
#pragma unroll
        for (int j = 0; j < 8; j++) { sh[0][wave - 1][8 * lane + j] = sa[j]; sh[1][wave - 1][8 * lane + j] = sb[j]; }
    }
    __syncthreads();
    if (wave == 0 && col < p) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int c = 8 * lane + j;
            part[((int64_t)0 * chunks + blockIdx.y) * p + col + j] = sa[j] + sh[0][0][c] + sh[0][1][c] + sh[0][2][c];
            part[((int64_t)1 * chunks + blockIdx.y) * p + col + j] = sb[j] + sh[1][0][c] + sh[1][1][c] + sh[1][2][c];
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { ma = __builtin_fmaxf(ma, __shfl_xor(ma, off, 64)); mb = __builtin_fmaxf(mb, __shfl_xor(mb, off, 64)); }
    if (lane == 0) { atomicMax(amax, __float_as_uint(ma)); atomicMax(amax + 1, __float_as_uint(mb)); }       // (non-negative floats order like their patterns)
    if (bad && flags != nullptr) atomicOr(flags + key, MECHA_FLAG_INPUT_NONFINITE);
    if (differ && not_close != nullptr) atomicOr(not_close + key, 1u);
}

// Keys with very few columns (3 x 3 convolutions flatten to (out * in) x 9, svd.py:49-53): rot_colsum_kernel would use p of its 64
// column lanes.  Here a thread owns whole rows -- p consecutive values each, neighbouring threads neighbouring rows -- and keeps
// the p column sums in registers.  grid (1, chunks, keys); P_MAX = 16.
constexpr int COLSUM_SMALL_P = 16;
__global__ __launch_bounds__(256) void rot_colsum_small_kernel(const void *a, const void *b, int dt, int64_t n, int p, int chunks, double *part, uint32_t *flags,
                                                               Batch bt) {
    __shared__ double sh[2][4][COLSUM_SMALL_P];
    const int key = blockIdx.z;
    const int64_t base = key_off(bt, key);
    part = key_ptr(part, bt, key);
    const int64_t rows_per = (n + chunks - 1) / chunks;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per, r1 = (r0 + rows_per < n) ? r0 + rows_per : n;
    double sa[COLSUM_SMALL_P], sb[COLSUM_SMALL_P];
#pragma unroll
    for (int j = 0; j < COLSUM_SMALL_P; j++) { sa[j] = 0.0; sb[j] = 0.0; }
    bool bad = false;
    for (int64_t r = r0 + threadIdx.x; r < r1; r += 256) {
#pragma unroll
        for (int j = 0; j < COLSUM_SMALL_P; j++)
            if (j < p) {
                const float x = ld_any(dt, a, base + r * p + j), y = ld_any(dt, b, base + r * p + j);
                bad |= nonfinite_f32(x) | nonfinite_f32(y);
                sa[j] += (double)x; sb[j] += (double)y;
            }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int j = 0; j < COLSUM_SMALL_P; j++) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { sa[j] += __shfl_down(sa[j], off, 64); sb[j] += __shfl_down(sb[j], off, 64); }
        if (lane == 0) { sh[0][wave][j] = sa[j]; sh[1][wave][j] = sb[j]; }
    }
    __syncthreads();
    if (threadIdx.x < 2 * COLSUM_SMALL_P) {
        const int which = threadIdx.x / COLSUM_SMALL_P, j = threadIdx.x % COLSUM_SMALL_P;
        if (j < p) part[((int64_t)which * chunks + blockIdx.y) * p + j] = sh[which][0][j] + sh[which][1][j] + sh[which][2][j] + sh[which][3][j];
    }
    if (bad && flags != nullptr) atomicOr(flags + key, MECHA_FLAG_INPUT_NONFINITE);
}

__global__ __launch_bounds__(256) void rot_means_kernel(const double *part, int chunks, int64_t n, int64_t p, float cz, float t,
                                                        float *a_mean, float *b_mean, float *t_mean, double *scal, Batch bt) {
    const int key = blockIdx.y;
    if (scal != nullptr && blockIdx.x == 0 && threadIdx.x == 0) {         // the image scales (ROT_SCAL_T) from the largest magnitudes
        double *sk = key_ptr(scal, bt, key);
        const uint32_t *amax = reinterpret_cast<const uint32_t *>(sk + ROT_SCAL_AMAX);
        float *ts = reinterpret_cast<float *>(sk + ROT_SCAL_T);
        float tv[2];
        for (int w = 0; w < 2; w++) {
            const float m = __uint_as_float(amax[w]);
            float e = (m > 0.0f && m < 3e38f) ? -ceilf(log2f(m)) : 0.0f;
            e = e < -100.0f ? -100.0f : (e > 100.0f ? 100.0f : e);
            tv[w] = exp2f(e);
        }
        ts[0] = tv[0]; ts[1] = tv[1];
        *reinterpret_cast<float *>(sk + ROT_SCAL_INV_TA) = 1.0f / tv[0];
    }
    part = key_ptr(part, bt, key); a_mean = key_ptr(a_mean, bt, key); b_mean = key_ptr(b_mean, bt, key); t_mean = key_ptr(t_mean, bt, key);
    const int64_t col = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (col >= p) return;
    double sa = 0.0, sb = 0.0;
    for (int k = 0; k < chunks; k++) { sa += part[((int64_t)0 * chunks + k) * p + col]; sb += part[((int64_t)1 * chunks + k) * p + col]; }
    a_mean[col] = (float)(sa / (double)n) * cz;          // a_2d.mean(0) * centralization  (svd.py:67)
    b_mean[col] = (float)(sb / (double)n) * cz;
    // torch.lerp(a_mean, b_mean, alignment)  (svd.py:85; Lerp.h:12-34)
    const float am = a_mean[col], bm = b_mean[col];
    t_mean[col] = fabsf(t) < 0.5f ? __builtin_fmaf(t, bm - am, am) : bm - (bm - am) * (1.0f - t);
}

__global__ __launch_bounds__(256) void rot_center_kernel(const void *a, const void *b, int dt, int64_t n, int64_t p, const float *a_mean,
                                                         const float *b_mean, float *ac, float *bc, Batch bt) {
    const int key = blockIdx.y;
    const int64_t base = key_off(bt, key);
    a_mean = key_ptr(a_mean, bt, key); b_mean = key_ptr(b_mean, bt, key); ac = key_ptr(ac, bt, key); bc = key_ptr(bc, bt, key);
    const int64_t total = n * p;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t c = i % p;
        ac[i] = ld_any(dt, a, base + i) - a_mean[c];      // a_2d - a_mean  (svd.py:69-70)
        bc[i] = ld_any(dt, b, base + i) - b_mean[c];
    }
}

// ------------------------------------------------------------------------------------------------ GEMM
// C[m][n] = sum_k A[k * a_ks + m * a_ms] * B[k * b_ks + n * b_ns],  C row-major with leading dimension N.
// Wave tile 32 x 32, block = 4 waves = 64 x 64, gridDim.z = split-K slabs.
struct Operand { const float *p; int64_t ks, ms; };
struct GemmEpilogue {
    const float *minus;         // v = v - minus[idx]
    const float *plus;          // v = plus[idx] + v
    const float *col_bias;      // v = v + col_bias[col]
    void *out;                  // if set: write v in out_dt (and raise the finite flag); else fp32 C
    int out_dt;
    uint32_t *flags;
    const float *axpy;          // if set: v = axpy_a * axpy[idx] + axpy_b * v   (the Newton-Schulz update), applied first
    float axpy_a, axpy_b;
    const float *mix;           // if set: v = mix_w0 * mix[idx] + mix_w1 * v, after minus / plus (the alpha interpolation, svd.py:82)
    float mix_w0, mix_w1;
    int64_t out_off;            // element offset of this key inside `out` (set per block from the batch descriptor)
    // gemm128h_kernel only (its epilogue knows nothing but these and axpy_a / axpy_b; C may be NULL there):
    _Float16 *sp, *spt;         // if set: the result / its transpose ALSO go out as split-fp16 images (that kernel's operand format)
    const _Float16 *axpy_img;   // the axpy term, read from its split image (22 of its 24 significand bits)
    const float *acc_scale;     // if set: one float per key, multiplied in after the axpy and before col_bias (1 / t_a of scaled images)
    int h;                      // set by gemm_h: tells gemm_reduce_kernel whose epilogue to run
};
// the epilogue of key `key` of a batch: workspace pointers move by the key's workspace stride, `out` by its arena offset
__device__ __forceinline__ GemmEpilogue epilogue_for_key(GemmEpilogue e, const Batch &bt, int key) {
    e.minus = key_ptr(e.minus, bt, key); e.plus = key_ptr(e.plus, bt, key); e.col_bias = key_ptr(e.col_bias, bt, key);
    e.axpy = key_ptr(e.axpy, bt, key); e.mix = key_ptr(e.mix, bt, key); e.sp = key_ptr(e.sp, bt, key); e.spt = key_ptr(e.spt, bt, key); e.axpy_img = key_ptr(e.axpy_img, bt, key); e.acc_scale = key_ptr(e.acc_scale, bt, key);
    e.out_off = key_off(bt, key);
    if (e.flags != nullptr) e.flags += key;
    return e;
}

// Split-fp16 image of a row-major fp32 matrix with rows of `len` values (len % 8 == 0), the operand format of gemm128h_kernel:
// every value x is the pair  h = fp16(64 x),  l = fp16(64 x - h)  (22 significand bits; 64: keeps the low parts of entries
// around 1 / sqrt(r), what a scaled polar iterate holds, out of fp16's subnormal range).  Row m lies at 2 len halves from row
// m - 1; its k-values go in slots of 8: halves [16 (k >> 3), +8) hold the high parts of k .. k + 7, the next 8 the low parts --
// a lane's MFMA fragment (8 consecutive k of one row, high and low) is 32 contiguous bytes.
constexpr float H_SCALE = 64.0f, H_UNSCALE = 1.0f / (64.0f * 64.0f);
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ int64_t split_at(int64_t row, int64_t k, int64_t len) { return row * 2 * len + ((k >> 3) << 4) + (k & 7); }
__device__ __forceinline__ void split_store1(_Float16 *img, int64_t at, float v) {
    const float sc = v * H_SCALE;
    const _Float16 h = (_Float16)sc;
    img[at] = h; img[at + 8] = (_Float16)(sc - (float)h);
}
__device__ __forceinline__ float split_load1(const _Float16 *img, int64_t at) { return ((float)img[at] + (float)img[at + 8]) * (1.0f / H_SCALE); }
// four consecutive k (k % 4 == 0) of one row
__device__ __forceinline__ void split_store4(_Float16 *img, int64_t at, const f32x4 &v) {
    const f32x4 sc = v * H_SCALE;
    const f16x4 h = __builtin_convertvector(sc, f16x4);
    *reinterpret_cast<f16x4 *>(img + at) = h;
    *reinterpret_cast<f16x4 *>(img + at + 8) = __builtin_convertvector(sc - __builtin_convertvector(h, f32x4), f16x4);
}

__device__ __forceinline__ float gemm_store(const GemmEpilogue &e, float *c, int64_t ldc, int64_t row, int64_t col, float v, bool &bad) {
    const int64_t idx = row * ldc + col;
    if (e.axpy != nullptr) v = __builtin_fmaf(e.axpy_a, e.axpy[idx], e.axpy_b * v);
    if (e.minus != nullptr) v = v - e.minus[idx];
    if (e.plus != nullptr) v = e.plus[idx] + v;
    if (e.mix != nullptr) v = __builtin_fmaf(e.mix_w0, e.mix[idx], e.mix_w1 * v);
    if (e.col_bias != nullptr) v = v + e.col_bias[col];
    if (e.out != nullptr) {
        bad |= nonfinite_f32(v);
        if (e.out_dt == MECHA_BF16) store1<MECHA_BF16>(e.out, e.out_off + idx, v);
        else if (e.out_dt == MECHA_F16) store1<MECHA_F16>(e.out, e.out_off + idx, v);
        else store1<MECHA_F32>(e.out, e.out_off + idx, v);
    } else c[idx] = v;
    return v;
}

// One operand tile (GK k-values x 64 m-values) of the block: 8 floats per thread, loaded with 16-byte accesses along whichever
// of the two directions is contiguous in memory, and stored to LDS as [k][m] (what the MFMA lanes read: one float each,
// 32 consecutive m per half-wave).
constexpr int GK = 32;                 // k-chunk per LDS buffer
constexpr int GP = 68;                 // LDS row pitch in floats (272 B: 16-byte aligned rows, rows 4 banks apart)
struct TileRegs { f32x4 v[2]; };

__device__ __forceinline__ TileRegs tile_load(const Operand &op, int64_t m0, int64_t M, int64_t k0, int64_t k_end) {
    TileRegs t;
    const int tid = threadIdx.x;
    if (op.ms == 1) {                  // m-contiguous: thread = (k = tid / 16 [+16], 4 consecutive m)
        const int64_t m = m0 + (tid & 15) * 4;
        const bool vec = ((op.ks & 3) == 0) && (m + 4 <= M) && ((reinterpret_cast<uintptr_t>(op.p) & 15u) == 0);
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int64_t k = k0 + (tid >> 4) + 16 * h;
            if (k < k_end && vec) t.v[h] = *reinterpret_cast<const f32x4 *>(op.p + k * op.ks + m);
            else {
#pragma unroll
                for (int j = 0; j < 4; j++) t.v[h][j] = (k < k_end && m + j < M) ? op.p[k * op.ks + m + j] : 0.0f;
            }
        }
    } else if (op.ks == 1) {           // k-contiguous: thread = (m = tid / 8 [+32], 4 consecutive k at 4 * (tid % 8)): 8 lanes
        // cover the 128 contiguous bytes of one row, a wave instruction touches 8 rows (one lane per row: 64 rows, 16 B of each)
        const int64_t k = k0 + 4 * (tid & 7);
        const bool vec = ((op.ms & 3) == 0) && ((k0 & 3) == 0) && ((reinterpret_cast<uintptr_t>(op.p) & 15u) == 0);
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int64_t m = m0 + (tid >> 3) + 32 * h;
            if (m < M && k + 4 <= k_end && vec) t.v[h] = *reinterpret_cast<const f32x4 *>(op.p + m * op.ms + k);
            else {
#pragma unroll
                for (int j = 0; j < 4; j++) t.v[h][j] = (m < M && k + j < k_end) ? op.p[m * op.ms + k + j] : 0.0f;
            }
        }
    } else {                           // neither: element (k, m) = (e / 64, e % 64), e = tid + 256 j
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int e = tid + 256 * j;
            const int64_t k = k0 + (e >> 6), m = m0 + (e & 63);
            t.v[j >> 2][j & 3] = (k < k_end && m < M) ? op.p[k * op.ks + m * op.ms] : 0.0f;
        }
    }
    return t;
}
__device__ __forceinline__ void tile_store(const Operand &op, const TileRegs &t, float *lds) {
    const int tid = threadIdx.x;
    if (op.ms == 1) {
#pragma unroll
        for (int h = 0; h < 2; h++) *reinterpret_cast<f32x4 *>(&lds[((tid >> 4) + 16 * h) * GP + (tid & 15) * 4]) = t.v[h];
    } else if (op.ks == 1) {
#pragma unroll
        for (int h = 0; h < 2; h++)
#pragma unroll
            for (int j = 0; j < 4; j++) lds[(4 * (tid & 7) + j) * GP + (tid >> 3) + 32 * h] = t.v[h][j];
    } else {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int e = tid + 256 * j;
            lds[(e >> 6) * GP + (e & 63)] = t.v[j >> 2][j & 3];
        }
    }
}

__global__ __launch_bounds__(256) void gemm_kernel(Operand A, Operand B, float *C, int64_t M, int64_t N, int64_t K, int64_t k_per_slab, int n_slabs, Batch bt,
                                                   GemmEpilogue e) {
    __shared__ __attribute__((aligned(16))) float lds[2][2][GK * GP];       // [buffer][A | B]
    const int key = blockIdx.z / n_slabs, slab = blockIdx.z - key * n_slabs;
    if (bt.skip != nullptr && bt.skip[key] != 0u) return;
    A.p = key_ptr(A.p, bt, key); B.p = key_ptr(B.p, bt, key); C = key_ptr(C, bt, key);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 31, kk = lane >> 5;
    const int64_t bm0 = (int64_t)blockIdx.y * 64, bn0 = (int64_t)blockIdx.x * 64;
    const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
    const int64_t m0 = bm0 + wm, n0 = bn0 + wn;
    const int64_t k_begin = (int64_t)slab * k_per_slab;
    const int64_t k_end = (k_begin + k_per_slab < K) ? k_begin + k_per_slab : K;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0.0f;
    // software pipeline: the next chunk's global loads are in flight while the MFMAs consume the current LDS buffer
    TileRegs ra = tile_load(A, bm0, M, k_begin, k_end), rb = tile_load(B, bn0, N, k_begin, k_end);
    tile_store(A, ra, lds[0][0]);
    tile_store(B, rb, lds[0][1]);
    __syncthreads();
    int buf = 0;
    for (int64_t k = k_begin; k < k_end; k += GK) {
        const bool more = k + GK < k_end;
        if (more) { ra = tile_load(A, bm0, M, k + GK, k_end); rb = tile_load(B, bn0, N, k + GK, k_end); }
        const float *la = lds[buf][0] + wm + i, *lb = lds[buf][1] + wn + i;
#pragma unroll
        for (int u = 0; u < GK / 2; u++)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(la[(2 * u + kk) * GP], lb[(2 * u + kk) * GP], acc, 0, 0, 0);
        if (more) { tile_store(A, ra, lds[buf ^ 1][0]); tile_store(B, rb, lds[buf ^ 1][1]); }
        __syncthreads();
        buf ^= 1;
    }
    float *c = C + (int64_t)slab * M * N;
    GemmEpilogue ee = epilogue_for_key(e, bt, key);
    if (n_slabs > 1) ee = GemmEpilogue{};                       // the epilogue then runs in gemm_reduce
    bool bad = false;
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const int64_t row = m0 + (r & 3) + 8 * (r >> 2) + 4 * kk, col = n0 + i;
        if (row < M && col < N) gemm_store(ee, c, N, row, col, acc[r], bad);
    }
    if (bad && ee.flags != nullptr) atomicOr(ee.flags, MECHA_FLAG_OUTPUT_NONFINITE);
}

// The same product for outputs of at least 256 x 256: block tile 128 x 128, wave tile 64 x 64 = 2 x 2 accumulators.  Every LDS
// read feeds two MFMAs and a wave has four independent accumulator chains, so one or two blocks per CU keep the matrix
// pipes fed (the 64 x 64 kernel: one dependent chain per wave, 400 small blocks for a 1280 x 1280 product).  16-deep
// k-chunks: 2 operands x 2 buffers x 16 x 132 floats = 33 KiB of LDS.
#ifndef MECHA_ROT_XCD_MAP
#define MECHA_ROT_XCD_MAP 1
#endif
constexpr int GK2 = 16;
constexpr int GP2 = 132;               // LDS row pitch in floats (528 B: 16-byte aligned rows, rows 4 banks apart)

enum : int { LAYOUT_M = 0, LAYOUT_K = 1, LAYOUT_ANY = 2 };     // which direction of an operand is contiguous in memory (compiled in)
template <int LAYOUT>
__device__ __forceinline__ TileRegs tile_load128(const Operand &op, int64_t m0, int64_t M, int64_t k0, int64_t k_end) {
    TileRegs t;
    const int tid = threadIdx.x;
    if constexpr (LAYOUT == LAYOUT_M) { // m-contiguous: thread = (k = tid / 16, 4 consecutive m at 4 * (tid % 16) [+64])
        const int64_t k = k0 + (tid >> 4);
        const bool vec_ok = ((op.ks & 3) == 0) && ((reinterpret_cast<uintptr_t>(op.p) & 15u) == 0);
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int64_t m = m0 + (tid & 15) * 4 + 64 * h;
            if (k < k_end && vec_ok && m + 4 <= M) t.v[h] = *reinterpret_cast<const f32x4 *>(op.p + k * op.ks + m);
            else {
#pragma unroll
                for (int j = 0; j < 4; j++) t.v[h][j] = (k < k_end && m + j < M) ? op.p[k * op.ks + m + j] : 0.0f;
            }
        }
    } else if constexpr (LAYOUT == LAYOUT_K) {   // k-contiguous: thread = (m = tid / 4 [+64], 4 consecutive k at 4 * (tid % 4)): 4 lanes per row
        const int64_t k = k0 + 4 * (tid & 3);
        const bool vec_ok = ((op.ms & 3) == 0) && ((k0 & 3) == 0) && ((reinterpret_cast<uintptr_t>(op.p) & 15u) == 0);
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int64_t m = m0 + (tid >> 2) + 64 * h;
            if (m < M && k + 4 <= k_end && vec_ok) t.v[h] = *reinterpret_cast<const f32x4 *>(op.p + m * op.ms + k);
            else {
#pragma unroll
                for (int j = 0; j < 4; j++) t.v[h][j] = (m < M && k + j < k_end) ? op.p[m * op.ms + k + j] : 0.0f;
            }
        }
    } else {                           // neither: element (k, m) = (e / 128, e % 128), e = tid + 256 j
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int e = tid + 256 * j;
            const int64_t k = k0 + (e >> 7), m = m0 + (e & 127);
            t.v[j >> 2][j & 3] = (k < k_end && m < M) ? op.p[k * op.ks + m * op.ms] : 0.0f;
        }
    }
    return t;
}
template <int LAYOUT>
__device__ __forceinline__ void tile_store128(const TileRegs &t, float *lds) {
    const int tid = threadIdx.x;
    if constexpr (LAYOUT == LAYOUT_M) {
#pragma unroll
        for (int h = 0; h < 2; h++) *reinterpret_cast<f32x4 *>(&lds[(tid >> 4) * GP2 + (tid & 15) * 4 + 64 * h]) = t.v[h];
    } else if constexpr (LAYOUT == LAYOUT_K) {
#pragma unroll
        for (int h = 0; h < 2; h++)
#pragma unroll
            for (int j = 0; j < 4; j++) lds[(4 * (tid & 3) + j) * GP2 + (tid >> 2) + 64 * h] = t.v[h][j];
    } else {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int e = tid + 256 * j;
            lds[(e >> 7) * GP2 + (e & 127)] = t.v[j >> 2][j & 3];
        }
    }
}

// The 128 x 128 block's epilogue: accumulator register r of lane (i, kk) is C[32 tm + (r & 3) + 8 (r >> 2) + 4 kk][32 tn + i] of the wave's
// 64 x 64 tile, so a lane holds 4 consecutive rows of one column per register group: whatever is stored transposed (the mirror
// tile of a symmetric product, an image of the transpose) goes out as 16 (fp32) or 2 x 8 (split image) contiguous bytes per lane.
// symmetric: the epilogue is at most the axpy of a symmetric matrix; mirror: the tile lies above the diagonal.
__device__ __forceinline__ void gemm128_finish(f32x16 (&acc)[2][2], float *c, const GemmEpilogue &ee, int64_t M, int64_t N, int64_t bm0, int64_t bn0,
                                               int wm, int wn, int i, int kk, bool symmetric, bool mirror) {
    bool bad = false;
    if (symmetric) {
        const bool vec = (N & 3) == 0;
#pragma unroll
        for (int tm = 0; tm < 2; tm++)
#pragma unroll
            for (int tn = 0; tn < 2; tn++)
#pragma unroll
                for (int g4 = 0; g4 < 4; g4++) {
                    const int64_t row0 = bm0 + wm + 32 * tm + 8 * g4 + 4 * kk, col = bn0 + wn + 32 * tn + i;
                    f32x4 v4;
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        float v = acc[tm][tn][4 * g4 + j];
                        const int64_t row = row0 + j;
                        if (row < M && col < N) {
                            if (ee.axpy != nullptr) v = __builtin_fmaf(ee.axpy_a, ee.axpy[row * N + col], ee.axpy_b * v);
                            c[row * N + col] = v;
                        }
                        v4[j] = v;
                    }
                    if (mirror && col < N) {
                        if (vec && row0 + 4 <= M) *reinterpret_cast<f32x4 *>(c + col * N + row0) = v4;
                        else {
#pragma unroll
                            for (int j = 0; j < 4; j++)
                                if (row0 + j < M) c[col * N + row0 + j] = v4[j];
                        }
                    }
                }
        return;
    }
#pragma unroll
    for (int tm = 0; tm < 2; tm++)
#pragma unroll
        for (int tn = 0; tn < 2; tn++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int64_t row = bm0 + wm + 32 * tm + (r & 3) + 8 * (r >> 2) + 4 * kk, col = bn0 + wn + 32 * tn + i;
                if (row < M && col < N) gemm_store(ee, c, N, row, col, acc[tm][tn][r], bad);
            }
    if (bad && ee.flags != nullptr) atomicOr(ee.flags, MECHA_FLAG_OUTPUT_NONFINITE);
}
// gemm128h_kernel's: v = axpy_a * axpy_img + axpy_b * acc (+ col_bias); then `out` in its dtype with the finite flag, or fp32 C if
// given; the image `sp` (with the mirror tile of a symmetric product) and the transpose's image `spt`.  M % 4 == 0 (images have
// rows of whole slots).
__device__ __forceinline__ void gemm128h_finish(f32x16 (&acc)[2][2], float *c, const GemmEpilogue &ee, int64_t M, int64_t N, int64_t bm0, int64_t bn0,
                                                int wm, int wn, int i, int kk, bool symmetric, bool mirror) {
    _Float16 *tsp = symmetric ? (mirror ? ee.sp : nullptr) : ee.spt;
    float *tc = (symmetric && mirror) ? c : nullptr;
    const float scale = ee.acc_scale != nullptr ? *ee.acc_scale : 1.0f;
    bool bad = false;
#pragma unroll
    for (int tm = 0; tm < 2; tm++)
#pragma unroll
        for (int tn = 0; tn < 2; tn++) {
            const int64_t col = bn0 + wn + 32 * tn + i;
            const float bias = (ee.col_bias != nullptr && col < N) ? ee.col_bias[col] : 0.0f;
#pragma unroll
            for (int g4 = 0; g4 < 4; g4++) {
                const int64_t row0 = bm0 + wm + 32 * tm + 8 * g4 + 4 * kk;
                if (row0 >= M || col >= N) continue;            // (whole groups: M % 4 == 0)
                f32x4 v4;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    float v = acc[tm][tn][4 * g4 + j];
                    const int64_t at = split_at(row0 + j, col, N);
                    if (ee.axpy_img != nullptr) v = __builtin_fmaf(ee.axpy_a, split_load1(ee.axpy_img, at), ee.axpy_b * v);
                    v = v * scale + bias;
                    if (ee.out != nullptr) {
                        bad |= nonfinite_f32(v);
                        const int64_t idx = ee.out_off + (row0 + j) * N + col;
                        if (ee.out_dt == MECHA_BF16) store1<MECHA_BF16>(ee.out, idx, v);
                        else if (ee.out_dt == MECHA_F16) store1<MECHA_F16>(ee.out, idx, v);
                        else store1<MECHA_F32>(ee.out, idx, v);
                    } else if (c != nullptr) c[(row0 + j) * N + col] = v;
                    if (ee.sp != nullptr) split_store1(ee.sp, at, v);
                    v4[j] = v;
                }
                if (tsp != nullptr) split_store4(tsp, split_at(col, row0, M), v4);
                if (tc != nullptr) *reinterpret_cast<f32x4 *>(tc + col * M + row0) = v4;
            }
        }
    if (bad && ee.flags != nullptr) atomicOr(ee.flags, MECHA_FLAG_OUTPUT_NONFINITE);
}

// The same epilogue staged through LDS (the k-loop's buffers are free by then), so that every global access is 16 bytes of one
// row: written straight from the accumulator layout (a lane = one column) the images took two 2-byte stores and the axpy term
// two 2-byte loads per ELEMENT -- as many memory instructions as the whole k-loop issued.  Two halves of 64 rows (33 KiB each):
// the two waves that own the half store their accumulators as fp32 [row][132]; then every thread takes groups of 8 consecutive
// columns of a row (axpy term: one 32-byte slot of the image; results: `out` / C / `sp` in whole 16-byte pieces) and writes the
// finished values back; then, for what goes out transposed (mirror tile, `spt`), groups of 8 consecutive rows of a column.
// M % 8 == 0 and N % 8 == 0 (images), so a group is inside the matrix or outside it as a whole.
constexpr int HF_PITCH = 132;
__device__ __forceinline__ void gemm128h_finish_lds(f32x16 (&acc)[2][2], float *tile /* 64 x HF_PITCH floats of LDS */, float *c, const GemmEpilogue &ee,
                                                    int64_t M, int64_t N, int64_t bm0, int64_t bn0, int wm, int wn, int i, int kk, bool symmetric, bool mirror) {
    _Float16 *tsp = symmetric ? (mirror ? ee.sp : nullptr) : ee.spt;
    float *tc = (symmetric && mirror) ? c : nullptr;
    const float scale = ee.acc_scale != nullptr ? *ee.acc_scale : 1.0f;
    const int tid = threadIdx.x;
    bool bad = false;
    for (int half = 0; half < 2; half++) {
        __syncthreads();                                        // the k-loop's last LDS reads / the previous half's transposed reads are done
        if (wm == 64 * half) {
#pragma unroll
            for (int tm = 0; tm < 2; tm++)
#pragma unroll
                for (int tn = 0; tn < 2; tn++)
#pragma unroll
                    for (int r = 0; r < 16; r++)
                        tile[(32 * tm + (r & 3) + 8 * (r >> 2) + 4 * kk) * HF_PITCH + wn + 32 * tn + i] = acc[tm][tn][r];
        }
        __syncthreads();
        const int64_t row_base = bm0 + 64 * half;
#pragma unroll
        for (int j = 0; j < 4; j++) {                           // 64 rows x 16 column groups = 1024 groups of 8
            const int gidx = tid + 256 * j, lr = gidx >> 4, lc = (gidx & 15) * 8;
            const int64_t row = row_base + lr, col = bn0 + lc;
            if (row >= M || col >= N) continue;
            float *tp = tile + lr * HF_PITCH + lc;
            f32x4 v0 = *reinterpret_cast<const f32x4 *>(tp), v1 = *reinterpret_cast<const f32x4 *>(tp + 4);
            const int64_t at = split_at(row, col, N);
            if (ee.axpy_img != nullptr) {
                const f16x8 xh = *reinterpret_cast<const f16x8 *>(ee.axpy_img + at), xl = *reinterpret_cast<const f16x8 *>(ee.axpy_img + at + 8);
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    v0[q] = __builtin_fmaf(ee.axpy_a, ((float)xh[q] + (float)xl[q]) * (1.0f / H_SCALE), ee.axpy_b * v0[q]);
                    v1[q] = __builtin_fmaf(ee.axpy_a, ((float)xh[q + 4] + (float)xl[q + 4]) * (1.0f / H_SCALE), ee.axpy_b * v1[q]);
                }
            }
            f32x4 b0 = f32x4{0.0f, 0.0f, 0.0f, 0.0f}, b1 = b0;
            if (ee.col_bias != nullptr) { b0 = *reinterpret_cast<const f32x4 *>(ee.col_bias + col); b1 = *reinterpret_cast<const f32x4 *>(ee.col_bias + col + 4); }
            v0 = v0 * scale + b0; v1 = v1 * scale + b1;
            const float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
            if (ee.out != nullptr) {
#pragma unroll
                for (int q = 0; q < 8; q++) bad |= nonfinite_f32(v[q]);
                const int64_t idx = ee.out_off + row * N + col;
                const bool vec = (idx & 7) == 0 && (reinterpret_cast<uintptr_t>(ee.out) & 15u) == 0;   // (arena offsets are multiples of 128 elements; a lone key's pointer may not be aligned)
                if (ee.out_dt == MECHA_BF16) { if (vec) store8<MECHA_BF16, true>(ee.out, idx, v); else for (int q = 0; q < 8; q++) store1<MECHA_BF16>(ee.out, idx + q, v[q]); }
                else if (ee.out_dt == MECHA_F16) { if (vec) store8<MECHA_F16, true>(ee.out, idx, v); else for (int q = 0; q < 8; q++) store1<MECHA_F16>(ee.out, idx + q, v[q]); }
                else { if (vec) store8<MECHA_F32, true>(ee.out, idx, v); else for (int q = 0; q < 8; q++) store1<MECHA_F32>(ee.out, idx + q, v[q]); }
            } else if (c != nullptr) {
                *reinterpret_cast<f32x4 *>(c + row * N + col) = v0; *reinterpret_cast<f32x4 *>(c + row * N + col + 4) = v1;
            }
            if (ee.sp != nullptr) { split_store4(ee.sp, at, v0); split_store4(ee.sp, at + 4, v1); }
            if (tsp != nullptr || tc != nullptr) { *reinterpret_cast<f32x4 *>(tp) = v0; *reinterpret_cast<f32x4 *>(tp + 4) = v1; }
        }
        if (tsp == nullptr && tc == nullptr) continue;          // (block-uniform)
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; j++) {                           // 128 columns x 8 row groups of 8
            const int gidx = tid + 256 * j, lc = gidx & 127, lr = (gidx >> 7) * 8;
            const int64_t row = row_base + lr, col = bn0 + lc;
            if (row >= M || col >= N) continue;
            f32x4 v0, v1;
#pragma unroll
            for (int q = 0; q < 4; q++) { v0[q] = tile[(lr + q) * HF_PITCH + lc]; v1[q] = tile[(lr + 4 + q) * HF_PITCH + lc]; }
            if (tsp != nullptr) { const int64_t at = split_at(col, row, M); split_store4(tsp, at, v0); split_store4(tsp, at + 4, v1); }
            if (tc != nullptr) { *reinterpret_cast<f32x4 *>(tc + col * M + row) = v0; *reinterpret_cast<f32x4 *>(tc + col * M + row + 4) = v1; }
        }
    }
    if (bad && ee.flags != nullptr) atomicOr(ee.flags, MECHA_FLAG_OUTPUT_NONFINITE);
}

// symmetric: A and B are the same operand (C = X X^T, or G G of a symmetric G) and the epilogue is at most the axpy of a
// symmetric matrix.  Only the tiles on and above the diagonal are computed; a tile above it is also stored transposed
// (every lane holds 4 consecutive rows of one column = 16 contiguous bytes of the transposed tile).  The result is symmetric
// bit for bit, as the full product was (the same products in the same order on both sides of the diagonal).
template <int LA, int LB>
__global__ __launch_bounds__(256, 3) void gemm128_kernel(Operand A, Operand B, float *C, int64_t M, int64_t N, int64_t K, int64_t k_per_slab, int n_slabs, Batch bt,
                                                      GemmEpilogue e, int symmetric) {
    __shared__ __attribute__((aligned(16))) float lds[2][2][GK2 * GP2];     // [buffer][A | B]
    // Workgroups go to the 8 XCDs round-robin by their linear id, each XCD with an L2 of its own: XCD x takes a CONTIGUOUS
    // range of the launch's tiles (x fastest, then y, then key and slab), so that the tiles of one key -- which read the
    // same operand panels at about the same time -- share one L2 instead of pulling every panel into all eight.
    unsigned bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (MECHA_ROT_XCD_MAP) {
        const uint64_t gx = gridDim.x, gy = gridDim.y, total = gx * gy * gridDim.z;
        const uint64_t id = bx + gx * (by + gy * (uint64_t)bz), xcd = id & 7u;
        const uint64_t logical = xcd * (total >> 3) + (xcd < (total & 7u) ? xcd : (total & 7u)) + (id >> 3);
        bx = (unsigned)(logical % gx); by = (unsigned)((logical / gx) % gy); bz = (unsigned)(logical / (gx * gy));
    }
    const int key = bz / n_slabs, slab = bz - key * n_slabs;
    if (symmetric && bx < by) return;
    if (bt.skip != nullptr && bt.skip[key] != 0u) return;
    A.p = key_ptr(A.p, bt, key); B.p = key_ptr(B.p, bt, key); C = key_ptr(C, bt, key);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 31, kk = lane >> 5;
    const int64_t bm0 = (int64_t)by * 128, bn0 = (int64_t)bx * 128;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    const int64_t k_begin = (int64_t)slab * k_per_slab;
    const int64_t k_end = (k_begin + k_per_slab < K) ? k_begin + k_per_slab : K;
    f32x16 acc[2][2];
#pragma unroll
    for (int tm = 0; tm < 2; tm++)
#pragma unroll
        for (int tn = 0; tn < 2; tn++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[tm][tn][r] = 0.0f;
    // software pipeline: the next chunk's global loads are in flight while the MFMAs consume the current LDS buffer.  (Two
    // chunks of register prefetch were tried: 221 VGPRs, one block per CU, 74 -> 112 us per launch on the SDXL mix.  The kernel
    // is bound by the latency of the operand loads; more blocks per CU hide it better than a deeper pipeline in registers.)
    TileRegs ra = tile_load128<LA>(A, bm0, M, k_begin, k_end), rb = tile_load128<LB>(B, bn0, N, k_begin, k_end);
    tile_store128<LA>(ra, lds[0][0]);
    tile_store128<LB>(rb, lds[0][1]);
    __syncthreads();
    int buf = 0;
    for (int64_t k = k_begin; k < k_end; k += GK2) {
        const bool more = k + GK2 < k_end;
        if (more) { ra = tile_load128<LA>(A, bm0, M, k + GK2, k_end); rb = tile_load128<LB>(B, bn0, N, k + GK2, k_end); }
        const float *la = lds[buf][0] + wm + i, *lb = lds[buf][1] + wn + i;
#pragma unroll
        for (int u = 0; u < GK2 / 2; u++) {
            const float a0 = la[(2 * u + kk) * GP2], a1 = la[(2 * u + kk) * GP2 + 32];
            const float b0 = lb[(2 * u + kk) * GP2], b1 = lb[(2 * u + kk) * GP2 + 32];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        if (more) { tile_store128<LA>(ra, lds[buf ^ 1][0]); tile_store128<LB>(rb, lds[buf ^ 1][1]); }
        __syncthreads();
        buf ^= 1;
    }
    float *c = C + (int64_t)slab * M * N;
    GemmEpilogue ee = epilogue_for_key(e, bt, key);
    if (n_slabs > 1) ee = GemmEpilogue{};                       // the epilogue then runs in gemm_reduce
    gemm128_finish(acc, c, ee, M, N, bm0, bn0, wm, wn, i, kk, symmetric != 0, bx > by);
}

// ------------------------------------------------------------------------------------------------ split-fp16 GEMM
// The same 128 x 128 product on v_mfma_f32_32x32x16_f16 (16x the fp32 MFMA's rate) from operands kept as split-fp16 images
// (split_at above): the product is the three MFMAs  h_a h_b + h_a l_b + l_a h_b  accumulated in fp32 (the dropped l_a l_b term
// is 2^-22 of the product), scaled back by 2^-12 at the end, exactly.  So a result carries ~5e-7 relative error per product term
// against 6e-8 for the fp32 instruction: used for the polar steps that later fp32 steps correct (polar_newton_schulz), never for a
// step whose output is final.  Operands are bounded there (|x| <= ~6 for every matrix of a scaled iteration), far from fp16's
// 65504 / 64.  The images are made once, by the epilogue of the product that made the matrix (GemmEpilogue::sp / spt) or by
// rot_split_kernel -- converting in this kernel's loop, the first version, cost as many VALU cycles as the MFMAs took
// (profiles/r03_rotate_notes.md), once per consuming block instead of once per value.
// Both operands are k-contiguous by construction (rows of an image).  LDS holds a chunk as two planes [row][k] (high parts, low
// parts): a lane's fragment is one ds_read_b128 from each.  The 16-byte slots of a row are XOR-swizzled with row bits so that
// the 16 lanes ds_read_b128 serves per LDS cycle (MI355X_MICROARCH.md, LDS: lanes {0-3, 12-15, 20-27}, ...) touch 16 different slots.
// HK: k-chunk per LDS buffer, 16 or 32 (a plane row is then 32 or 64 bytes; 32 or 64 KiB of LDS per block).
struct HOperand { const _Float16 *p; int64_t ms; };            // image base, halves between rows (2 x row length)
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));

// offset (in halves) of 16-byte slot `slot` (8 k-values) of row `row` in a plane (128 x HK halves)
// The low plane keeps row r where the high plane keeps row r ^ H_LO_FLIP (the row 64 bytes away): the 8 lanes one ds_write_b128
// cycle serves hold the high AND low parts of the same rows, and both planes start at the same bank.
template <int HK> constexpr int H_LO_FLIP = 32 / HK;
template <int HK> __device__ __forceinline__ int h_slot(int row, int slot) {
    const int sw = HK == 16 ? ((row >> 3) & 1) : ((row >> 2) & 3);
    return row * HK + ((slot ^ sw) << 3);
}
// One operand chunk (128 rows x HK k-values, high and low parts: 128 x 4 HK bytes) is 512 HK / 16 pieces of 16 bytes; thread tid
// takes pieces tid, tid + 256, ...: piece e = (row e / LPR, 16-byte unit e % LPR of the row's 4 HK contiguous bytes), unit u =
// (slot u >> 1, high | low part u & 1).  A wave instruction so reads whole rows' contiguous bytes (HK = 32: 8 full 128-byte lines).
template <int HK> struct HTileRegs { u32x4_t v[HK / 8]; };
template <int HK> __device__ __forceinline__ HTileRegs<HK> h_tile_load(const HOperand &op, int64_t m0, int64_t M, int64_t k0, int64_t k_end) {
    constexpr int LPR = HK / 4, RPP = 256 / LPR;
    HTileRegs<HK> t;
    const int tid = threadIdx.x, u = tid % LPR;
    const int64_t k = k0 + 8 * (u >> 1);
#pragma unroll
    for (int h = 0; h < HK / 8; h++) {
        const int64_t m = m0 + tid / LPR + RPP * h;
        if (m < M && k < k_end) t.v[h] = *reinterpret_cast<const u32x4_t *>(op.p + m * op.ms + 2 * k + 8 * (u & 1));   // (k_end % 8 == 0: whole slots)
        else t.v[h] = u32x4_t{0u, 0u, 0u, 0u};
    }
    return t;
}
template <int HK> __device__ __forceinline__ void h_tile_store(const HTileRegs<HK> &t, _Float16 *lds) {
    constexpr int LPR = HK / 4, RPP = 256 / LPR, H_PLANE = 128 * HK;
    const int tid = threadIdx.x, u = tid % LPR;
#pragma unroll
    for (int h = 0; h < HK / 8; h++)
        *reinterpret_cast<u32x4_t *>(lds + (u & 1) * H_PLANE + h_slot<HK>((tid / LPR + RPP * h) ^ ((u & 1) * H_LO_FLIP<HK>), u >> 1)) = t.v[h];
}

// D: chunks of register prefetch.  The operand loads come from L2 or beyond at ~2 us under this kernel's own load (measured:
// throughput followed the bytes in flight per CU, whatever the chunk size, load shape or blocks per CU), so a block keeps D chunks
// in flight: at the top of chunk c's iteration ring slot c % D holds chunk c + 1 and the others chunks c + 2 .. c + D; after
// the MFMAs of chunk c, slot c % D goes to the other LDS buffer and is reloaded with chunk c + D + 1.
template <int HK, int D>
__global__ __launch_bounds__(256, (HK == 16 && D == 1) ? 3 : 2) void gemm128h_kernel(HOperand A, HOperand B, float *C, int64_t M, int64_t N, int64_t K, int64_t k_per_slab,
                                                                                   int n_slabs, Batch bt, GemmEpilogue e, int symmetric) {
    constexpr int H_PLANE = 128 * HK;
    __shared__ __attribute__((aligned(16))) _Float16 lds[2][2][2 * H_PLANE];   // [buffer][A | B][high plane, low plane]
    unsigned bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (MECHA_ROT_XCD_MAP) {                                    // as gemm128_kernel: an XCD takes a contiguous range of the tiles
        const uint64_t gx = gridDim.x, gy = gridDim.y, total = gx * gy * gridDim.z;
        const uint64_t id = bx + gx * (by + gy * (uint64_t)bz), xcd = id & 7u;
        const uint64_t logical = xcd * (total >> 3) + (xcd < (total & 7u) ? xcd : (total & 7u)) + (id >> 3);
        bx = (unsigned)(logical % gx); by = (unsigned)((logical / gx) % gy); bz = (unsigned)(logical / (gx * gy));
    }
    const int key = bz / n_slabs, slab = bz - key * n_slabs;
    if (symmetric && bx < by) return;
    if (bt.skip != nullptr && bt.skip[key] != 0u) return;
    A.p = key_ptr(A.p, bt, key); B.p = key_ptr(B.p, bt, key); C = key_ptr(C, bt, key);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 31, kk = lane >> 5;
    const int64_t bm0 = (int64_t)by * 128, bn0 = (int64_t)bx * 128;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    const int64_t k_begin = (int64_t)slab * k_per_slab;
    const int64_t k_end = (k_begin + k_per_slab < K) ? k_begin + k_per_slab : K;
    const int64_t n_chunks = (k_end - k_begin + HK - 1) / HK;
    f32x16 acc[2][2];
#pragma unroll
    for (int tm = 0; tm < 2; tm++)
#pragma unroll
        for (int tn = 0; tn < 2; tn++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[tm][tn][r] = 0.0f;
    HTileRegs<HK> ra[D], rb[D];
    {
        const HTileRegs<HK> a0 = h_tile_load<HK>(A, bm0, M, k_begin, k_end), b0 = h_tile_load<HK>(B, bn0, N, k_begin, k_end);
#pragma unroll
        for (int j = 0; j < D; j++) {                           // (chunks past the end load nothing: zeros)
            ra[j] = h_tile_load<HK>(A, bm0, M, k_begin + (j + 1) * HK, k_end);
            rb[j] = h_tile_load<HK>(B, bn0, N, k_begin + (j + 1) * HK, k_end);
        }
        h_tile_store<HK>(a0, lds[0][0]);
        h_tile_store<HK>(b0, lds[0][1]);
    }
    __syncthreads();
    for (int64_t c0 = 0; c0 < n_chunks; c0 += D) {
#pragma unroll
        for (int j = 0; j < D; j++) {
            const int64_t c = c0 + j;
            if (c >= n_chunks) break;
            const int buf = (int)(c & 1);
            const _Float16 *pa = lds[buf][0], *pb = lds[buf][1];
#pragma unroll
            for (int s = 0; s < HK / 16; s++) {
                f16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
                for (int t = 0; t < 2; t++) {
                    const int oa = h_slot<HK>(wm + 32 * t + i, 2 * s + kk), ob = h_slot<HK>(wn + 32 * t + i, 2 * s + kk);
                    const int oal = h_slot<HK>((wm + 32 * t + i) ^ H_LO_FLIP<HK>, 2 * s + kk), obl = h_slot<HK>((wn + 32 * t + i) ^ H_LO_FLIP<HK>, 2 * s + kk);
                    ah[t] = *reinterpret_cast<const f16x8 *>(pa + oa); al[t] = *reinterpret_cast<const f16x8 *>(pa + H_PLANE + oal);
                    bh[t] = *reinterpret_cast<const f16x8 *>(pb + ob); bl[t] = *reinterpret_cast<const f16x8 *>(pb + H_PLANE + obl);
                }
#pragma unroll
                for (int tm = 0; tm < 2; tm++)
#pragma unroll
                    for (int tn = 0; tn < 2; tn++) {
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[tm], bh[tn], acc[tm][tn], 0, 0, 0);
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[tm], bl[tn], acc[tm][tn], 0, 0, 0);
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[tm], bh[tn], acc[tm][tn], 0, 0, 0);
                    }
            }
            if (c + 1 < n_chunks) { h_tile_store<HK>(ra[j], lds[buf ^ 1][0]); h_tile_store<HK>(rb[j], lds[buf ^ 1][1]); }
            ra[j] = h_tile_load<HK>(A, bm0, M, k_begin + (c + 1 + D) * HK, k_end);
            rb[j] = h_tile_load<HK>(B, bn0, N, k_begin + (c + 1 + D) * HK, k_end);
            __syncthreads();
        }
    }
#pragma unroll
    for (int tm = 0; tm < 2; tm++)
#pragma unroll
        for (int tn = 0; tn < 2; tn++) acc[tm][tn] *= H_UNSCALE;
    float *c = (C == nullptr) ? nullptr : C + (int64_t)slab * M * N;
    GemmEpilogue ee = epilogue_for_key(e, bt, key);
    if (n_slabs > 1) ee = GemmEpilogue{};
    if constexpr (HK == 32) gemm128h_finish_lds(acc, reinterpret_cast<float *>(&lds[0][0][0]), c, ee, M, N, bm0, bn0, wm, wn, i, kk, symmetric != 0, bx > by);
    else gemm128h_finish(acc, c, ee, M, N, bm0, bn0, wm, wn, i, kk, symmetric != 0, bx > by);
}

// The factor a key's matrix is multiplied by on its way into an image (rot_split_kernel), from the key's scalars:
//   scal[2] = |X|_F^2 of the unscaled matrix  ->  t = 2^-ceil(log2 |X|_F): t X has every entry and singular value <= 1, exactly;
//   scal[0] = |G|_F^2 of G = (t X)(t X)^T      ->  s = |G|_F^(-1/2) <= 1 / sigma_max(t X), the polar iteration's scaling.
enum : int { SPLIT_PLAIN = 0, SPLIT_T = 1, SPLIT_TS = 2, SPLIT_SS = 3 };      // x,  t x,  t s x,  s^2 x
__device__ __forceinline__ float split_factor(int mode, const double *scal) {
    if (mode == SPLIT_PLAIN) return 1.0f;
    const double ss = scal[2], h = scal[0];
    const double t = (ss > 1e-150 && ss < 1e150) ? exp2(-ceil(0.5 * log2(ss))) : (ss == ss ? 0.0 : ss);
    if (mode == SPLIT_T) return (float)t;
    const double s = (h > 0.0 && h < 1e300) ? 1.0 / sqrt(sqrt(h)) : (h == h ? 0.0 : h);
    return mode == SPLIT_TS ? (float)(t * s) : (float)(s * s);
}
// sp = the split image of f x (x: r x c row-major fp32, f: split_factor), spt (nullable) = the image of its transpose; every key of
// the batch that `skip` does not excuse.  64 x 64 tiles through LDS.  r % 8 == 0 and c % 8 == 0.
__global__ __launch_bounds__(256) void rot_split_kernel(const float *x, _Float16 *sp, _Float16 *spt, int64_t r, int64_t c, int mode, const double *scal, Batch bt) {
    __shared__ __attribute__((aligned(16))) float tile[64][68];
    const int key = blockIdx.z;
    if (bt.skip != nullptr && bt.skip[key] != 0u) return;
    x = key_ptr(x, bt, key); sp = key_ptr(sp, bt, key); spt = key_ptr(spt, bt, key);
    const float f = split_factor(mode, key_ptr(scal, bt, key));
    const int64_t r0 = (int64_t)blockIdx.y * 64, c0 = (int64_t)blockIdx.x * 64;
    // groups of 8 consecutive values of a row (whole slots: 32 contiguous bytes of the image), two per thread
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const int gidx = threadIdx.x + 256 * j, lr = gidx >> 3, lc = (gidx & 7) * 8;
        if (r0 + lr < r && c0 + lc < c) {
            const float *src = x + (r0 + lr) * c + c0 + lc;
            const f32x4 v0 = *reinterpret_cast<const f32x4 *>(src) * f, v1 = *reinterpret_cast<const f32x4 *>(src + 4) * f;
            const int64_t at = split_at(r0 + lr, c0 + lc, c);
            split_store4(sp, at, v0); split_store4(sp, at + 4, v1);
            *reinterpret_cast<f32x4 *>(&tile[lr][lc]) = v0; *reinterpret_cast<f32x4 *>(&tile[lr][lc + 4]) = v1;
        }
    }
    if (spt == nullptr) return;
    __syncthreads();
    // the transpose: groups of 8 consecutive ROWS of one column
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const int gidx = threadIdx.x + 256 * j, lc = gidx & 63, lr = (gidx >> 6) * 8;
        if (r0 + lr < r && c0 + lc < c) {
            f32x4 v0, v1;
#pragma unroll
            for (int q = 0; q < 4; q++) { v0[q] = tile[lr + q][lc]; v1[q] = tile[lr + 4 + q][lc]; }
            const int64_t at = split_at(c0 + lc, r0 + lr, r);
            split_store4(spt, at, v0); split_store4(spt, at + 4, v1);
        }
    }
}
// scal[2] += sum x^2 (fp64; zeroed by the host)
__global__ __launch_bounds__(256) void rot_sumsq_kernel(const float *x, int64_t n, double *scal, Batch bt) {
    __shared__ double sh[4];
    const int key = blockIdx.y;
    if (bt.skip != nullptr && bt.skip[key] != 0u) return;
    x = key_ptr(x, bt, key); scal = key_ptr(scal, bt, key);
    double s0 = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) { const double v = (double)x[i]; s0 += v * v; }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s0 += __shfl_down(s0, off, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s0;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(&scal[2], sh[0] + sh[1] + sh[2] + sh[3]);
}

__global__ __launch_bounds__(256) void gemm_reduce_kernel(const float *slabs, int n_slabs, float *C, int64_t M, int64_t N, GemmEpilogue e, Batch bt) {
    const int key = blockIdx.y;
    if (bt.skip != nullptr && bt.skip[key] != 0u) return;
    slabs = key_ptr(slabs, bt, key); C = key_ptr(C, bt, key);
    e = epilogue_for_key(e, bt, key);
    const int64_t total = M * N;
    bool bad = false;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const int64_t row = t / N, col = t - row * N;
        float s = 0.0f;
        for (int z = 0; z < n_slabs; z++) s += slabs[(int64_t)z * total + t];
        if (e.h) {                                              // a product of gemm128h_kernel: gemm128h_finish's steps
            if (e.axpy_img != nullptr) s = __builtin_fmaf(e.axpy_a, split_load1(e.axpy_img, split_at(row, col, N)), e.axpy_b * s);
            if (e.acc_scale != nullptr) s *= *e.acc_scale;
            if (e.col_bias != nullptr) s += e.col_bias[col];
            if (e.out != nullptr) {
                bad |= nonfinite_f32(s);
                if (e.out_dt == MECHA_BF16) store1<MECHA_BF16>(e.out, e.out_off + t, s);
                else if (e.out_dt == MECHA_F16) store1<MECHA_F16>(e.out, e.out_off + t, s);
                else store1<MECHA_F32>(e.out, e.out_off + t, s);
            } else if (C != nullptr) C[t] = s;
            if (e.sp != nullptr) split_store1(e.sp, split_at(row, col, N), s);
            if (e.spt != nullptr) split_store1(e.spt, split_at(col, row, M), s);
        } else gemm_store(e, C, N, row, col, s, bad);
    }
    if (bad && e.flags != nullptr) atomicOr(e.flags, MECHA_FLAG_OUTPUT_NONFINITE);
}

// The centred matrices, times the key's powers of two t_a / t_b (ROT_SCAL_T: entries <= 2 whatever the weights' range; the polar
// factors do not see a scale, the application multiplies by 1 / t_a), straight as images (gemm128h_kernel's operands), 64 x 64
// tiles through LDS; n % 8 == 0, p % 8 == 0:
//   always  Ac (n x p: left operand of the application) and Ac^T (p x n);
//   tall keys: Bc^T (A^T B contracts over the rows);  wide keys: Bc (n x p) and Bc in fp32 (the first polar solve's input).
__global__ __launch_bounds__(256) void rot_center_img_kernel(const void *a, const void *b, int dt, int64_t n, int64_t p, const float *a_mean,
                                                             const float *b_mean, _Float16 *ac_img, _Float16 *act_img, _Float16 *bct_img, _Float16 *bc_img,
                                                             float *bc_f32, const double *scal, Batch bt) {
    __shared__ __attribute__((aligned(16))) float ta[64][68], tb[64][68];
    const int key = blockIdx.z;
    const int64_t base = key_off(bt, key);
    const float *ts = reinterpret_cast<const float *>(key_ptr(scal, bt, key) + ROT_SCAL_T);
    const float sa = ts[0], sb = ts[1];                         // powers of two: the products below are exact
    a_mean = key_ptr(a_mean, bt, key); b_mean = key_ptr(b_mean, bt, key);
    ac_img = key_ptr(ac_img, bt, key); act_img = key_ptr(act_img, bt, key); bct_img = key_ptr(bct_img, bt, key); bc_img = key_ptr(bc_img, bt, key);
    bc_f32 = key_ptr(bc_f32, bt, key);
    const int64_t r0 = (int64_t)blockIdx.y * 64, c0 = (int64_t)blockIdx.x * 64;
    // groups of 8 consecutive values of a row (16-byte loads of 16-bit inputs: the caller checked the alignment), two per thread
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const int gidx = threadIdx.x + 256 * j, lr = gidx >> 3, lc = (gidx & 7) * 8;
        if (r0 + lr < n && c0 + lc < p) {
            const int64_t at = (r0 + lr) * p + c0 + lc;
            float x[8], y[8];
            if (dt == MECHA_BF16) { load8<MECHA_BF16>(a, base + at, x); load8<MECHA_BF16>(b, base + at, y); }
            else if (dt == MECHA_F16) { load8<MECHA_F16>(a, base + at, x); load8<MECHA_F16>(b, base + at, y); }
            else { load8<MECHA_F32>(a, base + at, x); load8<MECHA_F32>(b, base + at, y); }
            f32x4 x0, x1, y0, y1;
#pragma unroll
            for (int q = 0; q < 4; q++) {                       // (svd.py:69-70), times t_a / t_b
                x0[q] = (x[q] - a_mean[c0 + lc + q]) * sa; x1[q] = (x[q + 4] - a_mean[c0 + lc + 4 + q]) * sa;
                y0[q] = (y[q] - b_mean[c0 + lc + q]) * sb; y1[q] = (y[q + 4] - b_mean[c0 + lc + 4 + q]) * sb;
            }
            const int64_t im = split_at(r0 + lr, c0 + lc, p);
            split_store4(ac_img, im, x0); split_store4(ac_img, im + 4, x1);
            if (bc_img != nullptr) { split_store4(bc_img, im, y0); split_store4(bc_img, im + 4, y1); }
            if (bc_f32 != nullptr) { *reinterpret_cast<f32x4 *>(bc_f32 + at) = y0; *reinterpret_cast<f32x4 *>(bc_f32 + at + 4) = y1; }
            *reinterpret_cast<f32x4 *>(&ta[lr][lc]) = x0; *reinterpret_cast<f32x4 *>(&ta[lr][lc + 4]) = x1;
            *reinterpret_cast<f32x4 *>(&tb[lr][lc]) = y0; *reinterpret_cast<f32x4 *>(&tb[lr][lc + 4]) = y1;
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 2; j++) {                               // the transposes: 8 consecutive rows of one column
        const int gidx = threadIdx.x + 256 * j, lc = gidx & 63, lr = (gidx >> 6) * 8;
        if (r0 + lr < n && c0 + lc < p) {
            f32x4 x0, x1, y0, y1;
#pragma unroll
            for (int q = 0; q < 4; q++) { x0[q] = ta[lr + q][lc]; x1[q] = ta[lr + 4 + q][lc]; y0[q] = tb[lr + q][lc]; y1[q] = tb[lr + 4 + q][lc]; }
            const int64_t at = split_at(c0 + lc, r0 + lr, n);
            split_store4(act_img, at, x0); split_store4(act_img, at + 4, x1);
            if (bct_img != nullptr) { split_store4(bct_img, at, y0); split_store4(bct_img, at + 4, y1); }
        }
    }
}
// img = the image of x - y (both n x p fp32), every key of the batch
__global__ __launch_bounds__(256) void rot_diff_img_kernel(const float *x, const float *y, _Float16 *img, int64_t n, int64_t p, Batch bt) {
    const int key = blockIdx.y;
    x = key_ptr(x, bt, key); y = key_ptr(y, bt, key); img = key_ptr(img, bt, key);
    const int64_t groups = n * p / 4;                           // four consecutive values of one row (p % 8 == 0)
    for (int64_t gi = (int64_t)blockIdx.x * 256 + threadIdx.x; gi < groups; gi += (int64_t)gridDim.x * 256) {
        const f32x4 d = *reinterpret_cast<const f32x4 *>(x + 4 * gi) - *reinterpret_cast<const f32x4 *>(y + 4 * gi);
        const int64_t row = 4 * gi / p, col = 4 * gi - row * p;
        split_store4(img, split_at(row, col, p), d);
    }
}

// ------------------------------------------------------------------------------------------------ polar factor
// scal[0] += sum G^2, scal[1] += sum (G - I)^2 over the r x r Gram matrix G (fp64 accumulation; scal zeroed by the host)
__global__ __launch_bounds__(256) void rot_gram_norms_kernel(const float *g, int64_t r, double *scal, Batch bt) {
    __shared__ double sh[2][4];
    const int key = blockIdx.y;
    if (bt.skip != nullptr && bt.skip[key] != 0u) return;
    g = key_ptr(g, bt, key); scal = key_ptr(scal, bt, key);
    const int64_t total = r * r;
    double s0 = 0.0, s1 = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const double v = (double)g[i];
        const double d = v - ((i / r == i % r) ? 1.0 : 0.0);
        s0 += v * v; s1 += d * d;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { s0 += __shfl_down(s0, off, 64); s1 += __shfl_down(s1, off, 64); }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { sh[0][wave] = s0; sh[1][wave] = s1; }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(&scal[0], sh[0][0] + sh[0][1] + sh[0][2] + sh[0][3]);
        atomicAdd(&scal[1], sh[1][0] + sh[1][1] + sh[1][2] + sh[1][3]);
    }
}

// torch.allclose(a.half(), b.half())  (svd.py:46): |a - b| <= 1e-8 + 1e-5 |b| on the fp16-rounded values
__global__ __launch_bounds__(256) void rot_allclose_kernel(const void *a, const void *b, int dt, int64_t n, unsigned int *not_close, Batch bt) {
    const int key = blockIdx.y;
    const int64_t base = key_off(bt, key);
    not_close += key;
    bool bad = false;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float x = (float)(_Float16)ld_any(dt, a, base + i), y = (float)(_Float16)ld_any(dt, b, base + i);
        const bool close = (x == y) || (__builtin_fabsf(x - y) <= 1e-8f + 1e-5f * __builtin_fabsf(y));
        bad |= !close;
    }
    if (bad) atomicOr(not_close, 1u);
}

// ------------------------------------------------------------------------------------------------ host
size_t up256(size_t v) { return (v + 255) / 256 * 256; }

constexpr size_t SLAB_FLOATS = (size_t)4096 * 64 * 64;          // most split-K slab floats one launch may use (all its keys together)
constexpr int GRID_Z_MAX = 65535;

// What a product needs beyond its operands: the key batch, the split-K slab area of key 0 and its size per key.
struct GemmCtx {
    Batch bt;
    float *slabs;
    size_t slab_floats;
    hipStream_t s;
};
const Batch ONE_KEY{1, 0, nullptr, nullptr};

void gemm(Operand A, Operand B, float *C, int64_t M, int64_t N, int64_t K, const GemmEpilogue &e, const GemmCtx &g, bool symmetric = false) {
    const int count = g.bt.count;
    const bool big = M >= 256 && N >= 256;
    const int64_t T = big ? 128 : 64;                           // block tile
    const int64_t tiles = ((M + T - 1) / T) * ((N + T - 1) / T) * count;
    // split K so that the launch fills the machine: about two resident 128 x 128 blocks per CU (512 on 256 CUs; 256 and 1024
    // measured slower), four 64 x 64 ones.  A batch of keys brings its own parallelism: the slabs shrink as the batch grows.
    int64_t z = big ? 512 / tiles : (1024 + tiles - 1) / tiles;
    const int64_t zmax = big ? (K + 127) / 128 : (K + 255) / 256;
    if (z > zmax) z = zmax;
    if (z > GRID_Z_MAX / count) z = GRID_Z_MAX / count;
    while (z > 1 && (size_t)z * (size_t)M * (size_t)N > g.slab_floats) z--;
    if (z < 1) z = 1;
    int64_t k_per = (K + z - 1) / z;
    if (big) k_per = (k_per + GK2 - 1) / GK2 * GK2;             // whole k-chunks per slab (keeps 16-byte loads along k aligned)
    else k_per = (k_per + 1) / 2 * 2;                           // the MFMA consumes k in pairs
    z = (K + k_per - 1) / k_per;
    dim3 grid((unsigned)((N + T - 1) / T), (unsigned)((M + T - 1) / T), (unsigned)(z * count)), block(256);
    float *dst = z == 1 ? C : g.slabs;
    const GemmEpilogue none{};
    // (symmetric products: the upper triangle only, when the epilogue runs in the kernel itself)
    const int sym = (symmetric && z == 1 && M == N && C != nullptr && e.out == nullptr && e.minus == nullptr && e.plus == nullptr && e.mix == nullptr &&
                     e.col_bias == nullptr) ? 1 : 0;
    if (big) {
        // the operand layouts are compiled in (a run-time choice cost 208 VGPRs and two waves per SIMD)
        const int la = A.ms == 1 ? LAYOUT_M : (A.ks == 1 ? LAYOUT_K : LAYOUT_ANY), lb = B.ms == 1 ? LAYOUT_M : (B.ks == 1 ? LAYOUT_K : LAYOUT_ANY);
#define MECHA_GEMM128(LA_, LB_) MECHA_LAUNCH(MK_ROT_GEMM, (gemm128_kernel<LA_, LB_>), grid, block, 0, g.s, A, B, dst, M, N, K, k_per, (int)z, g.bt, z == 1 ? e : none, sym)
        if (la == LAYOUT_M && lb == LAYOUT_M) MECHA_GEMM128(LAYOUT_M, LAYOUT_M);
        else if (la == LAYOUT_K && lb == LAYOUT_K) MECHA_GEMM128(LAYOUT_K, LAYOUT_K);
        else if (la == LAYOUT_K && lb == LAYOUT_M) MECHA_GEMM128(LAYOUT_K, LAYOUT_M);
        else if (la == LAYOUT_M && lb == LAYOUT_K) MECHA_GEMM128(LAYOUT_M, LAYOUT_K);
        else MECHA_GEMM128(LAYOUT_ANY, LAYOUT_ANY);
#undef MECHA_GEMM128
    }
    else MECHA_LAUNCH(MK_ROT_GEMM, gemm_kernel, grid, block, 0, g.s, A, B, dst, M, N, K, k_per, (int)z, g.bt, z == 1 ? e : none);
    if (z > 1) {
        const int64_t total = M * N;
        const int64_t want = (total + 255) / 256, cap = count >= 64 ? 16 : 1024 / count;
        MECHA_LAUNCH(MK_ROT_GEMM, gemm_reduce_kernel, dim3((unsigned)(want < cap ? want : cap), (unsigned)count), dim3(256), 0, g.s, g.slabs, (int)z, C, M, N, e, g.bt);
    }
}

// C[m][n] = sum_k a[m][k] b[n][k] from the split-fp16 images of a (M x K) and b (N x K); M, N >= 256, K % 8 == 0.  Same slab
// logic as gemm().  32-deep k-chunks, two in flight (measured best: 1.123 s against 1.149 s for MECHA_ROT_HK=16, one in flight,
// on the 1071 SDXL keys).
void gemm_h(const _Float16 *a, const _Float16 *b, float *C, int64_t M, int64_t N, int64_t K, const GemmEpilogue &e_in, const GemmCtx &g, bool symmetric = false) {
    GemmEpilogue e = e_in;
    e.h = 1;
    static const int hk = [] { const char *v = getenv("MECHA_ROT_HK"); return v != nullptr && atoi(v) == 16 ? 16 : 32; }();
    const int count = g.bt.count;
    const int64_t tiles = ((M + 127) / 128) * ((N + 127) / 128) * count;
    int64_t z = 512 / tiles;
    const int64_t zmax = (K + 127) / 128;
    if (z > zmax) z = zmax;
    if (z > GRID_Z_MAX / count) z = GRID_Z_MAX / count;
    while (z > 1 && (size_t)z * (size_t)M * (size_t)N > g.slab_floats) z--;
    if (z < 1) z = 1;
    int64_t k_per = (K + z - 1) / z;
    k_per = (k_per + hk - 1) / hk * hk;
    z = (K + k_per - 1) / k_per;
    dim3 grid((unsigned)((N + 127) / 128), (unsigned)((M + 127) / 128), (unsigned)(z * count)), block(256);
    float *dst = z == 1 ? C : g.slabs;
    const GemmEpilogue none{};
    const int sym = (symmetric && z == 1 && M == N && e.spt == nullptr && e.out == nullptr && e.col_bias == nullptr) ? 1 : 0;
    const HOperand A{a, 2 * K}, B{b, 2 * K};
#define MECHA_GEMMH(HK_, D_) MECHA_LAUNCH(MK_ROT_GEMM, (gemm128h_kernel<HK_, D_>), grid, block, 0, g.s, A, B, dst, M, N, K, k_per, (int)z, g.bt, z == 1 ? e : none, sym)
    static const int hd = [] { const char *v = getenv("MECHA_ROT_HD"); return v != nullptr ? atoi(v) : 2; }();
    if (hk == 32 && hd == 1) MECHA_GEMMH(32, 1);
    else if (hk == 32 && hd == 3) MECHA_GEMMH(32, 3);
    else if (hk == 32) MECHA_GEMMH(32, 2);
    else MECHA_GEMMH(16, 1);
#undef MECHA_GEMMH
    if (z > 1) {
        const int64_t total = M * N;
        const int64_t want = (total + 255) / 256, cap = count >= 64 ? 16 : 1024 / count;
        MECHA_LAUNCH(MK_ROT_GEMM, gemm_reduce_kernel, dim3((unsigned)(want < cap ? want : cap), (unsigned)count), dim3(256), 0, g.s, g.slabs, (int)z, C, M, N, e, g.bt);
    }
}

// Orthogonal polar factor of X (r x c row-major, r <= c), in place: X = U S V^T  ->  U V^T (zero singular values stay zero).
// Odd degree-5 iteration  X <- a_k X + (b_k G + c_k G^2) X,  G = X X^T,  with the coefficients of the OPTIMAL composition
// (Amsel, Persson, Musco, Gower 2025, "The Polar Express": the best composition of T such polynomials is greedy; step k is
// the minimax approximation of 1 on the interval [l_k, u_k] that still holds the singular values, found by a Remez exchange
// on its four equioscillation points, polar_schedule below).  Starting from [1e-7, 1] -- anything smaller against sigma_max
// is fp32 rounding noise of the products that made X -- 14 steps bring every singular value within 1e-6 of 1:
// the first steps grow a small singular value 5.6x, then 3.86x per step, against 2.5x for the fixed polynomial
// (2.5, -2.5, 1) this file used first (26.5 steps per SDXL key on average with a convergence test read by the host each step).
// The schedule is fixed, so nothing comes back to the host: no synchronisation inside a key.  X is scaled once, on the
// device, by |X X^T|_F^(-1/2) <= 1 / sigma_max (one norm kernel on the first Gram matrix, which is rescaled in place).
// tmp: r x c, G and G2: r x r, scal: device double[2].
struct PolarStep { float a, b, c; };
constexpr int POLAR_MAX_STEPS = 24;
struct PolarSchedule { int n; PolarStep step[POLAR_MAX_STEPS]; };

// minimax odd quintic a x + b x^3 + c x^5 for the constant 1 on [l, 1]: error E equioscillates at l, q, r, 1
void remez_odd5(double l, double &a, double &b, double &c, double &E) {
    double q = l < 1e-3 ? (3 * l > 0.15 ? 3 * l : 0.15) : l + (1 - l) * 0.3, r = l < 1e-3 ? 0.7 : l + (1 - l) * 0.7;
    a = 1.875; b = -1.25; c = 0.375; E = 0;
    for (int it = 0; it < 500; it++) {
        const double x[4] = {l, q, r, 1.0};
        double m[4][5];
        for (int i = 0; i < 4; i++) {
            const double x2 = x[i] * x[i];
            m[i][0] = x[i]; m[i][1] = x[i] * x2; m[i][2] = x[i] * x2 * x2; m[i][3] = (i & 1) ? -1.0 : 1.0; m[i][4] = 1.0;
        }
        for (int col = 0; col < 4; col++) {                       // Gauss-Jordan with partial pivoting
            int piv = col;
            for (int i = col + 1; i < 4; i++) if (fabs(m[i][col]) > fabs(m[piv][col])) piv = i;
            for (int j = 0; j < 5; j++) { const double t = m[col][j]; m[col][j] = m[piv][j]; m[piv][j] = t; }
            for (int i = 0; i < 4; i++) {
                if (i == col) continue;
                const double f = m[i][col] / m[col][col];
                for (int j = col; j < 5; j++) m[i][j] -= f * m[col][j];
            }
        }
        a = m[0][4] / m[0][0]; b = m[1][4] / m[1][1]; c = m[2][4] / m[2][2]; E = m[3][4] / m[3][3];
        const double disc = 9 * b * b - 20 * a * c;               // p'(x) = a + 3 b x^2 + 5 c x^4 = 0
        if (!(disc >= 0)) break;
        double q2 = (-3 * b - sqrt(disc)) / (10 * c), r2 = (-3 * b + sqrt(disc)) / (10 * c);
        q2 = sqrt(fabs(q2)); r2 = sqrt(fabs(r2));
        if (q2 > r2) { const double t = q2; q2 = r2; r2 = t; }
        const bool done = fabs(q2 - q) + fabs(r2 - r) < 1e-15;
        q = q2; r = r2;
        if (done) break;
    }
}

// A step made for a tiny l / u equioscillates between 1 - E ~ 0 and 1 + E ~ 2: it throws some LARGE singular values back to
// almost nothing, and lifting them again costs them their fp32 digits (measured on a 96 x 512 key: 1e-3 relative error of the
// factor).  So a step is never made for a ratio below POLAR_CUSHION: values above that fraction of the largest stay within
// [0.21, 1.79] of it, values below still grow 3.86x a step (4.26x without the cushion), 14 steps in all from 1e-7.
constexpr double POLAR_CUSHION = 0.1;
PolarSchedule build_polar_schedule(double l0) {
    PolarSchedule ps{};
    double l = l0, u = 1.0;
    for (ps.n = 0; ps.n < POLAR_MAX_STEPS;) {
        const bool cushioned = l / u < POLAR_CUSHION;
        const double rho = cushioned ? POLAR_CUSHION : l / u;
        double a, b, c, E;
        remez_odd5(rho, a, b, c, E);                          // on x / u
        const double u2 = u * u;
        ps.step[ps.n++] = PolarStep{(float)(a / u), (float)(b / (u * u2)), (float)(c / (u * u2 * u2))};
        if (!cushioned && E < 1e-6) break;                    // the next interval would be below fp32 resolution
        const double x = l / u, x2 = x * x, pl = x * (a + x2 * (b + x2 * c));   // the step rises monotonically on [0, rho u]
        l = (cushioned && pl < 1.0 - E) ? pl : 1.0 - E;
        u = (1.0 + E) * (1.0 + 1e-6);                         // a hair of room for the rounding of the products
    }
    return ps;
}
const PolarSchedule &polar_schedule() {
    static const PolarSchedule sched = build_polar_schedule(1e-7);
    return sched;
}
// The short schedule of the optimistic first phase (polar_newton_schulz): singular values down to POLAR_EASY_L of the scaled
// matrix's bound.  The scaling by |X X^T|_F^(-1/2) puts the largest singular value between r^(-1/4) and 1 (0.17 for 1280 rows),
// so 5e-3 covers condition numbers up to ~35 at that size -- a Gaussian 1280 x 2048 matrix has 9, 1280 x 5120 has 3.
constexpr double POLAR_EASY_L = 5e-3;
const PolarSchedule &polar_schedule_easy() {
    static const PolarSchedule sched = build_polar_schedule(POLAR_EASY_L);
    return sched;
}

// X *= s, G *= s^2 with s = (sum G^2)^(-1/4) read from scal[0] on the device (0 for a zero matrix: it stays zero)
__global__ __launch_bounds__(256) void rot_polar_scale_kernel(float *x, int64_t nx, float *g, int64_t ng, const double *scal, Batch bt) {
    const int key = blockIdx.y;
    if (bt.skip != nullptr && bt.skip[key] != 0u) return;
    x = key_ptr(x, bt, key); g = key_ptr(g, bt, key); scal = key_ptr(scal, bt, key);
    const double h = scal[0];
    const float s = (h > 0.0 && h < 1e300) ? (float)(1.0 / sqrt(sqrt(h))) : (h == h ? 0.0f : __uint_as_float(0x7fc00000u));
    const float s2 = s * s;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nx + ng; i += (int64_t)gridDim.x * 256) {
        if (i < nx) x[i] *= s; else g[i - nx] *= s2;
    }
}

// X (r x c row-major) <- (I - 1 1^T / r) X: every column loses its mean.  For a fully centred wide key the all-ones vector is
// the exact left null vector of both matrices the wide case takes a polar factor of (1^T Bc = 0; N^T = (Bc Qb)^T Ac with
// Bc Qb symmetric and 1 in its null space), but fp32 rounding leaves ~1e-8 of it in X and the iteration grows anything that
// small 3.86x per step -- back to a singular value of order 1 after 14 steps, along a direction made of rounding noise.
// Removing it every few steps keeps the zero singular value at rounding level, so the factor is the one of the rank-(r - 1)
// matrix (the minimum-norm member of the reference's family, DESIGN.md "rotate: reference indeterminacy").
// One block per 64 columns: 4 row groups x 64 columns, fp64 partial sums through LDS.
__global__ __launch_bounds__(256) void rot_deflate_ones_kernel(float *x, int64_t r, int64_t c, Batch bt) {
    __shared__ double part[4][64];
    const int key = blockIdx.y;
    if (bt.skip != nullptr && bt.skip[key] != 0u) return;
    x = key_ptr(x, bt, key);
    const int col_l = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int64_t col = (int64_t)blockIdx.x * 64 + col_l;
    double sum = 0.0;
    if (col < c)
        for (int64_t i = grp; i < r; i += 4) sum += (double)x[i * c + col];
    part[grp][col_l] = sum;
    __syncthreads();
    const float mean = (float)((part[0][col_l] + part[1][col_l] + part[2][col_l] + part[3][col_l]) / (double)r);
    if (col < c)
        for (int64_t i = grp; i < r; i += 4) x[i * c + col] -= mean;
}
constexpr int POLAR_DEFLATE_EVERY = 3;

// done[key] = 1 when the Gram matrix whose norms sit in scal says the key's polar iteration has converged:
// |G - I|_F^2 <= null + tol, `null` = the number of exactly-zero singular values the key is known to have (the deflated
// all-ones direction of a fully centred wide key).  One thread per key.
__global__ __launch_bounds__(64) void rot_converged_kernel(const double *scal, double null_dirs, double tol, uint32_t *done, Batch bt) {
    const int key = blockIdx.x * 64 + threadIdx.x;
    if (key >= bt.count) return;
    const double err2 = key_ptr(scal, bt, key)[1];
    done[key] = (err2 == err2 && err2 <= null_dirs + tol) ? 1u : 0u;
}

__global__ __launch_bounds__(64) void rot_zero_scal_kernel(double *scal, int slot, Batch bt) {
    const int key = blockIdx.x * 64 + threadIdx.x;
    if (key < bt.count) { double *s = key_ptr(scal, bt, key); s[slot] = 0.0; s[slot + 1] = 0.0; }
}
// scal[0] = |G|_F^2, scal[1] = |G - I|_F^2 of every key's r x r Gram matrix
void gram_norms(const float *G, int64_t r, double *scal, const GemmCtx &g) {
    const int count = g.bt.count;
    MECHA_LAUNCH(MK_ROT_NORMS, rot_zero_scal_kernel, dim3((unsigned)((count + 63) / 64)), dim3(64), 0, g.s, scal, 0, g.bt);
    const int64_t nbw = (r * r + 255) / 256, nb_cap = count >= 64 ? 16 : 1024 / count;
    MECHA_LAUNCH(MK_ROT_NORMS, rot_gram_norms_kernel, dim3((unsigned)(nbw < nb_cap ? nbw : nb_cap), (unsigned)count), dim3(256), 0, g.s, G, r, scal, g.bt);
}

// One step X <- a X + (b G + c G^2) X on every key of the batch that `g.bt.skip` does not excuse; returns the new current buffer.
// first: the scaling by |X X^T|_F^(-1/2) (see above) happens on this step's Gram matrix.
float *polar_step(float *cur, float *nxt, float *G, float *G2, int64_t r, int64_t c, double *scal, const PolarStep &st, bool first,
                  const GemmCtx &g) {
    const int64_t rr = r * r;
    const int count = g.bt.count;
    const GemmEpilogue none{};
    // G[m][n] = sum_k cur[m][k] cur[n][k]
    gemm(Operand{cur, 1, c}, Operand{cur, 1, c}, G, r, r, c, none, g, true);
    if (first) {
        gram_norms(G, r, scal, g);
        const int64_t both = r * c + rr;
        const int64_t sbw = (both + 255) / 256, sb_cap = count >= 64 ? 32 : 2048 / count;
        MECHA_LAUNCH(MK_ROT_NORMS, rot_polar_scale_kernel, dim3((unsigned)(sbw < sb_cap ? sbw : sb_cap), (unsigned)count), dim3(256), 0, g.s, cur, r * c, G, rr,
                     scal, g.bt);
    }
    // G2 = b G + c G G;  nxt = a cur + G2 cur
    GemmEpilogue mix{};
    mix.axpy = G; mix.axpy_a = st.b; mix.axpy_b = st.c;
    // (G and b G + c G G are symmetric bit for bit -- the same products in the same order on both sides of the diagonal --
    //  so both operands are read along their rows: contiguous 16-byte loads)
    gemm(Operand{G, r, 1}, Operand{G, r, 1}, G2, r, r, r, mix, g, true);
    GemmEpilogue upd{};
    upd.axpy = cur; upd.axpy_a = st.a; upd.axpy_b = 1.0f;
    gemm(Operand{G2, r, 1}, Operand{cur, c, 1}, nxt, r, c, r, upd, g);
    return nxt;
}

// The split-fp16 images a run of steps on gemm128h_kernel keeps beside the fp32 iterate: of the iterate and its transpose
// (two of each: current and next), of G and of b G + c G^2.
struct PolarImages { _Float16 *x[2], *xt[2], *g, *g2; };
// The same step on the split-fp16 kernel.  x_img / xt_img: images of cur and cur^T; all three products read rows of images:
// X X^T along the rows of X, G G along the rows of the symmetric G, G2 X as (rows of G2) x (rows of X^T).  G and b G + c G^2
// exist only as images, the axpy terms are read back from images (a batch streams every matrix from HBM -- ~100 flop per byte --
// so the bytes count: fp32 copies beside the images made such a step 1.6x the traffic).  The new iterate goes out as fp32 when
// an fp32 step follows (nxt != NULL) and as its two images when another step of this kind does.
// have_gram: g_img already holds the image of X X^T (the first step of a run makes it while it scales the iterate).
void polar_step_h(float *nxt, int64_t r, int64_t c, const PolarStep &st, const GemmCtx &g, const _Float16 *x_img, const _Float16 *xt_img,
                  _Float16 *g_img, _Float16 *g2_img, _Float16 *nx_img, _Float16 *nxt_img, bool have_gram = false) {
    if (!have_gram) {
        GemmEpilogue gram{};
        gram.sp = g_img;
        gemm_h(x_img, x_img, nullptr, r, r, c, gram, g, true);
    }
    GemmEpilogue mix{};
    mix.axpy_img = g_img; mix.axpy_a = st.b; mix.axpy_b = st.c; mix.sp = g2_img;
    gemm_h(g_img, g_img, nullptr, r, r, r, mix, g, true);
    GemmEpilogue upd{};
    upd.axpy_img = x_img; upd.axpy_a = st.a; upd.axpy_b = 1.0f; upd.sp = nx_img; upd.spt = nxt_img;
    gemm_h(g2_img, xt_img, nxt, r, c, r, upd, g);
}

// Polar factor of every key's X (r x c, r <= c), in place.
//   optimistic == false: the 14-step schedule made for singular values down to 1e-7 of the largest.
//   optimistic == true (wide keys: well conditioned as a rule -- a Gaussian n x p matrix has all its singular values
//       within (sqrt(p) +- sqrt(n)) sigma): first the short schedule made for [POLAR_EASY_L, 1]; then a Gram matrix of the
//       result tells, per key and on the device, whether every singular value has arrived (rot_converged_kernel); keys
//       that have leave, the others go through the long schedule as well (every step maps (0, u] into (0, u']: running
//       it on a matrix that is already partly converged is safe).  The host never learns which keys left: the later
//       launches are enqueued for the whole batch and their blocks exit on the key's done word.
// done: per-key device words (used only when optimistic).
// img (nullable): buffers for split-fp16 images; with them, and r >= 256, r and c multiples of 8, every step but the first of a run
//       (whose Gram matrix scales X) and the last POLAR_F32_TAIL of a schedule runs on the split-fp16 kernel (polar_step_h): each step's rounding
//       is a perturbation of the iterate that the following steps pull back to the same polar factor to first order (the
//       factor of X + D differs from that of X by ~|D| / (s_i + s_j) between singular directions i, j, the same sensitivity
//       the fp32 products have at a 10x smaller |D|), and the closing fp32 steps set the final orthogonality.
//       MECHA_ROT_PREC=f32 keeps every step on the fp32 kernel; MECHA_ROT_F32_TAIL=<n> overrides the tail length (tests).
constexpr int POLAR_F32_TAIL = 0;
int polar_newton_schulz(float *X, float *tmp, float *G, float *G2, int64_t r, int64_t c, double *scal, uint32_t *done, int *iters_out,
                        const GemmCtx &g_in, bool deflate_ones, bool optimistic, const PolarImages *img = nullptr) {
    GemmCtx g = g_in;
    // MECHA_ROT_POLAR=long: no optimistic phase; =both: every key goes through both schedules (tests of the fall-through)
    const char *mode = getenv("MECHA_ROT_POLAR");
    if (mode != nullptr && strcmp(mode, "long") == 0) optimistic = false;
    const bool force_both = mode != nullptr && strcmp(mode, "both") == 0;
    const char *prec = getenv("MECHA_ROT_PREC"), *tail_env = getenv("MECHA_ROT_F32_TAIL");
    const bool split_ok = img != nullptr && r >= 256 && (r & 7) == 0 && (c & 7) == 0 && !(prec != nullptr && strcmp(prec, "f32") == 0);
    const int f32_tail = tail_env != nullptr ? atoi(tail_env) : POLAR_F32_TAIL;
    float *cur = X, *nxt = tmp;
    int ib = 0;                                                 // img->x[ib], img->xt[ib]: the images of cur and cur^T ...
    bool t_valid = false;                                       // ... when this says so
    const int count = g.bt.count;
    // when: every POLAR_DEFLATE_EVERY steps and at the end; in a schedule of <= 8 steps at the end only (the direction starts at
    // rounding level, ~1e-8, and grows at most 5.6 x 3.86^5 = 4 800-fold over the short schedule's 6 steps: still 5e-5 when the
    // closing deflation removes it -- and every deflation costs an fp32 round trip of the iterate and new images)
    auto deflates_after = [&](int it, int n_steps) {
        return deflate_ones && (it + 1 == n_steps || (n_steps > 8 && (it + 1) % POLAR_DEFLATE_EVERY == 0));
    };
    auto deflate = [&](int it, int n_steps) {
        if (deflates_after(it, n_steps)) {
            MECHA_LAUNCH(MK_ROT_NORMS, rot_deflate_ones_kernel, dim3((unsigned)((c + 63) / 64), (unsigned)count), dim3(256), 0, g.s, cur, r, c, g.bt);
            t_valid = false;
        }
    };
    bool first = true;
    // step `it` of an n-step schedule
    auto step = [&](const PolarStep &st, int it, int n_steps) {
        const bool split = split_ok && it < n_steps - f32_tail;
        float *out;
        if (split) {
            const dim3 sgrid((unsigned)((c + 63) / 64), (unsigned)((r + 63) / 64), (unsigned)count);
            if (first) {
                // The scaling of the first step, without an fp32 product: images of t X (t: a power of two that brings |X|_F to
                // <= 1), G = (t X)(t X)^T from them in fp32, its norm -> s; then the images of t s X and of s^2 G.  The fp32 X
                // itself stays unscaled: nothing reads it again (the axpy terms come from images, the last step of the run
                // writes the fp32 iterate).
                MECHA_LAUNCH(MK_ROT_NORMS, rot_zero_scal_kernel, dim3((unsigned)((count + 63) / 64)), dim3(64), 0, g.s, scal, 2, g.bt);
                const int64_t nbw = (r * c + 255) / 256, nb_cap = count >= 64 ? 16 : 1024 / count;
                MECHA_LAUNCH(MK_ROT_NORMS, rot_sumsq_kernel, dim3((unsigned)(nbw < nb_cap ? nbw : nb_cap), (unsigned)count), dim3(256), 0, g.s, cur, r * c, scal, g.bt);
                MECHA_LAUNCH(MK_ROT_NORMS, rot_split_kernel, sgrid, dim3(256), 0, g.s, cur, img->x[ib], (_Float16 *)nullptr, r, c, (int)SPLIT_T, scal, g.bt);
                const GemmEpilogue none{};
                gemm_h(img->x[ib], img->x[ib], G, r, r, c, none, g, true);
                gram_norms(G, r, scal, g);
                MECHA_LAUNCH(MK_ROT_NORMS, rot_split_kernel, sgrid, dim3(256), 0, g.s, cur, img->x[ib], img->xt[ib], r, c, (int)SPLIT_TS, scal, g.bt);
                MECHA_LAUNCH(MK_ROT_NORMS, rot_split_kernel, dim3((unsigned)((r + 63) / 64), (unsigned)((r + 63) / 64), (unsigned)count), dim3(256), 0, g.s,
                             G, img->g, (_Float16 *)nullptr, r, r, (int)SPLIT_SS, scal, g.bt);
            } else if (!t_valid)
                MECHA_LAUNCH(MK_ROT_NORMS, rot_split_kernel, sgrid, dim3(256), 0, g.s, cur, img->x[ib], img->xt[ib], r, c, (int)SPLIT_PLAIN, scal, g.bt);
            // images for the next step unless it is an fp32 step or the deflation (which works on the fp32 iterate) comes first
            const bool deflates = deflates_after(it, n_steps);
            const bool next_split = it + 1 < n_steps - f32_tail && !deflates;
            // (the fp32 iterate is written by the last step of such a run only; in between, `nxt` is a stale buffer)
            polar_step_h(next_split ? nullptr : nxt, r, c, st, g, img->x[ib], img->xt[ib], img->g, img->g2, next_split ? img->x[ib ^ 1] : nullptr,
                         next_split ? img->xt[ib ^ 1] : nullptr, first);
            out = nxt;
            ib ^= 1;
            t_valid = next_split;
        } else {
            out = polar_step(cur, nxt, G, G2, r, c, scal, st, first, g);
            t_valid = false;
        }
        nxt = cur; cur = out; first = false;
        deflate(it, n_steps);
    };
    int steps = 0;
    if (optimistic) {
        const PolarSchedule &easy = polar_schedule_easy();
        for (int it = 0; it < easy.n; it++) step(easy.step[it], it, easy.n);
        steps += easy.n;
        // has every singular value arrived?  G of the current X, its distance from the identity, one word per key
        const GemmEpilogue none{};
        if (split_ok) {
            MECHA_LAUNCH(MK_ROT_NORMS, rot_split_kernel, dim3((unsigned)((c + 63) / 64), (unsigned)((r + 63) / 64), (unsigned)count), dim3(256), 0, g.s, cur,
                         img->x[ib], (_Float16 *)nullptr, r, c, (int)SPLIT_PLAIN, scal, g.bt);
            gemm_h(img->x[ib], img->x[ib], G, r, r, c, none, g, true);
        } else gemm(Operand{cur, 1, c}, Operand{cur, 1, c}, G, r, r, c, none, g, true);
        gram_norms(G, r, scal, g);
        // |G - I|_F^2 = sum (sigma^2 - 1)^2: 1e-9 r allows an rms deviation of 1.6e-5 per singular value; 1e-13 r^2 is the
        // rounding floor of the fp32 Gram matrix itself
        // (+ 4e-6 beside a deflated direction: the 1 it contributes is a sum of r^2 rounded entries of size 1 / r)
        const double tol = 1e-9 * (double)r + 1e-13 * (double)r * (double)r + (deflate_ones ? 4e-6 : 0.0);
        MECHA_LAUNCH(MK_ROT_NORMS, rot_converged_kernel, dim3((unsigned)((count + 63) / 64)), dim3(64), 0, g.s, scal, deflate_ones ? 1.0 : 0.0, tol, done, g.bt);
        // keys that are done must end in X: copy now if the short schedule ended in the twin buffer
        if (cur != X) {
            MECHA_CHECK_HIP(hipMemcpy2DAsync(X, (size_t)(g.bt.ws_bytes ? g.bt.ws_bytes : sizeof(float) * r * c), cur,
                                             (size_t)(g.bt.ws_bytes ? g.bt.ws_bytes : sizeof(float) * r * c), sizeof(float) * r * c, (size_t)count,
                                             hipMemcpyDeviceToDevice, g.s));
            cur = X; nxt = tmp;
        }
        t_valid = false;
        g.bt.skip = done;
    }
    const PolarSchedule &ps = polar_schedule();
    if (force_both) g.bt.skip = nullptr;
    for (int it = 0; it < ps.n; it++) {
        PolarStep st = ps.step[it];
        if (it == 0 && !first) {
            // The long schedule leaves a hair (1e-6) of room above its upper bounds, enough for a matrix scaled below 1 by
            // its own norm; one that arrives from the short schedule has singular values at 1 +- 1e-6, some of them above
            // 1, and the repeated cushioned step throws anything above its bound out (1 + 1e-5 overflows after 8 steps).
            // So the first step is taken on 0.97 X: p(0.97 x) = (0.97 a) x + (0.97^3 b) x^3 + (0.97^5 c) x^5.
            const float h = 0.97f, h3 = h * h * h;
            st = PolarStep{st.a * h, st.b * h3, st.c * h3 * h * h};
        }
        step(st, it, ps.n);
    }
    steps += ps.n;
    if (iters_out) *iters_out += steps;
    if (cur != X) {
        // (an odd number of steps; keys excused by `done` hold their result in X already, and their stale twin must not
        //  overwrite it: the schedules have an even number of steps, checked where they are built)
        mecha_set_error("polar schedule with an odd number of steps");
        return MECHA_ERR_ARG;
    }
    return MECHA_OK;
}

// Private workspace of one key, as offsets from its start; the same for every key of a shape.
struct RotLayout {
    size_t scal, part, a_mean, b_mean, t_mean, ac, bc, y, slabs, m0, m1, m2, m3, m4, m5, m6, m7, img_x0, img_x1, img_xt0, img_xt1, img_g, img_g2, img_q, img_qt, total;
    size_t slab_floats;
};
RotLayout rot_layout(int64_t n, int64_t p, int count) {
    RotLayout L{};
    const size_t q = (size_t)(n < p ? n : p);
    size_t cursor = 0;
    auto take = [&](size_t count_, size_t elem) { const size_t o = cursor; cursor = up256(cursor + count_ * elem); return o; };
    auto fl = [&](size_t count_) { return take(count_, sizeof(float)); };
    L.scal = take(8, sizeof(double));
    L.part = take((size_t)2 * 64 * p, sizeof(double));
    L.a_mean = fl(p); L.b_mean = fl(p); L.t_mean = fl(p);
    L.ac = fl(n * p); L.bc = fl(n * p); L.y = fl(n * p);
    // split-K slabs: what one launch may use is shared out over the keys of the batch, but a key always has room for
    // two slabs of its largest product (the products are at most max(n, p) x p)
    size_t slab = SLAB_FLOATS / (size_t)count;
    const size_t floor_sq = 2 * q * q;
    if (slab < floor_sq) slab = floor_sq;
    if (slab > SLAB_FLOATS) slab = SLAB_FLOATS;
    L.slab_floats = slab;
    L.slabs = fl(slab);
    if (n >= p) {                                               // M -> R, its Newton-Schulz twin, G, b G + c G^2
        L.m0 = fl(p * p); L.m1 = fl(p * p); L.m2 = fl(p * p); L.m3 = fl(p * p);
        // split-fp16 images (4 bytes a value, as fp32): the iterate, its transpose (current and next), G, b G + c G^2
        L.img_x0 = fl(p * p); L.img_x1 = fl(p * p); L.img_xt0 = fl(p * p); L.img_xt1 = fl(p * p); L.img_g = fl(p * p); L.img_g2 = fl(p * p);
    } else {                                                    // Qb^T, Z, twin; G, b G + c G^2, P, x P2^T, D
        L.m0 = fl(n * p); L.m1 = fl(n * p); L.m2 = fl(n * p);
        L.m3 = fl(n * n); L.m4 = fl(n * n); L.m5 = fl(n * n); L.m6 = fl(n * n); L.m7 = fl(n * n);
        L.img_x0 = fl(n * p); L.img_x1 = fl(n * p); L.img_xt0 = fl(n * p); L.img_xt1 = fl(n * p); L.img_g = fl(n * n); L.img_g2 = fl(n * n);
        L.img_q = fl(n * p); L.img_qt = fl(n * p);              // the first solve's result and its transpose, kept across the second
    }
    L.total = up256(cursor);
    return L;
}
// batched calls keep, ahead of the keys' workspaces, two words per key: `done` (polar convergence) and `not_close` (svd.py:46)
size_t batch_header_bytes(int count) { return 2 * up256(sizeof(uint32_t) * (size_t)count); }

// out = lerp(a, b, alpha) for the keys whose tensors are allclose in fp16 (svd.py:46-47); their flag word is recomputed
__global__ __launch_bounds__(256) void rot_same_kernel(const void *a, const void *b, void *out, int in_dt, int out_dt, int64_t numel, float alpha,
                                                       const uint32_t *not_close, uint32_t *flags, Batch bt) {
    const int key = blockIdx.y;
    if (not_close[key] != 0u) return;
    const int64_t base = key_off(bt, key);
    const float wm1 = alpha - 1.0f;
    const bool small = __builtin_fabsf(alpha) < 0.5f;
    bool bad_in = false, bad_out = false;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < numel; i += (int64_t)gridDim.x * 256) {
        const float x = ld_any(in_dt, a, base + i), y = ld_any(in_dt, b, base + i);
        const float v = lerp_torch(x, y, alpha, wm1, small);
        bad_in |= nonfinite_f32(x) | nonfinite_f32(y);
        bad_out |= nonfinite_f32(v);
        if (out_dt == MECHA_BF16) store1<MECHA_BF16>(out, base + i, v);
        else if (out_dt == MECHA_F16) store1<MECHA_F16>(out, base + i, v);
        else store1<MECHA_F32>(out, base + i, v);
    }
    if (flags != nullptr && (bad_in || bad_out)) atomicOr(flags + key, (bad_in ? MECHA_FLAG_INPUT_NONFINITE : 0u) | (bad_out ? MECHA_FLAG_OUTPUT_NONFINITE : 0u));
}
// the rotation's own output flag of such a key is void (its output was overwritten): clear it before rot_same_kernel runs
__global__ __launch_bounds__(64) void rot_same_clear_kernel(const uint32_t *not_close, uint32_t *flags, int count) {
    const int key = blockIdx.x * 64 + threadIdx.x;
    if (key < count && not_close[key] == 0u) flags[key] &= ~(uint32_t)MECHA_FLAG_OUTPUT_NONFINITE;
}

struct RotCall {
    const void *a, *b;
    void *out;
    const int64_t *offs;                    // device: element offset of every key inside a / b / out; nullptr = one key at 0
    int count;
    int64_t n, p;
    int in_dt, out_dt;
    float centralization;
    int alignment;
    float alpha;
    const float *transform_in;              // one key only
    float *transform_out;                   // one key only
    void *workspace;
    size_t workspace_bytes;
    int *sweeps_out;
    uint32_t *flags;                        // one word per key
    bool allclose_check;                    // batched form: svd.py:46 is evaluated on the device
    void *stream;
};

int rotate_run(const RotCall &c) {
    const int64_t n = c.n, p = c.p;
    const int count = c.count;
    const RotLayout L = rot_layout(n, p, count);
    const size_t header = c.offs != nullptr ? batch_header_bytes(count) : 0;
    const size_t need = header + L.total * (size_t)count;
    MECHA_REQUIRE(c.workspace != nullptr && c.workspace_bytes >= need && mecha_aligned16(c.workspace),
                  "rotate workspace too small or unaligned: need %zu bytes, got %zu", need, c.workspace_bytes);
    hipStream_t s = static_cast<hipStream_t>(c.stream);
    char *ws0 = static_cast<char *>(c.workspace);
    uint32_t *done = reinterpret_cast<uint32_t *>(ws0), *not_close = reinterpret_cast<uint32_t *>(ws0 + header / 2);
    char *ws = ws0 + header;                                    // key 0's workspace
    if (header) MECHA_CHECK_HIP(hipMemsetAsync(ws0, 0, header, s));
    const Batch bt{count, count > 1 ? (int64_t)L.total : 0, c.offs, nullptr};
    auto F = [&](size_t off) { return reinterpret_cast<float *>(ws + off); };
    double *scal = reinterpret_cast<double *>(ws + L.scal), *part = reinterpret_cast<double *>(ws + L.part);
    float *a_mean = F(L.a_mean), *b_mean = F(L.b_mean), *t_mean = F(L.t_mean), *ac = F(L.ac), *bc = F(L.bc), *y = F(L.y);
    GemmCtx g{bt, F(L.slabs), L.slab_floats, s};
    if (c.sweeps_out) *c.sweeps_out = 0;
    // a single key without the header keeps its done word in the scal block (8 doubles: words 0-1 norms, the rest spare)
    if (!header) done = reinterpret_cast<uint32_t *>(scal + 4);

    const int chunks = (int)((n + 255) / 256 < 64 ? (n + 255) / 256 : 64);
    // (batched calls: arena offsets are multiples of 128 elements, so every key is as aligned as the arenas are)
    const bool vec8 = (p & 7) == 0 && ((reinterpret_cast<uintptr_t>(c.a) | reinterpret_cast<uintptr_t>(c.b)) & 15u) == 0;
    if (vec8) {
        const dim3 cgrid((unsigned)((p + 511) / 512), (unsigned)chunks, (unsigned)count);
        uint32_t *nc = c.allclose_check ? not_close : nullptr;
        hipLaunchKernelGGL(rot_zero_scal_kernel, dim3((unsigned)((count + 63) / 64)), dim3(64), 0, s, scal, (int)ROT_SCAL_AMAX, bt);
        if (c.in_dt == MECHA_BF16) hipLaunchKernelGGL(rot_colsum8_kernel<MECHA_BF16>, cgrid, dim3(256), 0, s, c.a, c.b, n, p, chunks, part, c.flags, nc, scal, bt);
        else if (c.in_dt == MECHA_F16) hipLaunchKernelGGL(rot_colsum8_kernel<MECHA_F16>, cgrid, dim3(256), 0, s, c.a, c.b, n, p, chunks, part, c.flags, nc, scal, bt);
        else hipLaunchKernelGGL(rot_colsum8_kernel<MECHA_F32>, cgrid, dim3(256), 0, s, c.a, c.b, n, p, chunks, part, c.flags, nc, scal, bt);
    } else if (p <= COLSUM_SMALL_P)
        hipLaunchKernelGGL(rot_colsum_small_kernel, dim3(1, (unsigned)chunks, (unsigned)count), dim3(256), 0, s, c.a, c.b, c.in_dt, n, (int)p, chunks, part,
                           c.flags, bt);
    else
        hipLaunchKernelGGL(rot_colsum_kernel, dim3((unsigned)((p + 63) / 64), (unsigned)chunks, (unsigned)count), dim3(256), 0, s, c.a, c.b, c.in_dt, n, p,
                           chunks, part, c.flags, bt);
    hipLaunchKernelGGL(rot_means_kernel, dim3((unsigned)((p + 255) / 256), (unsigned)count), dim3(256), 0, s, part, chunks, n, p, c.centralization,
                       (float)c.alignment, a_mean, b_mean, t_mean, vec8 ? scal : (double *)nullptr, bt);
    const int64_t total = n * p;
    const int64_t cbw = (total + 255) / 256, cb_cap = count >= 64 ? 64 : 4096 / count;
    const unsigned cb = (unsigned)(cbw < cb_cap ? cbw : cb_cap);
    // Tall keys in the common configuration (rotate onto b, no interpolation) never hold their centred matrices in fp32: the
    // centring writes the three images its two big products read (A^T B and the application), on the split-fp16 kernel like the
    // polar steps between them (measured on SDXL shapes: the same 1e-6 relative error against the fp64 oracle either way).
    const char *prec_env = getenv("MECHA_ROT_PREC");
    const bool plain = vec8 && c.alignment == 1 && c.alpha == 0.0f && c.transform_in == nullptr && (n & 7) == 0 && (p & 7) == 0 &&
                       !(prec_env != nullptr && strcmp(prec_env, "f32") == 0);
    // (wide keys: the images of P = Bc Qb and D = x (Z - Qb^T)^T hold entries up to 2 sqrt(p) and 4 sqrt(p): within fp16's range / 64 up to p = 65536)
    const bool tall_h = plain && n >= p && p >= 256, wide_h = plain && n < p && n >= 256 && p <= 65536;
    _Float16 *ac_img = reinterpret_cast<_Float16 *>(ac), *bx_img = reinterpret_cast<_Float16 *>(bc), *act_img = reinterpret_cast<_Float16 *>(y);
    const dim3 tile_grid((unsigned)((p + 63) / 64), (unsigned)((n + 63) / 64), (unsigned)count);
    if (tall_h)             // bx_img: Bc^T
        hipLaunchKernelGGL(rot_center_img_kernel, tile_grid, dim3(256), 0, s, c.a, c.b, c.in_dt, n, p, a_mean, b_mean, ac_img, act_img, bx_img,
                           (_Float16 *)nullptr, (float *)nullptr, scal, bt);
    else if (wide_h)        // bx_img: Bc; its fp32 copy goes where the first polar solve works (m0)
        hipLaunchKernelGGL(rot_center_img_kernel, tile_grid, dim3(256), 0, s, c.a, c.b, c.in_dt, n, p, a_mean, b_mean, ac_img, act_img, (_Float16 *)nullptr,
                           bx_img, F(L.m0), scal, bt);
    else
        hipLaunchKernelGGL(rot_center_kernel, dim3(cb, (unsigned)count), dim3(256), 0, s, c.a, c.b, c.in_dt, n, p, a_mean, b_mean, ac, bc, bt);
    if (c.allclose_check && !vec8)
        hipLaunchKernelGGL(rot_allclose_kernel, dim3(cb, (unsigned)count), dim3(256), 0, s, c.a, c.b, c.in_dt, total, not_close, bt);

    auto H = [&](size_t off) { return reinterpret_cast<_Float16 *>(ws + off); };
    const PolarImages img{{H(L.img_x0), H(L.img_x1)}, {H(L.img_xt0), H(L.img_xt1)}, H(L.img_g), H(L.img_g2)};
    const bool has_alpha = c.alpha != 0.0f;
    const int alignment = c.alignment;
    const float alpha = c.alpha;
    const GemmEpilogue none{};
    GemmEpilogue fin{};                                         // + lerp(a_mean, b_mean, alignment), cast, finite flag  (svd.py:85-86)
    fin.col_bias = t_mean; fin.out = c.out; fin.out_dt = c.out_dt; fin.flags = c.flags;
    int rc;
    if (tall_h) {
        float *R = F(L.m0), *twin = F(L.m1), *G = F(L.m2), *G2 = F(L.m3);
        // M = Ac^T Bc as (rows of Ac^T) x (rows of Bc^T);  R = its polar factor;  out = Ac R + mean as (rows of Ac) x (rows of R^T)
        gemm_h(act_img, bx_img, R, p, p, n, none, g);
        if ((rc = polar_newton_schulz(R, twin, G, G2, p, p, scal, done, c.sweeps_out, g, false, false, &img)) != MECHA_OK) return rc;
        if (c.transform_out != nullptr) MECHA_CHECK_HIP(hipMemcpyAsync(c.transform_out, R, sizeof(float) * p * p, hipMemcpyDeviceToDevice, s));
        MECHA_LAUNCH(MK_ROT_NORMS, rot_split_kernel, dim3((unsigned)((p + 63) / 64), (unsigned)((p + 63) / 64), (unsigned)count), dim3(256), 0, s, R, img.x[0],
                     img.xt[0], p, p, (int)SPLIT_PLAIN, scal, bt);
        GemmEpilogue ea = fin;
        ea.acc_scale = reinterpret_cast<const float *>(scal + ROT_SCAL_INV_TA);          // the image of Ac holds t_a Ac
        gemm_h(ac_img, img.xt[0], nullptr, n, p, p, ea, g);
    } else if (wide_h) {
        // The wide case (below) on the split kernel, every product from images:
        //   Qb^T = polar(Bc);  P = Bc Qb as (rows of Bc) x (rows of Qb^T), kept as the image of P^T;  N^T = P^T Ac as (rows of P^T) x
        //   (rows of Ac^T);  Z = polar(N^T);  out = x + (x Z^T - x Qb) Qb^T + mean with x = Ac:  D = x (Z - Qb^T)^T as (rows of x) x
        //   (rows of Z - Qb^T) -- one product instead of two, the difference taken in fp32 before it is split --, then (rows of D) x
        //   (rows of Qb), the pass-through x read back from its image.
        float *qbt = F(L.m0), *z = F(L.m1), *twin = F(L.m2), *G = F(L.m3), *G2 = F(L.m4);
        _Float16 *q_img = H(L.img_q), *qt_img = H(L.img_qt), *pt_img = img.g2, *zq_img = img.x[0], *d_img = img.g;
        const bool deflate = c.centralization == 1.0f;
        if ((rc = polar_newton_schulz(qbt, twin, G, G2, n, p, scal, done, c.sweeps_out, g, deflate, true, &img)) != MECHA_OK) return rc;
        MECHA_LAUNCH(MK_ROT_NORMS, rot_split_kernel, tile_grid, dim3(256), 0, s, qbt, q_img, qt_img, n, p, (int)SPLIT_PLAIN, scal, bt);
        GemmEpilogue ep{};
        ep.spt = pt_img;
        gemm_h(bx_img, q_img, nullptr, n, n, p, ep, g);
        gemm_h(pt_img, act_img, z, n, p, n, none, g);
        if ((rc = polar_newton_schulz(z, twin, G, G2, n, p, scal, done, c.sweeps_out, g, deflate, true, &img)) != MECHA_OK) return rc;
        if (c.transform_out != nullptr) {
            MECHA_CHECK_HIP(hipMemcpyAsync(c.transform_out, qbt, sizeof(float) * n * p, hipMemcpyDeviceToDevice, s));
            MECHA_CHECK_HIP(hipMemcpyAsync(c.transform_out + n * p, z, sizeof(float) * n * p, hipMemcpyDeviceToDevice, s));
        }
        hipLaunchKernelGGL(rot_diff_img_kernel, dim3(cb, (unsigned)count), dim3(256), 0, s, z, qbt, zq_img, n, p, bt);
        GemmEpilogue ed{};
        ed.sp = d_img;
        gemm_h(ac_img, zq_img, nullptr, n, n, p, ed, g);
        GemmEpilogue ef = fin;
        ef.axpy_img = ac_img; ef.axpy_a = 1.0f; ef.axpy_b = 1.0f;
        ef.acc_scale = reinterpret_cast<const float *>(scal + ROT_SCAL_INV_TA);          // x and D both carry t_a
        gemm_h(d_img, qt_img, nullptr, n, p, n, ef, g);
    } else if (n >= p) {
        float *R = F(L.m0), *twin = F(L.m1), *G = F(L.m2), *G2 = F(L.m3);
        if (c.transform_in != nullptr) {
            MECHA_CHECK_HIP(hipMemcpyAsync(R, c.transform_in, sizeof(float) * p * p, hipMemcpyDeviceToDevice, s));
        } else {
            // M = Ac^T Bc:  M[i][j] = sum_r Ac[r][i] Bc[r][j];  R = U V^T = the orthogonal polar factor of M
            gemm(Operand{ac, p, 1}, Operand{bc, p, 1}, R, p, p, n, none, g);
            if ((rc = polar_newton_schulz(R, twin, G, G2, p, p, scal, done, c.sweeps_out, g, false, false, &img)) != MECHA_OK) return rc;
        }
        if (c.transform_out != nullptr) MECHA_CHECK_HIP(hipMemcpyAsync(c.transform_out, R, sizeof(float) * p * p, hipMemcpyDeviceToDevice, s));
        const float *src = ac;
        if (has_alpha) {
            // lerp(Ac, Bc R^T, alpha)  (svd.py:81-82; transform(b_2d, -1) = b_2d @ R.mH):  (Bc R^T)[r][i] = sum_j Bc[r][j] R[i][j]
            GemmEpilogue e = alignment == 0 ? fin : none;
            e.axpy = ac; e.axpy_a = 1.0f - alpha; e.axpy_b = alpha;
            gemm(Operand{bc, 1, p}, Operand{R, 1, p}, y, n, p, p, e, g);
            src = y;
        }
        // out[r][j] = sum_i src[r][i] R[i][j] + mean[j]   (alignment 1),  sum_i src[r][i] R[j][i] + mean[j]   (alignment -1)
        if (alignment == 1) gemm(Operand{src, 1, p}, Operand{R, p, 1}, nullptr, n, p, p, fin, g);
        else if (alignment == -1) gemm(Operand{src, 1, p}, Operand{R, 1, p}, nullptr, n, p, p, fin, g);
    } else {
        float *qbt = F(L.m0), *z = F(L.m1), *twin = F(L.m2);
        float *G = F(L.m3), *G2 = F(L.m4), *P = F(L.m5), *xq = F(L.m6), *D = F(L.m7);
        const size_t pitch = bt.ws_bytes ? (size_t)bt.ws_bytes : sizeof(float) * n * p;
        if (c.transform_in != nullptr) {
            MECHA_CHECK_HIP(hipMemcpyAsync(qbt, c.transform_in, sizeof(float) * n * p, hipMemcpyDeviceToDevice, s));
            MECHA_CHECK_HIP(hipMemcpyAsync(z, c.transform_in + n * p, sizeof(float) * n * p, hipMemcpyDeviceToDevice, s));
        } else {
            // Qb^T = polar factor of Bc: orthonormal rows spanning the row space of Bc (= columns of b.mH)
            MECHA_CHECK_HIP(hipMemcpy2DAsync(qbt, pitch, bc, pitch, sizeof(float) * n * p, (size_t)count, hipMemcpyDeviceToDevice, s));
            // fully centred (the default): both wide-case matrices have the all-ones vector as their exact left null vector
            const bool deflate = c.centralization == 1.0f;
            if ((rc = polar_newton_schulz(qbt, twin, G, G2, n, p, scal, done, c.sweeps_out, g, deflate, true, &img)) != MECHA_OK) return rc;
            // T^T = Bc Qb =: P (n x n):  P[r][k] = sum_c Bc[r][c] Qb^T[k][c]
            gemm(Operand{bc, 1, p}, Operand{qbt, 1, p}, P, n, n, p, none, g);
            // N^T = P^T Ac (n x p):  N^T[k][i] = sum_r P[r][k] Ac[r][i];  N = U S W^T, Z := polar(N^T) = W U^T
            gemm(Operand{P, n, 1}, Operand{ac, p, 1}, z, n, p, n, none, g);
            if ((rc = polar_newton_schulz(z, twin, G, G2, n, p, scal, done, c.sweeps_out, g, deflate, true, &img)) != MECHA_OK) return rc;
        }
        if (c.transform_out != nullptr) {
            MECHA_CHECK_HIP(hipMemcpyAsync(c.transform_out, qbt, sizeof(float) * n * p, hipMemcpyDeviceToDevice, s));
            MECHA_CHECK_HIP(hipMemcpyAsync(c.transform_out + n * p, z, sizeof(float) * n * p, hipMemcpyDeviceToDevice, s));
        }
        // With V = Qb W (the reference's v) and U (its u):  U V^T = Z^T Qb^T,  V V^T = Qb Qb^T,  U U^T = Z^T Z,  V U^T = Qb Z.  So
        //   transform(x, +1) = x - x V V^T + x U V^T = x + (x Z^T - x Qb) Qb^T      (svd.py:181)   -> low_rank(x, Z, Qb^T)
        //   transform(x, -1) = x - x U U^T + x V U^T = x + (x Qb - x Z^T) Z         (svd.py:183)   -> low_rank(x, Qb^T, Z)
        auto low_rank = [&](const float *x, const float *P1, const float *P2, float *C, GemmEpilogue e) {
            gemm(Operand{x, 1, p}, Operand{P2, 1, p}, xq, n, n, p, none, g);                // xq[r][k] = sum_c x[r][c] P2[k][c]
            GemmEpilogue minus_xq{};
            minus_xq.minus = xq;
            gemm(Operand{x, 1, p}, Operand{P1, 1, p}, D, n, n, p, minus_xq, g);             // D = x P1^T - x P2^T
            e.plus = x;
            gemm(Operand{D, 1, n}, Operand{P2, p, 1}, C, n, p, n, e, g);                    // C[r][c] = x[r][c] + sum_k D[r][k] P2[k][c]
        };
        const float *src = ac;
        if (has_alpha) {
            GemmEpilogue e = alignment == 0 ? fin : none;
            e.mix = ac; e.mix_w0 = 1.0f - alpha; e.mix_w1 = alpha;                          // lerp(Ac, transform(Bc, -1), alpha)
            low_rank(bc, qbt, z, y, e);
            src = y;
        }
        if (alignment == 1) low_rank(src, z, qbt, nullptr, fin);
        else if (alignment == -1) low_rank(src, qbt, z, nullptr, fin);
    }
    if (c.allclose_check) {
        if (c.flags != nullptr)
            hipLaunchKernelGGL(rot_same_clear_kernel, dim3((unsigned)((count + 63) / 64)), dim3(64), 0, s, not_close, c.flags, count);
        hipLaunchKernelGGL(rot_same_kernel, dim3(cb, (unsigned)count), dim3(256), 0, s, c.a, c.b, c.out, c.in_dt, c.out_dt, total, alpha, not_close,
                           c.flags, bt);
    }
    MECHA_CHECK_HIP(hipGetLastError());
    return MECHA_OK;
}

// ------------------------------------------------------------------------------------------------ truncate_rank
// Best rank-r approximation of a key (svd.py:89-143, exact mode: u[:, :r] s[:r] vh[:r] of the full SVD).  By Eckart-Young it is
// A P (or P A on the smaller side), P the orthogonal projector on the r leading eigenvectors of the Gram matrix W = A^T A (A A^T).
// No factor of the SVD is needed on its own: with a threshold tau between the r-th and (r+1)-th eigenvalue,
//     P = (I + sign(W - tau I)) / 2,     sign(S) = the polar factor of the symmetric S,
// and the polar factor is what polar_newton_schulz computes with products alone.  tau is found by bisection on
//     count(tau) = #{eigenvalues > tau} = (q + trace sign(W - tau I)) / 2,
// one polar solve per trial (q x q), driven from the host (two doubles read back per trial).  A trial whose threshold lands
// within fp32 resolution of an eigenvalue leaves that eigenvalue's sign unconverged (|sign|_F^2 < q): it still steers the
// bisection but is never accepted.  If the r-th and (r+1)-th eigenvalues cannot be told apart at all (a tie, or both below the
// Gram matrix's fp32 floor ~1e-7 |W|: singular values under ~3e-4 of the largest) the projector of the largest count <= r found
// is used -- the reference's own answer is arbitrary inside such a cluster.
__global__ __launch_bounds__(256) void trunc_load_kernel(const void *a, int dt, int64_t numel, float *a32, uint32_t *flags) {
    bool bad = false;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < numel; i += (int64_t)gridDim.x * 256) {
        const float v = ld_any(dt, a, i);
        bad |= nonfinite_f32(v);
        a32[i] = v;
    }
    if (bad && flags != nullptr) atomicOr(flags, MECHA_FLAG_INPUT_NONFINITE);
}
// x = w - tau I  (q x q)
__global__ __launch_bounds__(256) void trunc_shift_kernel(const float *w, float *x, int64_t q, float tau) {
    const int64_t total = q * q;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) x[i] = w[i] - ((i / q == i % q) ? tau : 0.0f);
}
// out[0] = trace(x) (one block, fp64)
__global__ __launch_bounds__(256) void trunc_trace_kernel(const float *x, int64_t q, double *out) {
    __shared__ double sh[4];
    double s0 = 0.0;
    for (int64_t i = threadIdx.x; i < q; i += 256) s0 += (double)x[i * q + i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s0 += __shfl_down(s0, off, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s0;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = sh[0] + sh[1] + sh[2] + sh[3];
}
// x = -I
__global__ __launch_bounds__(256) void trunc_minus_identity_kernel(float *x, int64_t q) {
    const int64_t total = q * q;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) x[i] = (i / q == i % q) ? -1.0f : 0.0f;
}

struct TruncLayout { size_t scal, a32, w, x, twin, g, g2, best, slabs, img[6], total, slab_floats; bool images; };
TruncLayout trunc_layout(int64_t m, int64_t n) {
    TruncLayout L{};
    const size_t q = (size_t)(m < n ? m : n);
    size_t cursor = 0;
    auto fl = [&](size_t count_) { const size_t o = cursor; cursor = up256(cursor + count_ * sizeof(float)); return o; };
    L.scal = cursor; cursor = up256(cursor + 8 * sizeof(double));
    L.a32 = fl((size_t)m * (size_t)n);
    L.w = fl(q * q); L.x = fl(q * q); L.twin = fl(q * q); L.g = fl(q * q); L.g2 = fl(q * q); L.best = fl(q * q);
    L.slab_floats = 2 * q * q > SLAB_FLOATS ? SLAB_FLOATS : (2 * q * q < 4096 ? 4096 : 2 * q * q);
    L.slabs = fl(L.slab_floats);
    L.images = q >= 256 && (q & 7) == 0;
    for (int i = 0; i < 6; i++) L.img[i] = L.images ? fl(q * q) : 0;
    L.total = up256(cursor);
    return L;
}
constexpr int TRUNC_MAX_TRIALS = 48;

int truncate_rank_run(const void *a, void *out, int64_t m, int64_t n, int64_t rank, int in_dt, int out_dt, void *workspace, size_t workspace_bytes,
                      int *trials_out, uint32_t *flags, void *stream) {
    const TruncLayout L = trunc_layout(m, n);
    MECHA_REQUIRE(workspace != nullptr && workspace_bytes >= L.total && mecha_aligned16(workspace),
                  "truncate_rank workspace too small or unaligned: need %zu bytes, got %zu", L.total, workspace_bytes);
    hipStream_t s = static_cast<hipStream_t>(stream);
    char *ws = static_cast<char *>(workspace);
    auto F = [&](size_t off) { return reinterpret_cast<float *>(ws + off); };
    auto H = [&](size_t off) { return reinterpret_cast<_Float16 *>(ws + off); };
    double *scal = reinterpret_cast<double *>(ws + L.scal);
    float *a32 = F(L.a32), *W = F(L.w), *X = F(L.x), *twin = F(L.twin), *G = F(L.g), *G2 = F(L.g2), *best = F(L.best);
    const PolarImages img{{H(L.img[0]), H(L.img[1])}, {H(L.img[2]), H(L.img[3])}, H(L.img[4]), H(L.img[5])};
    const GemmCtx g{ONE_KEY, F(L.slabs), L.slab_floats, s};
    const bool left = m <= n;                                   // the Gram matrix of the smaller side
    const int64_t q = left ? m : n, total = m * n, qq = q * q;
    const unsigned eb = (unsigned)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048), qb = (unsigned)((qq + 255) / 256 < 1024 ? (qq + 255) / 256 : 1024);
    hipLaunchKernelGGL(trunc_load_kernel, dim3(eb), dim3(256), 0, s, a, in_dt, total, a32, flags);
    const GemmEpilogue none{};
    if (left) gemm(Operand{a32, 1, n}, Operand{a32, 1, n}, W, m, m, n, none, g, true);          // W = A A^T
    else gemm(Operand{a32, n, 1}, Operand{a32, n, 1}, W, n, n, m, none, g, true);               // W = A^T A
    // trace W = sum of the eigenvalues, |W|_F >= the largest: the r-th eigenvalue lies in [0, min(|W|_F, trace / r)]
    hipLaunchKernelGGL(rot_zero_scal_kernel, dim3(1), dim3(64), 0, s, scal, 2, ONE_KEY);
    hipLaunchKernelGGL(rot_sumsq_kernel, dim3(qb), dim3(256), 0, s, W, qq, scal, ONE_KEY);
    hipLaunchKernelGGL(trunc_trace_kernel, dim3(1), dim3(256), 0, s, W, q, scal + 3);
    double host[2];
    MECHA_CHECK_HIP(hipMemcpyAsync(host, scal + 2, sizeof(host), hipMemcpyDeviceToHost, s));
    MECHA_CHECK_HIP(hipStreamSynchronize(s));
    const double w_fro = sqrt(host[0]), w_trace = host[1];
    hipLaunchKernelGGL(trunc_minus_identity_kernel, dim3(qb), dim3(256), 0, s, best, q);        // count 0: the projector on nothing
    int64_t best_count = 0;
    int trials = 0;
    if (w_fro > 0.0 && w_fro == w_fro && w_fro < 1e300) {
        // bisection on t = sqrt(tau) (the singular-value scale: eigenvalues of W span twice the orders of magnitude)
        const double cap = w_trace / (double)rank < w_fro ? w_trace / (double)rank : w_fro;
        double lo = 0.0, hi = sqrt(cap) * (1.0 + 1e-6);
        const double hi0 = hi;
        for (; trials < TRUNC_MAX_TRIALS && hi - lo > 1e-7 * hi0; ) {
            const double t = 0.5 * (lo + hi);
            trials++;
            hipLaunchKernelGGL(trunc_shift_kernel, dim3(qb), dim3(256), 0, s, W, X, q, (float)(t * t));
            int rc = polar_newton_schulz(X, twin, G, G2, q, q, scal, reinterpret_cast<uint32_t *>(scal + 4), nullptr, g, false, false, L.images ? &img : nullptr);
            if (rc != MECHA_OK) return rc;
            hipLaunchKernelGGL(rot_zero_scal_kernel, dim3(1), dim3(64), 0, s, scal, 2, ONE_KEY);
            hipLaunchKernelGGL(rot_sumsq_kernel, dim3(qb), dim3(256), 0, s, X, qq, scal, ONE_KEY);
            hipLaunchKernelGGL(trunc_trace_kernel, dim3(1), dim3(256), 0, s, X, q, scal + 3);
            MECHA_CHECK_HIP(hipMemcpyAsync(host, scal + 2, sizeof(host), hipMemcpyDeviceToHost, s));
            MECHA_CHECK_HIP(hipStreamSynchronize(s));
            if (!(host[0] == host[0]) || !(host[1] == host[1])) break;       // NaN: non-finite input, flagged already
            const bool converged = fabs(host[0] - (double)q) < 0.01;        // |sign|_F^2 = q when every eigenvalue reached +-1
            const int64_t count = (int64_t)llround(0.5 * ((double)q + host[1]));
            if (count > rank) lo = t; else hi = t;
            if (converged && count <= rank && count > best_count) {
                MECHA_CHECK_HIP(hipMemcpyAsync(best, X, sizeof(float) * qq, hipMemcpyDeviceToDevice, s));
                best_count = count;
            }
            if (converged && count == rank) break;
        }
    }
    if (trials_out) *trials_out = trials;
    // out = A P = (A + A sign) / 2   (P A = (A + sign A) / 2 on the left)
    GemmEpilogue fin{};
    fin.axpy = a32; fin.axpy_a = 0.5f; fin.axpy_b = 0.5f; fin.out = out; fin.out_dt = out_dt; fin.flags = flags;
    if (left) gemm(Operand{best, 1, m}, Operand{a32, n, 1}, nullptr, m, n, m, fin, g);          // C[i][j] = sum_k sign[i][k] A[k][j]
    else gemm(Operand{a32, 1, n}, Operand{best, 1, n}, nullptr, m, n, n, fin, g);               // C[i][j] = sum_k A[i][k] sign[j][k]
    MECHA_CHECK_HIP(hipGetLastError());
    return MECHA_OK;
}

}  // namespace

extern "C" {

size_t mecha_rotate_workspace_bytes(int64_t n_rows, int64_t n_cols) {
    if (n_rows < 1) n_rows = 1;
    if (n_cols < 1) n_cols = 1;
    return rot_layout(n_rows, n_cols, 1).total;
}

size_t mecha_batched_rotate_workspace_bytes(int64_t n_rows, int64_t n_cols, int n_keys) {
    if (n_rows < 1) n_rows = 1;
    if (n_cols < 1) n_cols = 1;
    if (n_keys < 1) n_keys = 1;
    return batch_header_bytes(n_keys) + rot_layout(n_rows, n_cols, n_keys).total * (size_t)n_keys;
}

int64_t mecha_rotate_transform_floats(int64_t n_rows, int64_t n_cols) {
    if (n_rows < 1 || n_cols < 1) return 0;
    return n_rows >= n_cols ? n_cols * n_cols : 2 * n_rows * n_cols;
}

int mecha_rotate_ex(const void *a, const void *b, void *out, int64_t n, int64_t p, int in_dtype, int out_dtype, float centralization,
                    int alignment, float alpha, const float *transform_in, float *transform_out, void *workspace,
                    size_t workspace_bytes, int *sweeps_out, uint32_t *flags, void *stream) {
    MECHA_REQUIRE(a != nullptr && b != nullptr && out != nullptr, "NULL tensor pointer");
    MECHA_REQUIRE(n >= 1 && p >= 1, "rotate needs a non-empty 2-D key (got %lld x %lld)", (long long)n, (long long)p);
    MECHA_REQUIRE(mecha_valid_dtype(in_dtype) && mecha_valid_dtype(out_dtype), "unknown dtype code");
    MECHA_REQUIRE(alignment >= -1 && alignment <= 1, "rotate: alignment must be -1, 0 or 1 (fractional powers are outside this path), got %d", alignment);
    MECHA_REQUIRE(alpha == alpha, "rotate: alpha is NaN");
    MECHA_REQUIRE(alignment != 0 || alpha != 0.0f, "rotate: alignment == 0 with alpha == 0 is the identity (svd.py:39-40); return a instead");
    MECHA_REQUIRE(transform_in == nullptr || mecha_aligned16(transform_in), "rotate: transform_in must be 16-byte aligned");
    RotCall c{a, b, out, nullptr, 1, n, p, in_dtype, out_dtype, centralization, alignment, alpha, transform_in, transform_out, workspace,
              workspace_bytes, sweeps_out, flags, false, stream};
    return rotate_run(c);
}

int mecha_rotate(const void *a, const void *b, void *out, int64_t n, int64_t p, int in_dtype, int out_dtype, float centralization,
                 void *workspace, size_t workspace_bytes, int *sweeps_out, uint32_t *flags, void *stream) {
    return mecha_rotate_ex(a, b, out, n, p, in_dtype, out_dtype, centralization, 1, 0.0f, nullptr, nullptr, workspace, workspace_bytes,
                           sweeps_out, flags, stream);
}

// Every key of one 2-D shape in one launch sequence (DESIGN.md "rotate": batches).  The keys lie in the per-model arenas at
// key_offsets[i] elements (DEVICE table, offsets multiples of 8 elements); flags: one word per key.  svd.py:46's
// allclose(a.half(), b.half()) early-out is evaluated per key on the device: such keys get lerp(a, b, alpha).
int mecha_batched_rotate(const void *a_arena, const void *b_arena, void *out_arena, const int64_t *key_offsets, int n_keys, int64_t n,
                         int64_t p, int in_dtype, int out_dtype, float centralization, int alignment, float alpha, void *workspace,
                         size_t workspace_bytes, uint32_t *flags, void *stream) {
    MECHA_REQUIRE(a_arena != nullptr && b_arena != nullptr && out_arena != nullptr && key_offsets != nullptr, "NULL pointer");
    MECHA_REQUIRE(n_keys >= 1 && n_keys <= 16384, "batched rotate takes 1..16384 keys per call (got %d)", n_keys);
    MECHA_REQUIRE(n >= 1 && p >= 1, "rotate needs a non-empty 2-D key (got %lld x %lld)", (long long)n, (long long)p);
    MECHA_REQUIRE(mecha_valid_dtype(in_dtype) && mecha_valid_dtype(out_dtype), "unknown dtype code");
    MECHA_REQUIRE(alignment >= -1 && alignment <= 1, "rotate: alignment must be -1, 0 or 1 (fractional powers are outside this path), got %d", alignment);
    MECHA_REQUIRE(alpha == alpha, "rotate: alpha is NaN");
    MECHA_REQUIRE(alignment != 0 || alpha != 0.0f, "rotate: alignment == 0 with alpha == 0 is the identity (svd.py:39-40); return a instead");
    RotCall c{a_arena, b_arena, out_arena, key_offsets, n_keys, n, p, in_dtype, out_dtype, centralization, alignment, alpha, nullptr, nullptr,
              workspace, workspace_bytes, nullptr, flags, true, stream};
    return rotate_run(c);
}

size_t mecha_truncate_rank_workspace_bytes(int64_t n_rows, int64_t n_cols) {
    if (n_rows < 1) n_rows = 1;
    if (n_cols < 1) n_cols = 1;
    return trunc_layout(n_rows, n_cols).total;
}

// truncate_rank, exact mode (svd.py:89-143 with use_approximate_basis=False), of one key flattened to n_rows x n_cols
// (a.flatten(start_dim=1), svd.py:110); 0 < rank < min(n_rows, n_cols) (the reference returns `a` or zeros outside, svd.py:111-116:
// the caller's shortcut).  Synchronises the stream once per bisection trial; *trials_out (nullable) = polar solves run.
int mecha_truncate_rank(const void *a, void *out, int64_t n_rows, int64_t n_cols, int64_t rank, int in_dtype, int out_dtype, void *workspace,
                        size_t workspace_bytes, int *trials_out, uint32_t *flags, void *stream) {
    MECHA_REQUIRE(a != nullptr && out != nullptr, "NULL tensor pointer");
    MECHA_REQUIRE(n_rows >= 2 && n_cols >= 2, "truncate_rank needs a 2-D key with both sides >= 2 (got %lld x %lld)", (long long)n_rows, (long long)n_cols);
    MECHA_REQUIRE(mecha_valid_dtype(in_dtype) && mecha_valid_dtype(out_dtype), "unknown dtype code");
    MECHA_REQUIRE(rank > 0 && rank < (n_rows < n_cols ? n_rows : n_cols), "truncate_rank: rank must lie strictly between 0 and min(rows, cols) = %lld, got %lld",
                  (long long)(n_rows < n_cols ? n_rows : n_cols), (long long)rank);
    return truncate_rank_run(a, out, n_rows, n_cols, rank, in_dtype, out_dtype, workspace, workspace_bytes, trials_out, flags, stream);
}

// torch.allclose(a.half(), b.half()) of svd.py:46.  ORs 1 into *not_close (device word, zeroed by the caller) if any element differs.
int mecha_allclose_half(const void *a, const void *b, int64_t numel, int in_dtype, uint32_t *not_close, void *stream) {
    MECHA_REQUIRE(numel >= 0 && mecha_valid_dtype(in_dtype) && not_close != nullptr, "bad arguments");
    if (numel == 0) return MECHA_OK;
    MECHA_REQUIRE(a != nullptr && b != nullptr, "NULL tensor pointer");
    const unsigned blocks = (unsigned)((numel + 255) / 256 < 2048 ? (numel + 255) / 256 : 2048);
    hipLaunchKernelGGL(rot_allclose_kernel, dim3(blocks, 1), dim3(256), 0, static_cast<hipStream_t>(stream), a, b, in_dtype, numel, not_close, ONE_KEY);
    MECHA_CHECK_HIP(hipGetLastError());
    return MECHA_OK;
}

}  // extern "C"
