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
#include <type_traits>
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// ------------------------------------------------------------------------------------------------ centring
__device__ __forceinline__ float ld_any(int dt, const void *p, int64_t i) {
    return (dt == MECHA_BF16) ? load1<MECHA_BF16>(p, i) : (dt == MECHA_F16) ? load1<MECHA_F16>(p, i) : load1<MECHA_F32>(p, i);
}

// partial column sums: grid (ceil(p/64), chunks); block 256 = 4 row lanes x 64 columns
__global__ __launch_bounds__(256) void rot_colsum_kernel(const void *a, const void *b, int dt, int64_t n, int64_t p, int chunks,
                                                         double *part /* [2][chunks][p] */, uint32_t *flags) {
    const int c = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int64_t col = (int64_t)blockIdx.x * 64 + c;
    const int64_t rows_per = (n + chunks - 1) / chunks;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per, r1 = (r0 + rows_per < n) ? r0 + rows_per : n;
    double sa = 0.0, sb = 0.0;
    bool bad = false;
    if (col < p) {
        for (int64_t r = r0 + rl; r < r1; r += 4) {
            const float x = ld_any(dt, a, r * p + col), y = ld_any(dt, b, r * p + col);
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

__global__ __launch_bounds__(256) void rot_means_kernel(const double *part, int chunks, int64_t n, int64_t p, float cz, float t,
                                                        float *a_mean, float *b_mean, float *t_mean) {
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
                                                         const float *b_mean, float *ac, float *bc) {
    const int64_t total = n * p;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t c = i % p;
        ac[i] = ld_any(dt, a, i) - a_mean[c];             // a_2d - a_mean  (svd.py:69-70)
        bc[i] = ld_any(dt, b, i) - b_mean[c];
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
};

__device__ __forceinline__ void gemm_store(const GemmEpilogue &e, float *c, int64_t ldc, int64_t row, int64_t col, float v, bool &bad) {
    const int64_t idx = row * ldc + col;
    if (e.axpy != nullptr) v = __builtin_fmaf(e.axpy_a, e.axpy[idx], e.axpy_b * v);
    if (e.minus != nullptr) v = v - e.minus[idx];
    if (e.plus != nullptr) v = e.plus[idx] + v;
    if (e.mix != nullptr) v = __builtin_fmaf(e.mix_w0, e.mix[idx], e.mix_w1 * v);
    if (e.col_bias != nullptr) v = v + e.col_bias[col];
    if (e.out != nullptr) {
        bad |= nonfinite_f32(v);
        if (e.out_dt == MECHA_BF16) store1<MECHA_BF16>(e.out, idx, v);
        else if (e.out_dt == MECHA_F16) store1<MECHA_F16>(e.out, idx, v);
        else store1<MECHA_F32>(e.out, idx, v);
    } else c[idx] = v;
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

__global__ __launch_bounds__(256) void gemm_kernel(Operand A, Operand B, float *C, int64_t M, int64_t N, int64_t K, int64_t k_per_slab,
                                                   GemmEpilogue e) {
    __shared__ __attribute__((aligned(16))) float lds[2][2][GK * GP];       // [buffer][A | B]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 31, kk = lane >> 5;
    const int64_t bm0 = (int64_t)blockIdx.y * 64, bn0 = (int64_t)blockIdx.x * 64;
    const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
    const int64_t m0 = bm0 + wm, n0 = bn0 + wn;
    const int64_t k_begin = (int64_t)blockIdx.z * k_per_slab;
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
    float *c = C + (int64_t)blockIdx.z * M * N;
    GemmEpilogue ee = e;
    if (gridDim.z > 1) ee = GemmEpilogue{};                     // the epilogue then runs in gemm_reduce
    bool bad = false;
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const int64_t row = m0 + (r & 3) + 8 * (r >> 2) + 4 * kk, col = n0 + i;
        if (row < M && col < N) gemm_store(ee, c, N, row, col, acc[r], bad);
    }
    if (bad && e.flags != nullptr) atomicOr(e.flags, MECHA_FLAG_OUTPUT_NONFINITE);
}

// The same product for outputs of at least 256 x 256: block tile 128 x 128, wave tile 64 x 64 = 2 x 2 accumulators.  Every LDS
// read feeds two MFMAs and a wave has four independent accumulator chains, so one or two blocks per CU keep the matrix
// pipes fed (the 64 x 64 kernel: one dependent chain per wave, 400 small blocks for a 1280 x 1280 product).  16-deep
// k-chunks: 2 operands x 2 buffers x 16 x 132 floats = 33 KiB of LDS.
constexpr int GK2 = 16;
constexpr int GP2 = 132;               // LDS row pitch in floats (528 B: 16-byte aligned rows, rows 4 banks apart)

__device__ __forceinline__ TileRegs tile_load128(const Operand &op, int64_t m0, int64_t M, int64_t k0, int64_t k_end) {
    TileRegs t;
    const int tid = threadIdx.x;
    if (op.ms == 1) {                  // m-contiguous: thread = (k = tid / 16, 4 consecutive m at 4 * (tid % 16) [+64])
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
    } else if (op.ks == 1) {           // k-contiguous: thread = (m = tid / 4 [+64], 4 consecutive k at 4 * (tid % 4)): 4 lanes per row
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
__device__ __forceinline__ void tile_store128(const Operand &op, const TileRegs &t, float *lds) {
    const int tid = threadIdx.x;
    if (op.ms == 1) {
#pragma unroll
        for (int h = 0; h < 2; h++) *reinterpret_cast<f32x4 *>(&lds[(tid >> 4) * GP2 + (tid & 15) * 4 + 64 * h]) = t.v[h];
    } else if (op.ks == 1) {
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

__global__ __launch_bounds__(256) void gemm128_kernel(Operand A, Operand B, float *C, int64_t M, int64_t N, int64_t K, int64_t k_per_slab,
                                                      GemmEpilogue e) {
    __shared__ __attribute__((aligned(16))) float lds[2][2][GK2 * GP2];     // [buffer][A | B]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 31, kk = lane >> 5;
    const int64_t bm0 = (int64_t)blockIdx.y * 128, bn0 = (int64_t)blockIdx.x * 128;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    const int64_t k_begin = (int64_t)blockIdx.z * k_per_slab;
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
    TileRegs ra = tile_load128(A, bm0, M, k_begin, k_end), rb = tile_load128(B, bn0, N, k_begin, k_end);
    tile_store128(A, ra, lds[0][0]);
    tile_store128(B, rb, lds[0][1]);
    __syncthreads();
    int buf = 0;
    for (int64_t k = k_begin; k < k_end; k += GK2) {
        const bool more = k + GK2 < k_end;
        if (more) { ra = tile_load128(A, bm0, M, k + GK2, k_end); rb = tile_load128(B, bn0, N, k + GK2, k_end); }
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
        if (more) { tile_store128(A, ra, lds[buf ^ 1][0]); tile_store128(B, rb, lds[buf ^ 1][1]); }
        __syncthreads();
        buf ^= 1;
    }
    float *c = C + (int64_t)blockIdx.z * M * N;
    GemmEpilogue ee = e;
    if (gridDim.z > 1) ee = GemmEpilogue{};                     // the epilogue then runs in gemm_reduce
    bool bad = false;
#pragma unroll
    for (int tm = 0; tm < 2; tm++)
#pragma unroll
        for (int tn = 0; tn < 2; tn++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int64_t row = bm0 + wm + 32 * tm + (r & 3) + 8 * (r >> 2) + 4 * kk, col = bn0 + wn + 32 * tn + i;
                if (row < M && col < N) gemm_store(ee, c, N, row, col, acc[tm][tn][r], bad);
            }
    if (bad && e.flags != nullptr) atomicOr(e.flags, MECHA_FLAG_OUTPUT_NONFINITE);
}

__global__ __launch_bounds__(256) void gemm_reduce_kernel(const float *slabs, int n_slabs, float *C, int64_t M, int64_t N, GemmEpilogue e) {
    const int64_t total = M * N;
    bool bad = false;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const int64_t row = t / N, col = t - row * N;
        float s = 0.0f;
        for (int z = 0; z < n_slabs; z++) s += slabs[(int64_t)z * total + t];
        gemm_store(e, C, N, row, col, s, bad);
    }
    if (bad && e.flags != nullptr) atomicOr(e.flags, MECHA_FLAG_OUTPUT_NONFINITE);
}

// ------------------------------------------------------------------------------------------------ polar factor
// scal[0] += sum G^2, scal[1] += sum (G - I)^2 over the r x r Gram matrix G (fp64 accumulation; scal zeroed by the host)
__global__ __launch_bounds__(256) void rot_gram_norms_kernel(const float *g, int64_t r, double *scal) {
    __shared__ double sh[2][4];
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
__global__ __launch_bounds__(256) void rot_allclose_kernel(const void *a, const void *b, int dt, int64_t n, unsigned int *not_close) {
    bool bad = false;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float x = (float)(_Float16)ld_any(dt, a, i), y = (float)(_Float16)ld_any(dt, b, i);
        const bool close = (x == y) || (__builtin_fabsf(x - y) <= 1e-8f + 1e-5f * __builtin_fabsf(y));
        bad |= !close;
    }
    if (bad) atomicOr(not_close, 1u);
}

// ------------------------------------------------------------------------------------------------ host
size_t up256(size_t v) { return (v + 255) / 256 * 256; }

struct Bump {
    char *base; size_t off;
    template <typename T> T *take(size_t count) {
        T *p = reinterpret_cast<T *>(base + off);
        off = up256(off + count * sizeof(T));
        return p;
    }
};

constexpr size_t SLAB_FLOATS = (size_t)4096 * 64 * 64;          // split-K only for outputs of at most one 64 x 64 tile... see below

int split_k_slabs(int64_t M, int64_t N, int64_t K) {
    const int64_t tiles = ((M + 63) / 64) * ((N + 63) / 64);
    int64_t z = (1024 + tiles - 1) / tiles;
    const int64_t zmax = (K + 255) / 256;
    if (z > zmax) z = zmax;
    while (z > 1 && (size_t)z * (size_t)M * (size_t)N > SLAB_FLOATS) z--;
    if (z < 1) z = 1;
    return (int)z;
}

void gemm(Operand A, Operand B, float *C, int64_t M, int64_t N, int64_t K, float *slabs, const GemmEpilogue &e, hipStream_t s) {
    if (M >= 256 && N >= 256) {
        // 128 x 128 tiles; split K so that about two blocks per CU are resident (512 on 256 CUs; 256 and 1024 measured slower)
        const int64_t tiles = ((M + 127) / 128) * ((N + 127) / 128);
        int64_t z = 512 / tiles;
        const int64_t zmax = (K + 127) / 128;
        if (z > zmax) z = zmax;
        while (z > 1 && (size_t)z * (size_t)M * (size_t)N > SLAB_FLOATS) z--;
        if (z < 1) z = 1;
        int64_t k_per = (K + z - 1) / z;
        k_per = (k_per + GK2 - 1) / GK2 * GK2;                  // whole k-chunks per slab (keeps 16-byte loads along k aligned)
        z = (K + k_per - 1) / k_per;
        dim3 grid((unsigned)((N + 127) / 128), (unsigned)((M + 127) / 128), (unsigned)z), block(256);
        if (z == 1) {
            MECHA_LAUNCH(MK_ROT_GEMM, gemm128_kernel, grid, block, 0, s, A, B, C, M, N, K, k_per, e);
        } else {
            GemmEpilogue none{};
            MECHA_LAUNCH(MK_ROT_GEMM, gemm128_kernel, grid, block, 0, s, A, B, slabs, M, N, K, k_per, none);
            const int64_t total = M * N;
            const unsigned rb = (unsigned)((total + 255) / 256 < 1024 ? (total + 255) / 256 : 1024);
            MECHA_LAUNCH(MK_ROT_GEMM, gemm_reduce_kernel, dim3(rb), dim3(256), 0, s, slabs, (int)z, C, M, N, e);
        }
        return;
    }
    const int z = split_k_slabs(M, N, K);
    int64_t k_per = (K + z - 1) / z;
    k_per = (k_per + 1) / 2 * 2;                                // the MFMA consumes k in pairs
    dim3 grid((unsigned)((N + 63) / 64), (unsigned)((M + 63) / 64), (unsigned)z), block(256);
    if (z == 1) {
        MECHA_LAUNCH(MK_ROT_GEMM, gemm_kernel, grid, block, 0, s, A, B, C, M, N, K, k_per, e);
    } else {
        GemmEpilogue none{};
        MECHA_LAUNCH(MK_ROT_GEMM, gemm_kernel, grid, block, 0, s, A, B, slabs, M, N, K, k_per, none);
        const int64_t total = M * N;
        const unsigned rb = (unsigned)((total + 255) / 256 < 1024 ? (total + 255) / 256 : 1024);
        MECHA_LAUNCH(MK_ROT_GEMM, gemm_reduce_kernel, dim3(rb), dim3(256), 0, s, slabs, z, C, M, N, e);
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
const PolarSchedule &polar_schedule() {
    static const PolarSchedule sched = [] {
        PolarSchedule ps{};
        double l = 1e-7, u = 1.0;
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
    }();
    return sched;
}

// X *= s, G *= s^2 with s = (sum G^2)^(-1/4) read from scal[0] on the device (0 for a zero matrix: it stays zero)
__global__ __launch_bounds__(256) void rot_polar_scale_kernel(float *x, int64_t nx, float *g, int64_t ng, const double *scal) {
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
__global__ __launch_bounds__(256) void rot_deflate_ones_kernel(float *x, int64_t r, int64_t c) {
    __shared__ double part[4][64];
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

int polar_newton_schulz(float *X, float *tmp, float *G, float *G2, int64_t r, int64_t c, double *scal, float *slabs, int *iters_out, hipStream_t s,
                        bool deflate_ones = false) {
    float *cur = X, *nxt = tmp;
    const int64_t rr = r * r;
    const unsigned nb = (unsigned)((rr + 255) / 256 < 1024 ? (rr + 255) / 256 : 1024);
    const GemmEpilogue none{};
    const PolarSchedule &ps = polar_schedule();
    for (int it = 0; it < ps.n; it++) {
        // G[m][n] = sum_k cur[m][k] cur[n][k]
        gemm(Operand{cur, 1, c}, Operand{cur, 1, c}, G, r, r, c, slabs, none, s);
        if (it == 0) {
            MECHA_CHECK_HIP(hipMemsetAsync(scal, 0, 2 * sizeof(double), s));
            MECHA_LAUNCH(MK_ROT_NORMS, rot_gram_norms_kernel, dim3(nb), dim3(256), 0, s, G, r, scal);
            const int64_t both = r * c + rr;
            const unsigned sb = (unsigned)((both + 255) / 256 < 2048 ? (both + 255) / 256 : 2048);
            MECHA_LAUNCH(MK_ROT_NORMS, rot_polar_scale_kernel, dim3(sb), dim3(256), 0, s, cur, r * c, G, rr, scal);
        }
        // G2 = b G + c G G;  nxt = a cur + G2 cur
        GemmEpilogue mix{};
        mix.axpy = G; mix.axpy_a = ps.step[it].b; mix.axpy_b = ps.step[it].c;
        // (G and b G + c G G are symmetric bit for bit -- the same products in the same order on both sides of the diagonal --
        //  so both operands are read along their rows: contiguous 16-byte loads)
        gemm(Operand{G, r, 1}, Operand{G, r, 1}, G2, r, r, r, slabs, mix, s);
        GemmEpilogue upd{};
        upd.axpy = cur; upd.axpy_a = ps.step[it].a; upd.axpy_b = 1.0f;
        gemm(Operand{G2, r, 1}, Operand{cur, c, 1}, nxt, r, c, r, slabs, upd, s);
        float *t = cur; cur = nxt; nxt = t;
        if (deflate_ones && ((it + 1) % POLAR_DEFLATE_EVERY == 0 || it + 1 == ps.n))
            MECHA_LAUNCH(MK_ROT_NORMS, rot_deflate_ones_kernel, dim3((unsigned)((c + 63) / 64)), dim3(256), 0, s, cur, r, c);
    }
    if (iters_out) *iters_out += ps.n;
    if (cur != X) MECHA_CHECK_HIP(hipMemcpyAsync(X, cur, sizeof(float) * r * c, hipMemcpyDeviceToDevice, s));
    return MECHA_OK;
}

size_t rotate_workspace_bytes_impl(int64_t n, int64_t p) {
    const size_t q = (size_t)(n < p ? n : p);
    size_t b = 4096;
    b += up256(sizeof(double) * 2 * 64 * p) + 3 * up256(sizeof(float) * p);
    b += 3 * up256(sizeof(float) * n * p);                      // Ac, Bc, and the alpha-interpolated copy
    b += up256(sizeof(float) * SLAB_FLOATS);
    if (n >= p) b += 4 * up256(sizeof(float) * p * p);          // M -> R, its Newton-Schulz twin, G, b G + c G^2
    else b += 3 * up256(sizeof(float) * q * p) + 5 * up256(sizeof(float) * q * q);   // Qb^T, Z, twin; G, b G + c G^2, P, x P2^T, D
    return b;
}

}  // namespace

extern "C" {

size_t mecha_rotate_workspace_bytes(int64_t n_rows, int64_t n_cols) {
    if (n_rows < 1) n_rows = 1;
    if (n_cols < 1) n_cols = 1;
    return rotate_workspace_bytes_impl(n_rows, n_cols);
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
    const bool has_alpha = alpha != 0.0f;
    MECHA_REQUIRE(alignment != 0 || has_alpha, "rotate: alignment == 0 with alpha == 0 is the identity (svd.py:39-40); return a instead");
    const size_t need = rotate_workspace_bytes_impl(n, p);
    MECHA_REQUIRE(workspace != nullptr && workspace_bytes >= need && mecha_aligned16(workspace),
                  "rotate workspace too small or unaligned: need %zu bytes, got %zu", need, workspace_bytes);
    MECHA_REQUIRE(transform_in == nullptr || mecha_aligned16(transform_in), "rotate: transform_in must be 16-byte aligned");
    hipStream_t s = static_cast<hipStream_t>(stream);
    Bump ws{static_cast<char *>(workspace), 0};
    const int chunks = (int)((n + 255) / 256 < 64 ? (n + 255) / 256 : 64);
    double *scal = ws.take<double>(8);
    double *part = ws.take<double>((size_t)2 * 64 * p);
    float *a_mean = ws.take<float>(p), *b_mean = ws.take<float>(p), *t_mean = ws.take<float>(p);
    float *ac = ws.take<float>(n * p), *bc = ws.take<float>(n * p), *y = ws.take<float>(n * p);
    float *slabs = ws.take<float>(SLAB_FLOATS);
    if (sweeps_out) *sweeps_out = 0;

    hipLaunchKernelGGL(rot_colsum_kernel, dim3((unsigned)((p + 63) / 64), (unsigned)chunks), dim3(256), 0, s, a, b, in_dtype, n, p, chunks, part, flags);
    hipLaunchKernelGGL(rot_means_kernel, dim3((unsigned)((p + 255) / 256)), dim3(256), 0, s, part, chunks, n, p, centralization,
                       (float)alignment, a_mean, b_mean, t_mean);
    const int64_t total = n * p;
    const unsigned cb = (unsigned)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(rot_center_kernel, dim3(cb), dim3(256), 0, s, a, b, in_dtype, n, p, a_mean, b_mean, ac, bc);

    const GemmEpilogue none{};
    GemmEpilogue fin{};                                         // + lerp(a_mean, b_mean, alignment), cast, finite flag  (svd.py:85-86)
    fin.col_bias = t_mean; fin.out = out; fin.out_dt = out_dtype; fin.flags = flags;
    int rc;
    if (n >= p) {
        float *R = ws.take<float>(p * p), *twin = ws.take<float>(p * p), *G = ws.take<float>(p * p), *G2 = ws.take<float>(p * p);
        if (transform_in != nullptr) {
            MECHA_CHECK_HIP(hipMemcpyAsync(R, transform_in, sizeof(float) * p * p, hipMemcpyDeviceToDevice, s));
        } else {
            // M = Ac^T Bc:  M[i][j] = sum_r Ac[r][i] Bc[r][j];  R = U V^T = the orthogonal polar factor of M
            gemm(Operand{ac, p, 1}, Operand{bc, p, 1}, R, p, p, n, slabs, none, s);
            if ((rc = polar_newton_schulz(R, twin, G, G2, p, p, scal, slabs, sweeps_out, s)) != MECHA_OK) return rc;
        }
        if (transform_out != nullptr) MECHA_CHECK_HIP(hipMemcpyAsync(transform_out, R, sizeof(float) * p * p, hipMemcpyDeviceToDevice, s));
        const float *src = ac;
        if (has_alpha) {
            // lerp(Ac, Bc R^T, alpha)  (svd.py:81-82; transform(b_2d, -1) = b_2d @ R.mH):  (Bc R^T)[r][i] = sum_j Bc[r][j] R[i][j]
            GemmEpilogue e = alignment == 0 ? fin : none;
            e.axpy = ac; e.axpy_a = 1.0f - alpha; e.axpy_b = alpha;
            gemm(Operand{bc, 1, p}, Operand{R, 1, p}, y, n, p, p, slabs, e, s);
            src = y;
        }
        // out[r][j] = sum_i src[r][i] R[i][j] + mean[j]   (alignment 1),  sum_i src[r][i] R[j][i] + mean[j]   (alignment -1)
        if (alignment == 1) gemm(Operand{src, 1, p}, Operand{R, p, 1}, nullptr, n, p, p, slabs, fin, s);
        else if (alignment == -1) gemm(Operand{src, 1, p}, Operand{R, 1, p}, nullptr, n, p, p, slabs, fin, s);
    } else {
        float *qbt = ws.take<float>(n * p), *z = ws.take<float>(n * p), *twin = ws.take<float>(n * p);
        float *G = ws.take<float>(n * n), *G2 = ws.take<float>(n * n), *P = ws.take<float>(n * n), *xq = ws.take<float>(n * n), *D = ws.take<float>(n * n);
        if (transform_in != nullptr) {
            MECHA_CHECK_HIP(hipMemcpyAsync(qbt, transform_in, sizeof(float) * n * p, hipMemcpyDeviceToDevice, s));
            MECHA_CHECK_HIP(hipMemcpyAsync(z, transform_in + n * p, sizeof(float) * n * p, hipMemcpyDeviceToDevice, s));
        } else {
            // Qb^T = polar factor of Bc: orthonormal rows spanning the row space of Bc (= columns of b.mH)
            MECHA_CHECK_HIP(hipMemcpyAsync(qbt, bc, sizeof(float) * n * p, hipMemcpyDeviceToDevice, s));
            // fully centred (the default): both wide-case matrices have the all-ones vector as their exact left null vector
            const bool deflate = centralization == 1.0f;
            if ((rc = polar_newton_schulz(qbt, twin, G, G2, n, p, scal, slabs, sweeps_out, s, deflate)) != MECHA_OK) return rc;
            // T^T = Bc Qb =: P (n x n):  P[r][k] = sum_c Bc[r][c] Qb^T[k][c]
            gemm(Operand{bc, 1, p}, Operand{qbt, 1, p}, P, n, n, p, slabs, none, s);
            // N^T = P^T Ac (n x p):  N^T[k][i] = sum_r P[r][k] Ac[r][i];  N = U S W^T, Z := polar(N^T) = W U^T
            gemm(Operand{P, n, 1}, Operand{ac, p, 1}, z, n, p, n, slabs, none, s);
            if ((rc = polar_newton_schulz(z, twin, G, G2, n, p, scal, slabs, sweeps_out, s, deflate)) != MECHA_OK) return rc;
        }
        if (transform_out != nullptr) {
            MECHA_CHECK_HIP(hipMemcpyAsync(transform_out, qbt, sizeof(float) * n * p, hipMemcpyDeviceToDevice, s));
            MECHA_CHECK_HIP(hipMemcpyAsync(transform_out + n * p, z, sizeof(float) * n * p, hipMemcpyDeviceToDevice, s));
        }
        // With V = Qb W (the reference's v) and U (its u):  U V^T = Z^T Qb^T,  V V^T = Qb Qb^T,  U U^T = Z^T Z,  V U^T = Qb Z.  So
        //   transform(x, +1) = x - x V V^T + x U V^T = x + (x Z^T - x Qb) Qb^T      (svd.py:181)   -> low_rank(x, Z, Qb^T)
        //   transform(x, -1) = x - x U U^T + x V U^T = x + (x Qb - x Z^T) Z         (svd.py:183)   -> low_rank(x, Qb^T, Z)
        auto low_rank = [&](const float *x, const float *P1, const float *P2, float *C, GemmEpilogue e) {
            gemm(Operand{x, 1, p}, Operand{P2, 1, p}, xq, n, n, p, slabs, none, s);         // xq[r][k] = sum_c x[r][c] P2[k][c]
            GemmEpilogue minus_xq{};
            minus_xq.minus = xq;
            gemm(Operand{x, 1, p}, Operand{P1, 1, p}, D, n, n, p, slabs, minus_xq, s);      // D = x P1^T - x P2^T
            e.plus = x;
            gemm(Operand{D, 1, n}, Operand{P2, p, 1}, C, n, p, n, slabs, e, s);             // C[r][c] = x[r][c] + sum_k D[r][k] P2[k][c]
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
    MECHA_CHECK_HIP(hipGetLastError());
    return MECHA_OK;
}

int mecha_rotate(const void *a, const void *b, void *out, int64_t n, int64_t p, int in_dtype, int out_dtype, float centralization,
                 void *workspace, size_t workspace_bytes, int *sweeps_out, uint32_t *flags, void *stream) {
    return mecha_rotate_ex(a, b, out, n, p, in_dtype, out_dtype, centralization, 1, 0.0f, nullptr, nullptr, workspace, workspace_bytes,
                           sweeps_out, flags, stream);
}

// torch.allclose(a.half(), b.half()) of svd.py:46.  ORs 1 into *not_close (device word, zeroed by the caller) if any element differs.
int mecha_allclose_half(const void *a, const void *b, int64_t numel, int in_dtype, uint32_t *not_close, void *stream) {
    MECHA_REQUIRE(numel >= 0 && mecha_valid_dtype(in_dtype) && not_close != nullptr, "bad arguments");
    if (numel == 0) return MECHA_OK;
    MECHA_REQUIRE(a != nullptr && b != nullptr, "NULL tensor pointer");
    const unsigned blocks = (unsigned)((numel + 255) / 256 < 2048 ? (numel + 255) / 256 : 2048);
    hipLaunchKernelGGL(rot_allclose_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), a, b, in_dtype, numel, not_close);
    MECHA_CHECK_HIP(hipGetLastError());
    return MECHA_OK;
}

}  // extern "C"
