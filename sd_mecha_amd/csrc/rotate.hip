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
// Kernels
//   rot_colsum / rot_means / rot_center : column means (fp64 accumulation) and the centred fp32 copies Ac, Bc
//   gemm_kernel   : C[m][n] = sum_k A(k, m) B(k, n) with arbitrary element strides, on v_mfma_f32_32x32x2_f32 (exact fp32
//                   fma chains); split-K slabs + an in-order reduce for skinny outputs (conv 3x3 keys: p = 9, n ~ 1.6 M);
//                   fused epilogue (minus / plus a matrix, per-column bias, cast to the output dtype, finite flag)
//   jacobi_step   : one-sided (Hestenes) Jacobi SVD.  The vectors being orthogonalised are stored as ROWS, so each is
//                   contiguous; a step rotates c/2 disjoint pairs in parallel (round-robin ordering), one workgroup per
//                   pair: both vectors are staged in registers, the three inner products reduced through LDS, and the
//                   pair of vectors and the matching pair of rows of W^T rotated.  A sweep is c - 1 steps; sweeps repeat
//                   until one makes no rotation (the host reads one counter per sweep).
//   rot_normalize : U = G / |column|
// The SVD's sign/order ambiguity cancels in U V^T; parity with the reference is <= 1e-3 relative Frobenius error
// (wide keys with centring: on the component the reference determines, DESIGN.md "rotate: reference indeterminacy").
// Roofline: MFMA (f32) for the GEMMs (2 n p^2 flop each), LDS/latency for the Jacobi sweeps (reported as sweeps + time).
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

__global__ __launch_bounds__(256) void rot_means_kernel(const double *part, int chunks, int64_t n, int64_t p, float cz,
                                                        float *a_mean, float *b_mean) {
    const int64_t col = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (col >= p) return;
    double sa = 0.0, sb = 0.0;
    for (int k = 0; k < chunks; k++) { sa += part[((int64_t)0 * chunks + k) * p + col]; sb += part[((int64_t)1 * chunks + k) * p + col]; }
    a_mean[col] = (float)(sa / (double)n) * cz;          // a_2d.mean(0) * centralization  (svd.py:67)
    b_mean[col] = (float)(sb / (double)n) * cz;
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
};

__device__ __forceinline__ void gemm_store(const GemmEpilogue &e, float *c, int64_t ldc, int64_t row, int64_t col, float v, bool &bad) {
    const int64_t idx = row * ldc + col;
    if (e.minus != nullptr) v = v - e.minus[idx];
    if (e.plus != nullptr) v = e.plus[idx] + v;
    if (e.col_bias != nullptr) v = v + e.col_bias[col];
    if (e.out != nullptr) {
        bad |= nonfinite_f32(v);
        if (e.out_dt == MECHA_BF16) store1<MECHA_BF16>(e.out, idx, v);
        else if (e.out_dt == MECHA_F16) store1<MECHA_F16>(e.out, idx, v);
        else store1<MECHA_F32>(e.out, idx, v);
    } else c[idx] = v;
}

__global__ __launch_bounds__(256) void gemm_kernel(Operand A, Operand B, float *C, int64_t M, int64_t N, int64_t K, int64_t k_per_slab,
                                                   GemmEpilogue e) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 31, kk = lane >> 5;
    const int64_t m0 = (int64_t)blockIdx.y * 64 + (wave >> 1) * 32, n0 = (int64_t)blockIdx.x * 64 + (wave & 1) * 32;
    if (m0 >= M || n0 >= N) return;
    const int64_t k_begin = (int64_t)blockIdx.z * k_per_slab;
    const int64_t k_end = (k_begin + k_per_slab < K) ? k_begin + k_per_slab : K;
    const bool a_ok = m0 + i < M, b_ok = n0 + i < N;
    const float *ap = A.p + (m0 + i) * A.ms, *bp = B.p + (n0 + i) * B.ms;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0.0f;
    int64_t k = k_begin;
    for (; k + 16 <= k_end; k += 16) {
        float av[8], bv[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int64_t kr = k + 2 * u + kk;
            av[u] = a_ok ? ap[kr * A.ks] : 0.0f;
            bv[u] = b_ok ? bp[kr * B.ks] : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < 8; u++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[u], acc, 0, 0, 0);
    }
    for (; k < k_end; k += 2) {
        const int64_t kr = k + kk;
        const float av = (a_ok && kr < k_end) ? ap[kr * A.ks] : 0.0f;
        const float bv = (b_ok && kr < k_end) ? bp[kr * B.ks] : 0.0f;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
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

// ------------------------------------------------------------------------------------------------ one-sided Jacobi
constexpr int JAC_MAX_REG = 32;       // floats of each vector a lane keeps in registers (length <= 8192)
constexpr int64_t JAC_MAX_LEN = (int64_t)JAC_MAX_REG * 256;

__global__ __launch_bounds__(256) void set_identity_kernel(float *w, int64_t c) {
    const int64_t total = c * c;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) w[i] = (i / c == i % c) ? 1.0f : 0.0f;
}

// Round-robin pairing of `players` (even) vectors at step s: player players-1 is fixed, the rest rotate.
__device__ __forceinline__ void jacobi_pair(int players, int s, int k, int &i, int &j) {
    const int m = players - 1;
    if (k == 0) { i = m; j = s % m; }
    else { i = (s + k) % m; j = (s - k + m) % m; }
    if (i > j) { const int t = i; i = j; j = t; }
}

__device__ __forceinline__ float block_sum(float v, float *sh) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}

// gt: c vectors of length m (rows), wt: c x c (row k = column k of W).  One block per pair.
__global__ __launch_bounds__(256) void jacobi_step_kernel(float *gt, float *wt, int c, int64_t m, int players, int step, float tol,
                                                          unsigned int *rotations) {
    int i, j;
    jacobi_pair(players, step, blockIdx.x, i, j);
    if (j >= c) return;                                         // pair with the padding player (odd c)
    __shared__ float sh[4];
    __shared__ float rot[2];
    float *gi = gt + (int64_t)i * m, *gj = gt + (int64_t)j * m;
    float xi[JAC_MAX_REG], xj[JAC_MAX_REG];
    float aa = 0.0f, bb = 0.0f, ab = 0.0f;
#pragma unroll
    for (int u = 0; u < JAC_MAX_REG; u++) {
        const int64_t t = (int64_t)u * 256 + threadIdx.x;
        xi[u] = (t < m) ? gi[t] : 0.0f;
        xj[u] = (t < m) ? gj[t] : 0.0f;
        aa = __builtin_fmaf(xi[u], xi[u], aa); bb = __builtin_fmaf(xj[u], xj[u], bb); ab = __builtin_fmaf(xi[u], xj[u], ab);
    }
    aa = block_sum(aa, sh); bb = block_sum(bb, sh); ab = block_sum(ab, sh);
    if (threadIdx.x == 0) {
        float cs = 1.0f, sn = 0.0f;
        if (__builtin_fabsf(ab) > tol * __builtin_sqrtf(aa * bb) && ab != 0.0f) {
            const float zeta = (bb - aa) / (2.0f * ab);
            const float t = ((zeta >= 0.0f) ? 1.0f : -1.0f) / (__builtin_fabsf(zeta) + __builtin_sqrtf(1.0f + zeta * zeta));
            cs = 1.0f / __builtin_sqrtf(1.0f + t * t);
            sn = cs * t;
            atomicAdd(rotations, 1u);
        }
        rot[0] = cs; rot[1] = sn;
    }
    __syncthreads();
    const float cs = rot[0], sn = rot[1];
    if (sn == 0.0f) return;
#pragma unroll
    for (int u = 0; u < JAC_MAX_REG; u++) {
        const int64_t t = (int64_t)u * 256 + threadIdx.x;
        if (t < m) { gi[t] = cs * xi[u] - sn * xj[u]; gj[t] = sn * xi[u] + cs * xj[u]; }
    }
    float *wi = wt + (int64_t)i * c, *wj = wt + (int64_t)j * c;
    for (int t = threadIdx.x; t < c; t += 256) {
        const float a = wi[t], b = wj[t];
        wi[t] = cs * a - sn * b; wj[t] = sn * a + cs * b;
    }
}

// each row of gt scaled to unit norm (U = G / sigma); an exactly-zero vector stays zero
__global__ __launch_bounds__(256) void rot_normalize_kernel(float *gt, int64_t m) {
    __shared__ float sh[4];
    float *g = gt + (int64_t)blockIdx.x * m;
    float s = 0.0f;
    for (int64_t t = threadIdx.x; t < m; t += 256) s = __builtin_fmaf(g[t], g[t], s);
    s = block_sum(s, sh);
    const float inv = (s > 0.0f) ? 1.0f / __builtin_sqrtf(s) : 0.0f;
    for (int64_t t = threadIdx.x; t < m; t += 256) g[t] = g[t] * inv;
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

// One-sided Jacobi on the c vectors (rows of gt, length m); wt = accumulated right rotations (as rows).
// Synchronises the stream once per sweep to read the rotation counter.
int jacobi_svd(float *gt, float *wt, int c, int64_t m, unsigned int *counter_dev, int max_sweeps, int *sweeps_out, hipStream_t s) {
    MECHA_REQUIRE(m <= JAC_MAX_LEN, "rotate: vector length %lld exceeds the Jacobi register window (%lld)", (long long)m, (long long)JAC_MAX_LEN);
    const unsigned ib = (unsigned)(((int64_t)c * c + 255) / 256 < 2048 ? ((int64_t)c * c + 255) / 256 : 2048);
    hipLaunchKernelGGL(set_identity_kernel, dim3(ib), dim3(256), 0, s, wt, (int64_t)c);
    const int players = (c + 1) / 2 * 2;
    const float tol = 1e-6f;
    int sweeps = 0;
    if (c >= 2) {
        while (sweeps < max_sweeps) {
            MECHA_CHECK_HIP(hipMemsetAsync(counter_dev, 0, sizeof(unsigned int), s));
            for (int step = 0; step < players - 1; step++)
                MECHA_LAUNCH(MK_ROT_JACOBI, jacobi_step_kernel, dim3(players / 2), dim3(256), 0, s, gt, wt, c, m, players, step, tol, counter_dev);
            unsigned int rotations = 0;
            MECHA_CHECK_HIP(hipMemcpyAsync(&rotations, counter_dev, sizeof(unsigned int), hipMemcpyDeviceToHost, s));
            MECHA_CHECK_HIP(hipStreamSynchronize(s));
            sweeps++;
            if (rotations == 0) break;
        }
    }
    if (sweeps_out) *sweeps_out += sweeps;
    hipLaunchKernelGGL(rot_normalize_kernel, dim3(c), dim3(256), 0, s, gt, m);
    return MECHA_OK;
}

size_t rotate_workspace_bytes_impl(int64_t n, int64_t p) {
    const size_t q = (size_t)(n < p ? n : p);                   // number of vectors the Jacobi runs on
    size_t b = 4096;
    b += up256(sizeof(double) * 2 * 64 * p) + 2 * up256(sizeof(float) * p);
    b += 2 * up256(sizeof(float) * n * p);                      // Ac, Bc
    b += up256(sizeof(float) * SLAB_FLOATS);
    if (n >= p) b += 3 * up256(sizeof(float) * p * p);          // G^T, W^T, R
    else b += 3 * up256(sizeof(float) * q * p) + 5 * up256(sizeof(float) * q * q);   // Qb^T, U^T, V^T; W1^T, W2^T, P, xV, D
    return b;
}

}  // namespace

extern "C" {

size_t mecha_rotate_workspace_bytes(int64_t n_rows, int64_t n_cols) {
    if (n_rows < 1) n_rows = 1;
    if (n_cols < 1) n_cols = 1;
    return rotate_workspace_bytes_impl(n_rows, n_cols);
}

int mecha_rotate(const void *a, const void *b, void *out, int64_t n, int64_t p, int in_dtype, int out_dtype, float centralization,
                 void *workspace, size_t workspace_bytes, int *sweeps_out, uint32_t *flags, void *stream) {
    MECHA_REQUIRE(a != nullptr && b != nullptr && out != nullptr, "NULL tensor pointer");
    MECHA_REQUIRE(n >= 1 && p >= 1, "rotate needs a non-empty 2-D key (got %lld x %lld)", (long long)n, (long long)p);
    MECHA_REQUIRE(mecha_valid_dtype(in_dtype) && mecha_valid_dtype(out_dtype), "unknown dtype code");
    MECHA_REQUIRE(p <= JAC_MAX_LEN, "rotate: %lld columns exceed the Jacobi register window (%lld)", (long long)p, (long long)JAC_MAX_LEN);
    const size_t need = rotate_workspace_bytes_impl(n, p);
    MECHA_REQUIRE(workspace != nullptr && workspace_bytes >= need && mecha_aligned16(workspace),
                  "rotate workspace too small or unaligned: need %zu bytes, got %zu", need, workspace_bytes);
    hipStream_t s = static_cast<hipStream_t>(stream);
    Bump ws{static_cast<char *>(workspace), 0};
    const int chunks = (int)((n + 255) / 256 < 64 ? (n + 255) / 256 : 64);
    unsigned int *counter = ws.take<unsigned int>(64);
    double *part = ws.take<double>((size_t)2 * 64 * p);
    float *a_mean = ws.take<float>(p), *b_mean = ws.take<float>(p);
    float *ac = ws.take<float>(n * p), *bc = ws.take<float>(n * p);
    float *slabs = ws.take<float>(SLAB_FLOATS);
    if (sweeps_out) *sweeps_out = 0;

    hipLaunchKernelGGL(rot_colsum_kernel, dim3((unsigned)((p + 63) / 64), (unsigned)chunks), dim3(256), 0, s, a, b, in_dtype, n, p, chunks, part, flags);
    hipLaunchKernelGGL(rot_means_kernel, dim3((unsigned)((p + 255) / 256)), dim3(256), 0, s, part, chunks, n, p, centralization, a_mean, b_mean);
    const int64_t total = n * p;
    const unsigned cb = (unsigned)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(rot_center_kernel, dim3(cb), dim3(256), 0, s, a, b, in_dtype, n, p, a_mean, b_mean, ac, bc);

    const GemmEpilogue none{};
    GemmEpilogue fin{};
    fin.col_bias = b_mean; fin.out = out; fin.out_dt = out_dtype; fin.flags = flags;
    int rc;
    if (n >= p) {
        float *gt = ws.take<float>(p * p), *wt = ws.take<float>(p * p), *R = ws.take<float>(p * p);
        // G^T = M^T:  G^T[j][i] = sum_r Bc[r][j] Ac[r][i]   (row j of G^T = column j of M = Ac^T Bc)
        gemm(Operand{bc, p, 1}, Operand{ac, p, 1}, gt, p, p, n, slabs, none, s);
        if ((rc = jacobi_svd(gt, wt, (int)p, p, counter, 30, sweeps_out, s)) != MECHA_OK) return rc;     // M W = U S; gt <- U^T, wt = W^T
        // R = U W^T:  R[i][j] = sum_k U^T[k][i] W^T[k][j]
        gemm(Operand{gt, p, 1}, Operand{wt, p, 1}, R, p, p, p, slabs, none, s);
        // out[r][j] = sum_i Ac[r][i] R[i][j] + b_mean[j]
        gemm(Operand{ac, 1, p}, Operand{R, p, 1}, nullptr, n, p, p, slabs, fin, s);
    } else {
        float *qbt = ws.take<float>(n * p), *ut = ws.take<float>(n * p), *vt = ws.take<float>(n * p);
        float *w1 = ws.take<float>(n * n), *w2 = ws.take<float>(n * n), *P = ws.take<float>(n * n), *xv = ws.take<float>(n * n), *D = ws.take<float>(n * n);
        // orthonormal basis of the row space of Bc: orthogonalise the n rows of Bc (= columns of b.mH)
        MECHA_CHECK_HIP(hipMemcpyAsync(qbt, bc, sizeof(float) * n * p, hipMemcpyDeviceToDevice, s));
        if ((rc = jacobi_svd(qbt, w1, (int)n, p, counter, 30, sweeps_out, s)) != MECHA_OK) return rc;     // qbt <- Qb^T
        // T^T = Bc Qb =: P (n x n):  P[r][k] = sum_c Bc[r][c] Qb^T[k][c]
        gemm(Operand{bc, 1, p}, Operand{qbt, 1, p}, P, n, n, p, slabs, none, s);
        // N^T = P^T Ac (n vectors of length p):  N^T[k][i] = sum_r P[r][k] Ac[r][i]
        gemm(Operand{P, n, 1}, Operand{ac, p, 1}, ut, n, p, n, slabs, none, s);
        if ((rc = jacobi_svd(ut, w2, (int)n, p, counter, 30, sweeps_out, s)) != MECHA_OK) return rc;      // N W2 = U S; ut <- U^T, w2 = W2^T
        // V^T = W2^T Qb^T:  V^T[k][c] = sum_j W2^T[k][j] Qb^T[j][c]
        gemm(Operand{w2, 1, n}, Operand{qbt, p, 1}, vt, n, p, n, slabs, none, s);
        // xV[r][k] = sum_c Ac[r][c] V^T[k][c];  D = xU - xV
        gemm(Operand{ac, 1, p}, Operand{vt, 1, p}, xv, n, n, p, slabs, none, s);
        GemmEpilogue minus_xv{};
        minus_xv.minus = xv;
        gemm(Operand{ac, 1, p}, Operand{ut, 1, p}, D, n, n, p, slabs, minus_xv, s);
        // out[r][c] = Ac[r][c] + sum_k D[r][k] V^T[k][c] + b_mean[c]
        fin.plus = ac;
        gemm(Operand{D, 1, n}, Operand{vt, p, 1}, nullptr, n, p, n, slabs, fin, s);
    }
    MECHA_CHECK_HIP(hipGetLastError());
    return MECHA_OK;
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
