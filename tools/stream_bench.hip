// stream_bench.hip -- standalone tuning harness for the streaming merge kernel shape (not part of the library).
// Variants of an N-input bf16 running-mean kernel: vectors per lane per iteration, non-temporal loads/stores,
// grid size, grid-stride vs contiguous chunk per block.  Build: hipcc -O3 --offload-arch=gfx950 -o stream_bench stream_bench.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

struct Ptrs { const uint16_t *in[8]; };

template <bool NT> __device__ __forceinline__ u32x4 ld(const uint16_t *p, int64_t i) {
    const u32x4 *q = reinterpret_cast<const u32x4 *>(p + i);
    if constexpr (NT) return __builtin_nontemporal_load(q);
    else return *q;
}
template <bool NT> __device__ __forceinline__ void st(uint16_t *p, int64_t i, u32x4 v) {
    u32x4 *q = reinterpret_cast<u32x4 *>(p + i);
    if constexpr (NT) __builtin_nontemporal_store(v, q);
    else *q = v;
}
__device__ __forceinline__ uint32_t f2bf(float f) { __bf16 h = (__bf16)f; return (uint32_t)__builtin_bit_cast(uint16_t, h); }

template <int N, int UNR, bool NT, bool CHUNK>
__global__ __launch_bounds__(256) void navg(Ptrs p, uint16_t *out, int64_t n_vec) {
    int64_t v0, v1, stride;
    if constexpr (CHUNK) {
        const int64_t per = (n_vec + gridDim.x - 1) / gridDim.x;
        v0 = (int64_t)blockIdx.x * per + threadIdx.x; v1 = (int64_t)(blockIdx.x + 1) * per; if (v1 > n_vec) v1 = n_vec; stride = 256;
    } else { v0 = (int64_t)blockIdx.x * 256 + threadIdx.x; v1 = n_vec; stride = (int64_t)gridDim.x * 256; }
    for (int64_t v = v0; v < v1; v += stride * UNR) {
        u32x4 r[UNR][N];
#pragma unroll
        for (int u = 0; u < UNR; u++)
#pragma unroll
            for (int m = 0; m < N; m++)
                if (v + u * stride < v1) r[u][m] = ld<NT>(p.in[m], (v + u * stride) * 8);
#pragma unroll
        for (int u = 0; u < UNR; u++) {
            if (v + u * stride >= v1) continue;
            float acc[8];
#pragma unroll
            for (int m = 0; m < N; m++) {
                uint32_t w[4] = {r[u][m].x, r[u][m].y, r[u][m].z, r[u][m].w};
                const float wt = 1.0f / (float)(m + 1);
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    float lo = __uint_as_float(w[j] << 16), hi = __uint_as_float(w[j] & 0xffff0000u);
                    if (m == 0) { acc[2 * j] = lo; acc[2 * j + 1] = hi; }
                    else { acc[2 * j] = __builtin_fmaf(wt, lo - acc[2 * j], acc[2 * j]); acc[2 * j + 1] = __builtin_fmaf(wt, hi - acc[2 * j + 1], acc[2 * j + 1]); }
                }
            }
            u32x4 o = {f2bf(acc[0]) | (f2bf(acc[1]) << 16), f2bf(acc[2]) | (f2bf(acc[3]) << 16), f2bf(acc[4]) | (f2bf(acc[5]) << 16), f2bf(acc[6]) | (f2bf(acc[7]) << 16)};
            st<NT>(out, (v + u * stride) * 8, o);
        }
    }
}

template <int N, int UNR, bool NT, bool CHUNK>
void run(const char *name, Ptrs p, uint16_t *out, int64_t n, int grid) {
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int64_t n_vec = n / 8;
    for (int i = 0; i < 2; i++) hipLaunchKernelGGL((navg<N, UNR, NT, CHUNK>), dim3(grid), dim3(256), 0, 0, p, out, n_vec);
    CHECK(hipEventRecord(e0, 0));
    const int iters = 10;
    for (int i = 0; i < iters; i++) hipLaunchKernelGGL((navg<N, UNR, NT, CHUNK>), dim3(grid), dim3(256), 0, 0, p, out, n_vec);
    CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= iters;
    printf("N=%d %-34s grid=%5d  %.3f ms  %.0f GB/s\n", N, name, grid, ms, (double)n * 2 * (N + 1) / ms / 1e6);
}

template <int N> void sweep(Ptrs p, uint16_t *out, int64_t n) {
    for (int grid : {2048, 4096, 8192}) {
        run<N, 1, false, false>("unr1 stride", p, out, n, grid);
        run<N, 2, false, false>("unr2 stride", p, out, n, grid);
        run<N, 1, true, false>("unr1 stride nt", p, out, n, grid);
        run<N, 2, true, false>("unr2 stride nt", p, out, n, grid);
        run<N, 1, false, true>("unr1 chunk", p, out, n, grid);
        run<N, 2, true, true>("unr2 chunk nt", p, out, n, grid);
    }
    run<N, 4, true, false>("unr4 stride nt", p, out, n, 2048);
}

int main() {
    const int64_t n = (int64_t)1 << 29;                          // 512 Mi elements = 1 GiB per bf16 array
    Ptrs p{};
    for (int m = 0; m < 8; m++) { void *d; CHECK(hipMalloc(&d, n * 2)); CHECK(hipMemset(d, 0x3c + m, n * 2)); p.in[m] = (const uint16_t *)d; }
    void *out; CHECK(hipMalloc(&out, n * 2));
    sweep<2>(p, (uint16_t *)out, n);
    sweep<4>(p, (uint16_t *)out, n);
    sweep<8>(p, (uint16_t *)out, n);
    return 0;
}
