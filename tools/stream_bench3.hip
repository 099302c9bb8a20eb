// stream_bench3.hip -- why do read+write streams flatten at 5.0-5.3 TB/s here when MI355X_MICROARCH.md records 6.29 TB/s for a
// float4 copy?  (round 1's tools/stream_bench2.hip never exceeded 5.29.)  Things stream_bench2 did not vary:
//   * the plain "one float4 per thread, one block per 256 vectors" grid (no loop at all);
//   * the footprint (256 MiB .. 6.9 GB);
//   * the RELATIVE PLACEMENT of the streams: every arena comes from its own allocation with the same (2 MiB) alignment and a
//     wave touches the same offset of every arena at the same time, i.e. addresses that differ only in high bits -- the
//     same HBM channel and bank for all streams.  `skew` shifts stream m by m * skew bytes;
//   * hipMemcpyAsync device-to-device as the runtime's own answer.
// Not part of the library.  Build: hipcc -O3 --offload-arch=gfx950 -o tools/bin/stream_bench3 tools/stream_bench3.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

struct Ptrs { const u32x4 *in[8]; };

template <bool NT> __device__ __forceinline__ u32x4 ld(const u32x4 *p) { if constexpr (NT) return __builtin_nontemporal_load(p); else return *p; }
template <bool NT> __device__ __forceinline__ void st(u32x4 *p, u32x4 v) { if constexpr (NT) __builtin_nontemporal_store(v, p); else *p = v; }

// one vector per thread, no loop
template <int N, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void flat(Ptrs p, u32x4 *out, int64_t n_vec) {
    const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (v >= n_vec) return;
    u32x4 r[N];
#pragma unroll
    for (int m = 0; m < N; m++) r[m] = ld<NTL>(p.in[m] + v);
    u32x4 acc = r[0];
#pragma unroll
    for (int m = 1; m < N; m++) { acc.x += r[m].x; acc.y ^= r[m].y; acc.z += r[m].z; acc.w ^= r[m].w; }
    st<NTS>(out + v, acc);
}

// contiguous chunk per block, UNR vectors in flight per lane (the library's shape)
template <int N, int UNR, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void chunk(Ptrs p, u32x4 *out, int64_t n_vec) {
    const int64_t step = 256 * UNR;
    const int64_t n_it = (n_vec + step - 1) / step;
    const int64_t per = (n_it + gridDim.x - 1) / gridDim.x;
    const int64_t it0 = (int64_t)blockIdx.x * per, it1 = it0 + per < n_it ? it0 + per : n_it;
    for (int64_t it = it0; it < it1; it++) {
        u32x4 r[UNR][N];
#pragma unroll
        for (int u = 0; u < UNR; u++) {
            const int64_t v = it * step + u * 256 + threadIdx.x;
#pragma unroll
            for (int m = 0; m < N; m++) if (v < n_vec) r[u][m] = ld<NTL>(p.in[m] + v);
        }
#pragma unroll
        for (int u = 0; u < UNR; u++) {
            const int64_t v = it * step + u * 256 + threadIdx.x;
            if (v >= n_vec) continue;
            u32x4 acc = r[u][0];
#pragma unroll
            for (int m = 1; m < N; m++) { acc.x += r[u][m].x; acc.y ^= r[u][m].y; acc.z += r[u][m].z; acc.w ^= r[u][m].w; }
            st<NTS>(out + v, acc);
        }
    }
}

static hipEvent_t e0, e1;
template <typename F> float timeit(F &&launch, int iters = 5) {
    launch();
    CHECK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; i++) launch();
    CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms / iters;
}

template <int N>
void report(const char *what, int64_t bytes_per_stream, int64_t skew, float ms) {
    printf("N=%d %-24s footprint/stream %7.0f MiB skew %8lld  %8.3f ms  %6.0f GB/s\n", N, what, bytes_per_stream / 1048576.0, (long long)skew, ms,
           (double)bytes_per_stream * (N + 1) / ms / 1e6);
    fflush(stdout);
}

template <int N>
void sweep(char *pool, int64_t stride, int64_t bytes, int64_t skew) {
    Ptrs p{};
    for (int m = 0; m < N; m++) p.in[m] = reinterpret_cast<const u32x4 *>(pool + m * stride + m * skew);
    u32x4 *out = reinterpret_cast<u32x4 *>(pool + N * stride + N * skew);
    const int64_t n_vec = bytes / 16;
    const unsigned flat_grid = (unsigned)((n_vec + 255) / 256);
    report<N>("flat nt/nt", bytes, skew, timeit([&] { hipLaunchKernelGGL((flat<N, true, true>), dim3(flat_grid), dim3(256), 0, 0, p, out, n_vec); }));
    report<N>("flat plain/plain", bytes, skew, timeit([&] { hipLaunchKernelGGL((flat<N, false, false>), dim3(flat_grid), dim3(256), 0, 0, p, out, n_vec); }));
    report<N>("flat nt/plain", bytes, skew, timeit([&] { hipLaunchKernelGGL((flat<N, true, false>), dim3(flat_grid), dim3(256), 0, 0, p, out, n_vec); }));
    report<N>("chunk u2 nt/nt g2048", bytes, skew, timeit([&] { hipLaunchKernelGGL((chunk<N, 2, true, true>), dim3(2048), dim3(256), 0, 0, p, out, n_vec); }));
    report<N>("chunk u4 nt/nt g2048", bytes, skew, timeit([&] { hipLaunchKernelGGL((chunk<N, 4, true, true>), dim3(2048), dim3(256), 0, 0, p, out, n_vec); }));
    report<N>("chunk u2 nt/nt g8192", bytes, skew, timeit([&] { hipLaunchKernelGGL((chunk<N, 2, true, true>), dim3(8192), dim3(256), 0, 0, p, out, n_vec); }));
    if (N == 1)
        report<N>("hipMemcpyAsync D2D", bytes, skew, timeit([&] { CHECK(hipMemcpyAsync(out, p.in[0], bytes, hipMemcpyDeviceToDevice, 0)); }));
}

int main(int argc, char **argv) {
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int64_t big = (int64_t)6937699840;                 // one SDXL arena of bf16, rounded to 256 B
    const int64_t stride = ((big + (64 << 20)) / (2 << 20) + 1) * (2 << 20);   // 2 MiB aligned slots with room for the skews
    char *pool;
    CHECK(hipMalloc((void **)&pool, (size_t)stride * 5 + (256 << 20)));
    CHECK(hipMemset(pool, 0x3c, (size_t)stride * 5 + (256 << 20)));
    const int64_t sizes[] = {(int64_t)256 << 20, (int64_t)1 << 30, big};
    const int64_t skews[] = {0, 256, 4096, 4096 + 256, 65536 + 4096 + 256, (1 << 20) + 65536 + 4096 + 256};
    for (int64_t bytes : sizes) {
        printf("-- copy (1 in, 1 out), %lld bytes per stream\n", (long long)bytes);
        for (int64_t skew : skews) sweep<1>(pool, stride, bytes, skew);
    }
    printf("-- 4 in, 1 out (n_average / TIES merge shape), SDXL footprint\n");
    for (int64_t skew : skews) sweep<4>(pool, stride, big, skew);
    return 0;
}
