// mall_probe.hip -- does the 256 MiB Infinity Cache serve the second sweep of a two-pass job (TIES: count, then merge)?
// Four 6.9 GB input streams and one output stream (the SDXL x4 arenas).  The job is cut into windows of W bytes of input
// (W/4 per stream); per window, pass A reads the four streams (the count pass: read-only), then pass B reads them again and
// writes the output (the merge pass).  Compared: B alone over all windows, A alone, A then B per window, for W from 32 MiB to
// 1 GiB and for the whole arena at once.  If the cache served B's reads, "A then B" would take less than A + B.
// Each (A, B) pair is one launch each, flat grids (one 16-byte vector per lane), like the library's kernels.
// Not part of the library.  Build: hipcc -O3 --offload-arch=gfx950 -o /tmp/mall_probe tools/mall_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

struct Ptrs { const u32x4 *in[4]; };
template <bool NT> __device__ __forceinline__ u32x4 ld(const u32x4 *p) { if constexpr (NT) return __builtin_nontemporal_load(p); else return *p; }

// pass A: read four streams, keep a checksum alive (one rare atomic so that the loads are not dead code)
template <bool NT>
__global__ __launch_bounds__(256) void pass_a(Ptrs p, int64_t v0, int64_t n_vec, uint32_t *sink) {
    const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (v >= n_vec) return;
    u32x4 r[4];
#pragma unroll
    for (int m = 0; m < 4; m++) r[m] = ld<NT>(p.in[m] + v0 + v);
    uint32_t acc = 0;
#pragma unroll
    for (int m = 0; m < 4; m++) acc += r[m].x ^ r[m].y ^ r[m].z ^ r[m].w;
    if (acc == 0x12345677u) atomicAdd(sink, 1u);
}
// pass B: read four streams, write one
template <bool NT>
__global__ __launch_bounds__(256) void pass_b(Ptrs p, u32x4 *out, int64_t v0, int64_t n_vec) {
    const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (v >= n_vec) return;
    u32x4 r[4];
#pragma unroll
    for (int m = 0; m < 4; m++) r[m] = ld<NT>(p.in[m] + v0 + v);
    u32x4 acc = r[0];
#pragma unroll
    for (int m = 1; m < 4; m++) { acc.x += r[m].x; acc.y ^= r[m].y; acc.z += r[m].z; acc.w ^= r[m].w; }
    __builtin_nontemporal_store(acc, out + v0 + v);
}

static hipEvent_t e0, e1;
template <typename F> float timeit(F &&run, int iters = 3) {
    run();
    CHECK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; i++) run();
    CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms / iters;
}

int main() {
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int64_t big = (int64_t)6937699840;                 // one SDXL arena of bf16
    const int64_t stride = (big / (2 << 20) + 2) * (2 << 20) + 4096 + 256;     // skewed slots (tools/stream_bench3.hip)
    char *pool; uint32_t *sink;
    CHECK(hipMalloc((void **)&pool, (size_t)stride * 5));
    CHECK(hipMemset(pool, 0x3c, (size_t)stride * 5));
    CHECK(hipMalloc((void **)&sink, 256)); CHECK(hipMemset(sink, 0, 256));
    Ptrs p{};
    for (int m = 0; m < 4; m++) p.in[m] = reinterpret_cast<const u32x4 *>(pool + m * stride);
    u32x4 *out = reinterpret_cast<u32x4 *>(pool + 4 * stride);
    const int64_t n_vec = big / 16;
    const int64_t windows_mib[] = {0, 1024, 512, 256, 192, 160, 128, 96, 64, 32};
    printf("window = bytes of INPUT per window (4 streams); times per whole job (27.75 GB in, 6.94 GB out)\n");
    const int policies[3][2] = {{1, 1}, {0, 0}, {0, 1}};                 // {pass A non-temporal, pass B non-temporal}
    for (const auto &pol : policies) {
        const int nta = pol[0], ntb = pol[1];
        for (int64_t w : windows_mib) {
            const int64_t per = (w == 0) ? n_vec : (w << 20) / 4 / 16;       // vectors per stream per window
            auto run = [&](bool a, bool b) {
                for (int64_t v0 = 0; v0 < n_vec; v0 += per) {
                    const int64_t n = (n_vec - v0 < per) ? n_vec - v0 : per;
                    const unsigned grid = (unsigned)((n + 255) / 256);
                    if (a) { if (nta) hipLaunchKernelGGL(pass_a<true>, dim3(grid), dim3(256), 0, 0, p, v0, n, sink);
                             else hipLaunchKernelGGL(pass_a<false>, dim3(grid), dim3(256), 0, 0, p, v0, n, sink); }
                    if (b) { if (ntb) hipLaunchKernelGGL(pass_b<true>, dim3(grid), dim3(256), 0, 0, p, out, v0, n);
                             else hipLaunchKernelGGL(pass_b<false>, dim3(grid), dim3(256), 0, 0, p, out, v0, n); }
                }
            };
            const float ta = timeit([&] { run(true, false); }), tb = timeit([&] { run(false, true); }), tab = timeit([&] { run(true, true); });
            printf("loads A %-7s B %-7s window %5lld MiB (%4lld windows)  A alone %7.3f ms (%5.0f GB/s)  B alone %7.3f ms (%5.0f GB/s)  A then B %7.3f ms  vs A+B %7.3f  saved %5.1f %%\n",
                   nta ? "nt" : "default", ntb ? "nt" : "default", (long long)w, (long long)((n_vec + per - 1) / per), ta, big * 4.0 / ta / 1e6, tb, big * 5.0 / tb / 1e6, tab, ta + tb,
                   100.0 * (ta + tb - tab) / (ta + tb));
            fflush(stdout);
        }
    }
    return 0;
}
