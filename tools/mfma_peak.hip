// What the f32 matrix pipes give on this box when nothing else is in the way: waves that do nothing but
// v_mfma_f32_32x32x2_f32 from registers (4 independent accumulators per wave, as rotate.hip's GEMM), 1 to 4 waves per SIMD.
// The rotate GEMM's roofline fraction is quoted against 157.3 TFLOP/s (256 CUs x 4 SIMDs x 64 flop / cycle x 2.4 GHz); this
// probe shows how much of that a pure MFMA stream reaches (clocks under matrix load included).
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak [launch length multiplier]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void spin(float *out, int iters, float a0, float b0) {
    f32x16 acc[4];
    for (int j = 0; j < 4; j++) for (int i = 0; i < 16; i++) acc[j][i] = 0.0f;
    float a = a0 + threadIdx.x * 1e-9f, b = b0;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++)
#pragma unroll
            for (int j = 0; j < 4; j++) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
    }
    float s = 0.0f;
    for (int j = 0; j < 4; j++) for (int i = 0; i < 16; i++) s += acc[j][i];
    if (s == 12345.678f) out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main(int argc, char **argv) {
    const int mult = argc > 1 ? atoi(argv[1]) : 1;      // longer launches: what the pipes sustain once the power limit acts
    float *out; hipMalloc(&out, 1 << 24);
    int cus = 256; hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int blocks_per_cu = 1; blocks_per_cu <= 4; blocks_per_cu++) {
        const int grid = cus * blocks_per_cu, iters = 4096 * mult;
        spin<<<grid, 256>>>(out, 64, 1.0f, 1.0f);
        hipDeviceSynchronize();
        float best = 1e30f;
        for (int rep = 0; rep < 5; rep++) {
            hipEventRecord(e0); spin<<<grid, 256>>>(out, iters, 1.0f, 1.0f); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        const double flop = (double)grid * 4 /*waves*/ * iters * 32.0 * 4096.0;
        printf("%d wave(s) per SIMD: %.3f ms, %.1f TFLOP/s (%.3f of 157.3)\n", blocks_per_cu, best, flop / best / 1e9, flop / best / 1e9 / 157.3);
    }
    return 0;
}
