// lora.hip -- LoRA / LyCORIS up.down reconstruction on MFMA, fused with add_difference into the base key.
//
// Reference: extensions/builtin/lycoris.py:226-263 (LoraAlgorithm.compose / compose_alpha):
//   delta_i = (up_i.view(out, -1) @ down_i.view(r_i, -1)) * (alpha_i / r_i  or 1), reshaped to the base shape,
// then add_difference(base, delta_i, w_i) = fma(delta_i, w_i, base) per network (linear.py:76).
// One pass over the base key applies EVERY network: out = base (+)_{i} fma(delta_i, w_i, .) in network order, so
// the base tile is read once and written once whatever the number of networks (SURVEY.md §7 "LoRA is memory-bound
// unless fused across networks").
//
// The product is computed transposed, D^T = down^T . up^T, with v_mfma_f32_32x32x16_{f16,bf16}: the 8-element k-run
// of up (contiguous in memory, up is [out, r]) is the B operand, the k-strided column of down the A operand, and each
// lane ends up holding 4 CONTIGUOUS output elements per accumulator group -> 8-byte base loads and output stores.
// fp16/bf16 products are exact in fp32; accumulation is fp32 (the reference's sgemm order is BLAS-defined, parity is
// the <= 1 ulp of the output dtype bar).  Scale is applied AFTER the matmul, in fp32, like the reference.
//
// Block = 4 waves, block tile = 64 output rows x 64 output columns, wave tile 32 x 32.
// Roofline: MFMA for many networks / large rank (2*rows*cols*sum(r_i) flop), HBM for few.
#include <type_traits>
#include "common.h"

namespace {

constexpr int LORA_MAX_NETS = MECHA_MAX_MODELS;
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint16_t u16x8 __attribute__((ext_vector_type(8)));

struct LoraParams {
    const void *base;                 // may be nullptr: delta only
    void *out;
    int64_t rows, cols;
    int base_dt, out_dt;
    int n_nets;
    const uint16_t *up[LORA_MAX_NETS];
    const uint16_t *down[LORA_MAX_NETS];
    int rank[LORA_MAX_NETS];
    float scale[LORA_MAX_NETS];
    float weight[LORA_MAX_NETS];
    uint32_t *flags;
};

template <int LDT>
__device__ __forceinline__ bool nonfinite16(uint32_t h) {
    if constexpr (LDT == MECHA_F16) return (h & 0x7c00u) == 0x7c00u;
    else return (h & 0x7f80u) == 0x7f80u;
}

template <int LDT>
__device__ __forceinline__ f32x16 mfma16(u16x8 a, u16x8 b, f32x16 c) {
    if constexpr (LDT == MECHA_F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

__device__ __forceinline__ float load_elem(int dt, const void *p, int64_t i) {
    return (dt == MECHA_BF16) ? load1<MECHA_BF16>(p, i) : (dt == MECHA_F16) ? load1<MECHA_F16>(p, i) : load1<MECHA_F32>(p, i);
}
__device__ __forceinline__ void store_elem(int dt, void *p, int64_t i, float v) {
    if (dt == MECHA_BF16) store1<MECHA_BF16>(p, i, v);
    else if (dt == MECHA_F16) store1<MECHA_F16>(p, i, v);
    else store1<MECHA_F32>(p, i, v);
}

template <int LDT>
__global__ __launch_bounds__(MECHA_BLOCK) void lora_kernel(LoraParams p) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    // transposed product: MFMA rows <-> output columns, MFMA columns <-> output rows
    const int64_t col0 = (int64_t)blockIdx.x * 64 + (wave & 1) * 32;      // output columns of this wave's tile
    const int64_t row0 = (int64_t)blockIdx.y * 64 + (wave >> 1) * 32;     // output rows of this wave's tile
    if (col0 >= p.cols || row0 >= p.rows) return;
    const int64_t out_row = row0 + r;                                     // D^T column index = lane & 31
    const bool row_ok = out_row < p.rows;
    const int64_t a_col = col0 + r;                                       // A operand: row of down^T = output column
    const bool a_ok = a_col < p.cols;
    const bool cols_vec4 = (p.cols & 3) == 0;

    // total starts at the base value (fp32).  D^T layout: reg g*4+q holds output column col0 + 8g + 4h + q of out_row.
    f32x16 total;
    bool bad_in = false;
#pragma unroll
    for (int g = 0; g < 4; g++) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int64_t c = col0 + 8 * g + 4 * h + q;
            float v = 0.0f;
            if (p.base != nullptr && row_ok && c < p.cols) v = load_elem(p.base_dt, p.base, out_row * p.cols + c);
            bad_in |= nonfinite_f32(v);
            total[g * 4 + q] = v;
        }
    }
    (void)cols_vec4;

    for (int n = 0; n < p.n_nets; n++) {
        const int rank = p.rank[n];
        const uint16_t *up = p.up[n], *down = p.down[n];
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; i++) acc[i] = 0.0f;
        const bool up_vec = ((rank & 7) == 0) && ((reinterpret_cast<uintptr_t>(up) & 15u) == 0);
        for (int k0 = 0; k0 < rank; k0 += 16) {
            const int kb = k0 + 8 * h;
            u16x8 a, b;
            // A = down^T: A[m = output column][k] = down[k][column]  (k-strided gathers, coalesced across lanes)
#pragma unroll
            for (int j = 0; j < 8; j++) {
                uint16_t v = 0;
                if (a_ok && kb + j < rank) v = down[(int64_t)(kb + j) * p.cols + a_col];
                bad_in |= nonfinite16<LDT>(v);
                a[j] = v;
            }
            // B = up^T: B[k][n = output row] = up[row][k]  (8 contiguous k per lane)
            if (up_vec && row_ok && kb + 8 <= rank) {
                b = *reinterpret_cast<const u16x8 *>(up + out_row * rank + kb);
            } else {
#pragma unroll
                for (int j = 0; j < 8; j++) b[j] = (row_ok && kb + j < rank) ? up[out_row * rank + kb + j] : (uint16_t)0;
            }
#pragma unroll
            for (int j = 0; j < 8; j++) bad_in |= nonfinite16<LDT>(b[j]);
            acc = mfma16<LDT>(a, b, acc);
        }
        const float scale = p.scale[n], weight = p.weight[n];
#pragma unroll
        for (int i = 0; i < 16; i++) total[i] = __builtin_fmaf(acc[i] * scale, weight, total[i]);   // delta * scale, then addcmul
    }

    bool bad_out = false;
#pragma unroll
    for (int g = 0; g < 4; g++) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int64_t c = col0 + 8 * g + 4 * h + q;
            if (row_ok && c < p.cols) {
                const float v = total[g * 4 + q];
                bad_out |= nonfinite_f32(v);
                store_elem(p.out_dt, p.out, out_row * p.cols + c, v);
            }
        }
    }
    if ((bad_in | bad_out) && p.flags != nullptr)
        atomicOr(p.flags, (bad_in ? MECHA_FLAG_INPUT_NONFINITE : 0u) | (bad_out ? MECHA_FLAG_OUTPUT_NONFINITE : 0u));
}

}  // namespace

extern "C" {

int mecha_lora_accumulate(const void *base, void *out, int64_t rows, int64_t cols, int base_dtype, int out_dtype,
                          const void *const *ups, const void *const *downs, const int32_t *ranks, const float *scales,
                          const float *weights, int n_networks, int lora_dtype, uint32_t *flags, void *stream) {
    MECHA_REQUIRE(rows >= 0 && cols >= 0, "rows/cols must be >= 0");
    if (rows == 0 || cols == 0) return MECHA_OK;
    MECHA_REQUIRE(out != nullptr, "out is NULL");
    MECHA_REQUIRE(mecha_valid_dtype(out_dtype) && (base == nullptr || mecha_valid_dtype(base_dtype)), "unknown dtype code");
    MECHA_REQUIRE(lora_dtype == MECHA_F16 || lora_dtype == MECHA_BF16,
                  "LoRA factors must be float16 or bfloat16 (got dtype code %d): the MFMA path multiplies 16-bit operands exactly", lora_dtype);
    MECHA_REQUIRE(n_networks >= 0 && n_networks <= LORA_MAX_NETS, "at most %d networks per call (got %d)", LORA_MAX_NETS, n_networks);
    MECHA_REQUIRE(n_networks == 0 || (ups && downs && ranks && scales && weights), "NULL network table");
    MECHA_REQUIRE(rows * cols < (int64_t)1 << 40, "key too large");
    LoraParams p{};
    p.base = base; p.out = out; p.rows = rows; p.cols = cols; p.base_dt = base_dtype; p.out_dt = out_dtype; p.flags = flags;
    int n = 0;
    for (int i = 0; i < n_networks; i++) {
        MECHA_REQUIRE(ups[i] != nullptr && downs[i] != nullptr && ranks[i] >= 1, "network %d: NULL factor or rank < 1", i);
        if (mecha_isclose(weights[i], 0.0)) continue;            // add_difference with alpha == 0 returns `a` (linear.py:72-73)
        p.up[n] = static_cast<const uint16_t *>(ups[i]);
        p.down[n] = static_cast<const uint16_t *>(downs[i]);
        p.rank[n] = ranks[i]; p.scale[n] = scales[i]; p.weight[n] = weights[i];
        n++;
    }
    p.n_nets = n;
    const int64_t gx = (cols + 63) / 64, gy = (rows + 63) / 64;
    MECHA_REQUIRE(gy <= 65535 && gx <= 0x7fffffff, "grid too large");
    dim3 grid((unsigned)gx, (unsigned)gy), block(MECHA_BLOCK);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (lora_dtype == MECHA_F16) MECHA_LAUNCH(MK_LORA, (lora_kernel<MECHA_F16>), grid, block, 0, s, p);
    else MECHA_LAUNCH(MK_LORA, (lora_kernel<MECHA_BF16>), grid, block, 0, s, p);
    MECHA_CHECK_HIP(hipGetLastError());
    return MECHA_OK;
}

}  // extern "C"
