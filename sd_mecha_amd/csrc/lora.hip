// lora.hip -- LoRA / LyCORIS up.down reconstruction on MFMA, fused with add_difference into the base key.
//
// Reference: extensions/builtin/lycoris.py:226-263 (LoraAlgorithm.compose / compose_alpha):
//   delta_i = (up_i.view(out, -1) @ down_i.view(r_i, -1)) * (alpha_i / r_i  or 1), reshaped to the base shape,
// then add_difference(base, delta_i, w_i) = fma(delta_i, w_i, base) per network (linear.py:76).
// One pass over the base key applies EVERY network: out = base (+)_{i} fma(delta_i, w_i, .) in network order, so
// the base tile is read once and written once whatever the number of networks (SURVEY.md §7 "LoRA is memory-bound
// unless fused across networks").  With rank-32 factors the product has 16 flop per byte of base + output: the kernel is
// HBM-bound, the MFMA only keeps the arithmetic out of the way.
//
// The product is computed transposed, D^T = down^T . up^T, with v_mfma_f32_32x32x16_{f16,bf16}:
//   * B operand = up^T: lane (row r, half h) takes the 8 contiguous k of up[row][.] straight from global memory (16 B);
//   * A operand = down^T: down is [r, cols], so a lane's 8 k-values of one column are a strided gather.  The block stages
//     the (32 k) x (64 columns) slab of down through LDS instead: coalesced 16-byte row loads, transposed 2-byte LDS
//     writes, one 16-byte LDS read per lane and k-step.  The slab of the next k-chunk / network is loaded while the
//     current one is multiplied (double-buffered LDS, one barrier per chunk);
//   * each lane ends up with 4 CONTIGUOUS output columns per accumulator group: 8-byte (16-byte for fp32) base loads and
//     output stores.
// fp16/bf16 products are exact in fp32; accumulation is fp32 (the reference's sgemm order is BLAS-defined, parity is
// the <= 1 ulp of the output dtype bar).  Scale is applied AFTER the matmul, in fp32, like the reference.
//
// Block = 4 waves, block tile = 64 output rows x 64 output columns, wave tile 32 x 32.  The batched entry point runs
// every key of a wave in ONE launch: tiles of all keys are numbered consecutively, a block finds its key by binary search
// in the tile prefix (keys are independent, a few thousand per checkpoint; a launch per key costs more than its tiles).
#include <type_traits>
#include <vector>
#include "common.h"

namespace {

constexpr int LORA_MAX_NETS = MECHA_MAX_MODELS;
constexpr int KC = 32;                 // k-chunk staged per LDS buffer
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint16_t u16x8 __attribute__((ext_vector_type(8)));
typedef uint16_t u16x4 __attribute__((ext_vector_type(4)));

struct LoraParams {
    const void *base;                  // may be nullptr: delta only
    void *out;
    int base_dt, out_dt;
    uint32_t *flags;                   // per key (batched) or one word
    // batched: device tables; per key: the by-value tables below
    const mecha_lora_key *keys;
    const mecha_lora_net *nets;
    int n_keys;
    mecha_lora_key key0;
    mecha_lora_net net0[LORA_MAX_NETS];
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

// 4 contiguous elements of the base / output key
__device__ __forceinline__ void load4(int dt, bool vec, const void *p, int64_t i, int valid, float (&v)[4]) {
    if (vec) {
        if (dt == MECHA_F32) {
            const f32x4 t = *reinterpret_cast<const f32x4 *>(reinterpret_cast<const float *>(p) + i);
            v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
        } else {
            const u16x4 t = *reinterpret_cast<const u16x4 *>(reinterpret_cast<const uint16_t *>(p) + i);
#pragma unroll
            for (int q = 0; q < 4; q++) v[q] = (dt == MECHA_BF16) ? bf16_bits_to_f32(t[q]) : f16_bits_to_f32(t[q]);
        }
    } else {
#pragma unroll
        for (int q = 0; q < 4; q++)
            v[q] = (q < valid) ? ((dt == MECHA_BF16) ? load1<MECHA_BF16>(p, i + q) : (dt == MECHA_F16) ? load1<MECHA_F16>(p, i + q) : load1<MECHA_F32>(p, i + q)) : 0.0f;
    }
}
__device__ __forceinline__ void store4(int dt, bool vec, void *p, int64_t i, int valid, const float (&v)[4]) {
    if (vec) {
        if (dt == MECHA_F32) {
            *reinterpret_cast<f32x4 *>(reinterpret_cast<float *>(p) + i) = f32x4{v[0], v[1], v[2], v[3]};
        } else {
            u16x4 t;
#pragma unroll
            for (int q = 0; q < 4; q++) t[q] = (uint16_t)((dt == MECHA_BF16) ? f32_to_bf16_bits(v[q]) : f32_to_f16_bits(v[q]));
            *reinterpret_cast<u16x4 *>(reinterpret_cast<uint16_t *>(p) + i) = t;
        }
    } else {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            if (q < valid) {
                if (dt == MECHA_BF16) store1<MECHA_BF16>(p, i + q, v[q]);
                else if (dt == MECHA_F16) store1<MECHA_F16>(p, i + q, v[q]);
                else store1<MECHA_F32>(p, i + q, v[q]);
            }
        }
    }
}

// one (k-chunk) slab of down for this block: thread t holds k = t / 8, columns 8 * (t % 8) .. + 7
template <int LDT>
__device__ __forceinline__ u16x8 load_down(const uint16_t *down, int rank, int64_t cols, int64_t col0, int k0, bool vec_ok, bool check, bool &bad) {
    const int k = k0 + (threadIdx.x >> 3);
    const int64_t c = col0 + 8 * (threadIdx.x & 7);
    u16x8 v;
    if (k < rank && vec_ok && c + 8 <= cols) {
        v = *reinterpret_cast<const u16x8 *>(down + (int64_t)k * cols + c);
    } else {
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = (k < rank && c + j < cols) ? down[(int64_t)k * cols + c + j] : (uint16_t)0;
    }
    if (check) {
#pragma unroll
        for (int j = 0; j < 8; j++) bad |= nonfinite16<LDT>(v[j]);
    }
    return v;
}
// LDS slab layout: [k / 8][column (+1 slot of padding per 8 columns)][k % 8]: a lane's A operand (8 k of one column) is
// one aligned 16-byte read, consecutive columns are consecutive 16-byte slots, and the transposing 2-byte writes of the
// 8 column groups of a wave land in 8 different banks (pitch 72 slots = 36 dwords between groups of 8 columns).
constexpr int SLAB_HALFS = (KC / 8) * 72 * 8;
__device__ __forceinline__ int slab_index(int k, int col) { return (((k >> 3) * 72 + col + (col >> 3)) << 3) + (k & 7); }
__device__ __forceinline__ void stage_down(uint16_t *slab, u16x8 v) {
    const int k = threadIdx.x >> 3, c = 8 * (threadIdx.x & 7);
#pragma unroll
    for (int j = 0; j < 8; j++) slab[slab_index(k, c + j)] = v[j];
}
// B operand: 8 contiguous k of one row of up for each of the chunk's two k-steps
template <int LDT>
__device__ __forceinline__ void load_up(const uint16_t *up, int rank, int64_t row, bool row_ok, int k0, int h, bool vec_ok, bool check, u16x8 (&b)[2], bool &bad) {
#pragma unroll
    for (int s = 0; s < 2; s++) {
        const int kb = k0 + 16 * s + 8 * h;
        if (vec_ok && row_ok && kb + 8 <= rank) {
            b[s] = *reinterpret_cast<const u16x8 *>(up + row * rank + kb);
        } else {
#pragma unroll
            for (int j = 0; j < 8; j++) b[s][j] = (row_ok && kb + j < rank) ? up[row * rank + kb + j] : (uint16_t)0;
        }
        if (check) {
#pragma unroll
            for (int j = 0; j < 8; j++) bad |= nonfinite16<LDT>(b[s][j]);
        }
    }
}

template <int LDT, bool BATCHED>
__global__ __launch_bounds__(MECHA_BLOCK) void lora_kernel(LoraParams p) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[2][2][SLAB_HALFS];   // [buffer][slot of the pair]
    // ---- which key, which tile
    mecha_lora_key key;
    const mecha_lora_net *nets;
    int key_index = 0;
    int64_t tile = blockIdx.x;
    if constexpr (BATCHED) {
        int lo = 0, hi = p.n_keys - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (p.keys[mid].tile_start <= tile) lo = mid; else hi = mid - 1;
        }
        key_index = lo;
        key = p.keys[lo];
        nets = p.nets + key.net_begin;
        tile -= key.tile_start;
    } else {
        key = p.key0;
        nets = p.net0;
    }
    const int64_t tiles_x = (key.cols + 63) / 64;
    const int64_t tile_y = tile / tiles_x, tile_x = tile % tiles_x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    // transposed product: MFMA rows <-> output columns, MFMA columns <-> output rows
    const int64_t bcol0 = tile_x * 64;                                    // block's first output column
    const int wcol = (wave & 1) * 32;                                     // this wave's columns inside the block tile
    const int64_t out_row = tile_y * 64 + (wave >> 1) * 32 + r;           // D^T column index = lane & 31
    const bool row_ok = out_row < key.rows;
    const char *base = static_cast<const char *>(p.base);
    const int64_t elem0 = key.offset + out_row * key.cols;                // first element of this lane's row in the arenas
    const bool vec4 = ((key.cols & 3) == 0) && ((key.offset & 3) == 0);   // rows start 4-element aligned (arenas are 16-B aligned)
    // every factor element is read by a whole row / column of tiles: the tiles on the key's first column check `up`,
    // those on its first row check `down`, once each
    const bool check_up = tile_x == 0, check_down = tile_y == 0;

    // total starts at the base value (fp32).  D^T layout: reg g*4+q holds output column bcol0 + wcol + 8g + 4h + q of out_row.
    f32x16 total;
    bool bad_in = false;
#pragma unroll
    for (int g = 0; g < 4; g++) {
        const int64_t c = bcol0 + wcol + 8 * g + 4 * h;
        float v[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        if (base != nullptr && row_ok && c < key.cols) {
            const int valid = key.cols - c >= 4 ? 4 : (int)(key.cols - c);
            load4(p.base_dt, vec4 && valid == 4, base, elem0 + c, valid, v);
        }
#pragma unroll
        for (int q = 0; q < 4; q++) { bad_in |= nonfinite_f32(v[q]); total[g * 4 + q] = v[q]; }
    }

    // ---- the chunk sequence over (network, k0), two chunks per round: the next pair's global loads are in flight while
    // the current pair is multiplied; one barrier per pair (double-buffered LDS)
    struct Chunk { int n, k0; bool ok; };
    int n = 0, k0 = 0;
    const int n_nets = key.net_count;
    auto take = [&]() {
        Chunk c{n, k0, n < n_nets};
        if (c.ok) { k0 += KC; if (k0 >= nets[n].rank) { k0 = 0; n++; } }
        return c;
    };
    u16x8 dv[2], bv[2][2];
    auto fetch = [&](const Chunk &c, int slot) {
        if (!c.ok) return;
        const mecha_lora_net net = nets[c.n];
        const uint16_t *down = static_cast<const uint16_t *>(net.down), *up = static_cast<const uint16_t *>(net.up);
        dv[slot] = load_down<LDT>(down, net.rank, key.cols, bcol0, c.k0, ((key.cols & 7) == 0) && ((reinterpret_cast<uintptr_t>(down) & 15u) == 0), check_down, bad_in);
        load_up<LDT>(up, net.rank, out_row, row_ok, c.k0, h, ((net.rank & 7) == 0) && ((reinterpret_cast<uintptr_t>(up) & 15u) == 0), check_up, bv[slot], bad_in);
    };
    Chunk next[2] = {take(), take()};
    fetch(next[0], 0);
    fetch(next[1], 1);
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; i++) acc[i] = 0.0f;
    int buf = 0;
    while (next[0].ok) {
        const Chunk now[2] = {next[0], next[1]};
        u16x8 b_now[2][2];
#pragma unroll
        for (int slot = 0; slot < 2; slot++) {
            if (now[slot].ok) stage_down(lds[buf][slot], dv[slot]);
            b_now[slot][0] = bv[slot][0]; b_now[slot][1] = bv[slot][1];
        }
        next[0] = take(); next[1] = take();
        fetch(next[0], 0);
        fetch(next[1], 1);
        __syncthreads();                                                  // pair `buf` is complete; the other buffer's readers are done (barrier of the previous round)
#pragma unroll
        for (int slot = 0; slot < 2; slot++) {
            if (!now[slot].ok) continue;
            const int rank = nets[now[slot].n].rank;
#pragma unroll
            for (int s = 0; s < 2; s++) {
                if (now[slot].k0 + 16 * s < rank) {
                    const u16x8 a = *reinterpret_cast<const u16x8 *>(&lds[buf][slot][slab_index(16 * s + 8 * h, wcol + r)]);
                    acc = mfma16<LDT>(a, b_now[slot][s], acc);
                }
            }
            if (now[slot].k0 + KC >= rank) {                              // the network is complete: delta * scale, then addcmul
                const float scale = nets[now[slot].n].scale, weight = nets[now[slot].n].weight;
#pragma unroll
                for (int i = 0; i < 16; i++) { total[i] = __builtin_fmaf(acc[i] * scale, weight, total[i]); acc[i] = 0.0f; }
            }
        }
        buf ^= 1;
    }

    bool bad_out = false;
#pragma unroll
    for (int g = 0; g < 4; g++) {
        const int64_t c = bcol0 + wcol + 8 * g + 4 * h;
        if (row_ok && c < key.cols) {
            float v[4];
#pragma unroll
            for (int q = 0; q < 4; q++) { v[q] = total[g * 4 + q]; }
            const int valid = key.cols - c >= 4 ? 4 : (int)(key.cols - c);
#pragma unroll
            for (int q = 0; q < 4; q++) bad_out |= (q < valid) && nonfinite_f32(v[q]);
            store4(p.out_dt, vec4 && valid == 4, p.out, elem0 + c, valid, v);
        }
    }
    if ((bad_in | bad_out) && p.flags != nullptr)
        atomicOr(p.flags + key_index, (bad_in ? MECHA_FLAG_INPUT_NONFINITE : 0u) | (bad_out ? MECHA_FLAG_OUTPUT_NONFINITE : 0u));
}

int check_common(const void *out, int base_dtype, int out_dtype, int lora_dtype, const void *base) {
    MECHA_REQUIRE(out != nullptr, "out is NULL");
    MECHA_REQUIRE(mecha_valid_dtype(out_dtype) && (base == nullptr || mecha_valid_dtype(base_dtype)), "unknown dtype code");
    MECHA_REQUIRE(lora_dtype == MECHA_F16 || lora_dtype == MECHA_BF16,
                  "LoRA factors must be float16 or bfloat16 (got dtype code %d): the MFMA path multiplies 16-bit operands exactly", lora_dtype);
    MECHA_REQUIRE((base == nullptr || mecha_aligned16(base)) && mecha_aligned16(out), "base / out must be 16-byte aligned");
    return MECHA_OK;
}

}  // namespace

extern "C" {

int mecha_lora_accumulate(const void *base, void *out, int64_t rows, int64_t cols, int base_dtype, int out_dtype,
                          const void *const *ups, const void *const *downs, const int32_t *ranks, const float *scales,
                          const float *weights, int n_networks, int lora_dtype, uint32_t *flags, void *stream) {
    MECHA_REQUIRE(rows >= 0 && cols >= 0, "rows/cols must be >= 0");
    if (rows == 0 || cols == 0) return MECHA_OK;
    if (int e = check_common(out, base_dtype, out_dtype, lora_dtype, base)) return e;
    MECHA_REQUIRE(n_networks >= 0 && n_networks <= LORA_MAX_NETS, "at most %d networks per call (got %d)", LORA_MAX_NETS, n_networks);
    MECHA_REQUIRE(n_networks == 0 || (ups && downs && ranks && scales && weights), "NULL network table");
    MECHA_REQUIRE(rows * cols < (int64_t)1 << 40, "key too large");
    LoraParams p{};
    p.base = base; p.out = out; p.base_dt = base_dtype; p.out_dt = out_dtype; p.flags = flags;
    int n = 0;
    for (int i = 0; i < n_networks; i++) {
        MECHA_REQUIRE(ups[i] != nullptr && downs[i] != nullptr && ranks[i] >= 1, "network %d: NULL factor or rank < 1", i);
        if (mecha_isclose(weights[i], 0.0)) continue;            // add_difference with alpha == 0 returns `a` (linear.py:72-73)
        p.net0[n] = mecha_lora_net{ups[i], downs[i], ranks[i], scales[i], weights[i], 0};
        n++;
    }
    p.key0 = mecha_lora_key{0, rows, cols, 0, 0, n};
    p.n_keys = 1;
    const int64_t tiles = ((cols + 63) / 64) * ((rows + 63) / 64);
    MECHA_REQUIRE(tiles <= 0x7fffffff, "grid too large");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (lora_dtype == MECHA_F16) MECHA_LAUNCH(MK_LORA, (lora_kernel<MECHA_F16, false>), dim3((unsigned)tiles), dim3(MECHA_BLOCK), 0, s, p);
    else MECHA_LAUNCH(MK_LORA, (lora_kernel<MECHA_BF16, false>), dim3((unsigned)tiles), dim3(MECHA_BLOCK), 0, s, p);
    MECHA_CHECK_HIP(hipGetLastError());
    return MECHA_OK;
}

int mecha_batched_lora_accumulate(const void *base_arena, void *out_arena, const mecha_lora_key *keys, const mecha_lora_key *keys_host,
                                  int n_keys, const mecha_lora_net *nets, int base_dtype, int out_dtype, int lora_dtype,
                                  uint32_t *key_flags, void *stream) {
    MECHA_REQUIRE(n_keys >= 0, "n_keys must be >= 0");
    if (n_keys == 0) return MECHA_OK;
    if (int e = check_common(out_arena, base_dtype, out_dtype, lora_dtype, base_arena)) return e;
    MECHA_REQUIRE(keys != nullptr && keys_host != nullptr && nets != nullptr, "NULL key / network table");
    int64_t tiles = 0;
    for (int i = 0; i < n_keys; i++) {
        const mecha_lora_key &k = keys_host[i];
        MECHA_REQUIRE(k.rows >= 1 && k.cols >= 1 && k.offset >= 0 && k.net_begin >= 0 && k.net_count >= 0 && k.net_count <= LORA_MAX_NETS,
                      "key %d: bad shape, offset or network range", i);
        MECHA_REQUIRE(k.tile_start == tiles, "key %d: tile_start must be the running sum of ceil(rows/64) * ceil(cols/64) (expected %lld)", i, (long long)tiles);
        tiles += ((k.cols + 63) / 64) * ((k.rows + 63) / 64);
    }
    MECHA_REQUIRE(tiles <= 0x7fffffff, "grid too large");
    LoraParams p{};
    p.base = base_arena; p.out = out_arena; p.base_dt = base_dtype; p.out_dt = out_dtype; p.flags = key_flags;
    p.keys = keys; p.nets = nets; p.n_keys = n_keys;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (lora_dtype == MECHA_F16) MECHA_LAUNCH(MK_LORA, (lora_kernel<MECHA_F16, true>), dim3((unsigned)tiles), dim3(MECHA_BLOCK), 0, s, p);
    else MECHA_LAUNCH(MK_LORA, (lora_kernel<MECHA_BF16, true>), dim3((unsigned)tiles), dim3(MECHA_BLOCK), 0, s, p);
    MECHA_CHECK_HIP(hipGetLastError());
    return MECHA_OK;
}

}  // extern "C"
