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
//     the (32 k) x (128 columns) slab of down through LDS as it lies in memory (coalesced 16-byte row loads, 16-byte LDS
//     writes) and the MFMA lanes take their fragments with ds_read_b64_tr_b16, gfx950's transposing LDS read: two 8-byte
//     reads per k-step and column group.  Rows are 320 B apart so that the 4 k-rows a 16-lane group gathers start 64 B
//     apart in the 256-B bank row: conflict-free.  The slab of the next k-chunk / network is loaded while the current one
//     is multiplied (double-buffered LDS, one barrier per chunk);
//   * each lane ends up with 4 CONTIGUOUS output columns per accumulator group: 8-byte (16-byte for fp32) base loads and
//     output stores.
// fp16/bf16 products are exact in fp32; accumulation is fp32 (the reference's sgemm order is BLAS-defined, parity is
// the <= 1 ulp of the output dtype bar).  Scale is applied AFTER the matmul, in fp32, like the reference.
//
// Block = 4 waves, block tile = 128 output rows x 128 output columns, wave tile 64 x 64 (2 x 2 MFMA tiles: 8 MFMAs per
// 4 operand loads from global memory and 8 from LDS per k-chunk; a 128 x 128 tile pulls 1 byte of factors through L2 per
// output element and network).  The batched entry point runs every key of a wave in ONE launch: tiles of all keys are
// numbered consecutively, a block finds its key by binary search in the tile prefix (keys are independent, a few thousand
// per checkpoint; a launch per key costs more than its tiles).  Workgroups go round-robin over the 8 XCDs, each with its
// own L2: block b takes tile (b % 8) * ceil(T / 8) + b / 8, so that one XCD works through a contiguous run of tiles and a
// key's factors are fetched into one L2, not eight.
#include <cstdlib>
#include <type_traits>
#include <vector>
#include "common.h"

namespace {

constexpr int LORA_MAX_NETS = MECHA_MAX_MODELS;
constexpr int LORA_MAX_GENERAL = 256;
constexpr int KC = 32;                 // k-chunk staged per LDS buffer
constexpr int LORA_TILE = MECHA_LORA_TILE;   // block tile, rows and columns
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint16_t u16x8 __attribute__((ext_vector_type(8)));
typedef uint16_t u16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

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
    // batched, general kernel: the keys it has to do, when the host knows them and they fit (else n_general = -1: it walks all tiles)
    int n_general;
    int32_t general_keys[LORA_MAX_GENERAL];
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
            // (non-temporal: the base and output tiles are touched once; the factors, re-read by every tile of a key, keep the L2)
            const f32x4 t = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(reinterpret_cast<const float *>(p) + i));
            v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
        } else {
            const u16x4 t = __builtin_nontemporal_load(reinterpret_cast<const u16x4 *>(reinterpret_cast<const uint16_t *>(p) + i));
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
            __builtin_nontemporal_store(f32x4{v[0], v[1], v[2], v[3]}, reinterpret_cast<f32x4 *>(reinterpret_cast<float *>(p) + i));
        } else {
            u16x4 t;
#pragma unroll
            for (int q = 0; q < 4; q++) t[q] = (uint16_t)((dt == MECHA_BF16) ? f32_to_bf16_bits(v[q]) : f32_to_f16_bits(v[q]));
            __builtin_nontemporal_store(t, reinterpret_cast<u16x4 *>(reinterpret_cast<uint16_t *>(p) + i));
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

// one (k-chunk) slab of down for this block: thread t holds k = t / 16 (+ 16), columns 8 * (t % 16) .. + 7
template <int LDT>
__device__ __forceinline__ void load_down(const uint16_t *down, int rank, int64_t cols, int64_t col0, int k0, bool vec_ok, bool check, u16x8 (&v)[2], bool &bad) {
    const int64_t c = col0 + 8 * (threadIdx.x & 15);
#pragma unroll
    for (int half = 0; half < 2; half++) {
        const int k = k0 + (threadIdx.x >> 4) + 16 * half;
        if (k < rank && vec_ok && c + 8 <= cols) {
            v[half] = *reinterpret_cast<const u16x8 *>(down + (int64_t)k * cols + c);
        } else {
#pragma unroll
            for (int j = 0; j < 8; j++) v[half][j] = (k < rank && c + j < cols) ? down[(int64_t)k * cols + c + j] : (uint16_t)0;
        }
        if (check) {
#pragma unroll
            for (int j = 0; j < 8; j++) bad |= nonfinite16<LDT>(v[half][j]);
        }
    }
}
// LDS slab: [k][PITCH], as in memory.  PITCH = 128 columns + 32 halfs (64 B): see the header.
constexpr int PITCH = 160;
constexpr int SLAB_HALFS = KC * PITCH;
__device__ __forceinline__ void stage_down(uint16_t *slab, const u16x8 (&v)[2]) {
#pragma unroll
    for (int half = 0; half < 2; half++)
        *reinterpret_cast<u16x8 *>(&slab[((threadIdx.x >> 4) + 16 * half) * PITCH + 8 * (threadIdx.x & 15)]) = v[half];
}
// A fragment of k-step s (16 k) for the 32 columns at `col` of the slab: lane (r = lane & 31, h = lane >> 5) gets
// down[16 s + 8 h + j][col + r], j = 0..7.  ds_read_b64_tr_b16 works per group of 16 lanes: lane 4q + p of a group gives the
// address of row q, columns 4p..4p+3 of a 4 x 16 block; lane i receives column i, row q in element q.  EXEC must be all ones:
// every call site is in block-uniform control flow.
__device__ __forceinline__ u16x8 read_a(const uint16_t *slab, int s, int col) {
    const int lane = threadIdx.x & 63, i = lane & 15, q = i >> 2, pp = i & 3, h = lane >> 5;
    const uint16_t *at = slab + (16 * s + 8 * h + q) * PITCH + col + 16 * ((lane >> 4) & 1) + 4 * pp;
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(at));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(at + 4 * PITCH));
    return __builtin_bit_cast(u16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
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

// ------------------------------------------------------------------------------------------------ the common case
// Every network of the key has a rank that is a multiple of 32 and 16-byte aligned factors, cols % 8 == 0 and the key starts
// 4-element aligned in the arenas (every LoRA the scheduler stages; SDXL / FLUX linear and conv keys).
// The general path below is bound by latency and instruction issue, not by bytes or flops: it holds one k-chunk of
// prefetch in registers (231 VGPRs: two waves per SIMD), so every chunk waits out an L2 / HBM round trip, and it spends
// ~250 VALU + ~170 SALU per chunk on 64-bit addresses, tail guards and the fold against 8 MFMAs (rocprofv3 PMC: 4044
// VALU and 128 MFMA instructions per wave and tile, MFMA pipe 9 % busy).  Here:
//   * both factors go global -> LDS with global_load_lds_dwordx4 (no registers) through a ring of RING stages, RING - 1
//     chunks in flight per block; waves wait with a counted s_waitcnt vmcnt and meet at a raw s_barrier; the chunk sequence
//     of the key (DMA base addresses, fold flags) is a table in LDS made when the block enters the key;
//   * the LDS-DMA writes 64 lanes x 16 B contiguously, so layouts are made on the SOURCE side: the down slab keeps its
//     256-byte rows with the four 64-byte windows of row k XOR-ed by k & 3 (the transposing reads of 4 consecutive k-rows then
//     hit 4 different windows: conflict-free), the up slab is [k / 8][row] in 16-byte slots (a lane's B fragment is one
//     ds_read_b128, 16 consecutive rows per lane group);
//   * partial tiles read clamped rows / columns (the key's last row, its last 8 columns) instead of guarding every load:
//     pad, don't mask -- only the base loads and the stores are predicated;
//   * the fold is one packed fma per two accumulators when the scale is a power of two (see below); the base tile's loads are
//     issued behind the first stages and consumed at the first fold;
//   * finite checks: base and output through a packed fma(x, 0, poison) (NaN iff some x is inf / NaN), factors (edge tiles
//     only, read back from the stage) through a packed 16-bit max of the magnitudes.
// The whole-line base loads / output stores of the fast path are non-temporal (bit 0 loads, bit 1 stores; tuning switch): touched once,
// they should not push the factors out of the L2.  SDXL x16: 8.45 -> 8.22 ms; the 32-byte pieces of the accumulator layout, which
// rely on L2 to be merged into lines, took 11.4 ms this way.
#ifndef MECHA_LORA_NT
#define MECHA_LORA_NT 3
#endif
#if MECHA_LORA_NT & 1
#define LORA_BASE_LOAD(p) __builtin_nontemporal_load(p)
#else
#define LORA_BASE_LOAD(p) (*(p))
#endif
#if MECHA_LORA_NT & 2
#define LORA_OUT_STORE(v, p) __builtin_nontemporal_store(v, p)
#else
#define LORA_OUT_STORE(v, p) (*(p) = (v))
#endif
typedef uint16_t u16x8v __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
enum : int { NET_ENDS_GROUP = 2, NET_ONE_STEP = 4 };   // mecha_lora_net::reserved of the LDS copy
constexpr int RING = 4;
constexpr int STAGE_UP = 8192;                  // down: 32 rows x 256 B, then up: 128 rows x 64 B
constexpr int STAGE_BYTES = 16384;
constexpr int DMA_PER_STAGE = 4;                // per wave: two 1-KiB pieces of each factor
constexpr int MAX_CHUNKS = 128;                 // 32-deep k-chunks of one key, all networks together (beyond: the general path)
enum : uint32_t { CH_FOLD = 1, CH_ONE_STEP = 2 };
// One k-chunk of one network, ready for the DMA: made once per key and block (enter_key), read once per round.  The loop
// around the MFMAs then holds no network bookkeeping: no rank compare, no 64-bit multiply, no group logic.
struct __attribute__((aligned(16))) LoraChunk {
    const char *down;                           // row k0 of down
    const char *up;                             // column k0 of up
    uint32_t rank2;                             // bytes per row of up
    uint32_t flags;                             // CH_FOLD: the chunk closes a run of networks with equal (scale, weight)
    float scale, weight;
};
typedef __attribute__((address_space(3))) unsigned char lds_byte;

// The LDS-DMA instructions of one wave and stage: lane l's 16 bytes at `src` go to dst + 16 l (dst wave-uniform, in M0).
// Written as inline assembly on purpose: for the builtin the compiler cannot tell which LDS bytes the DMA writes, so it puts
// s_waitcnt vmcnt(0) in front of every later LDS read -- that drains the ring in every round.  Here the waits are the counted
// s_waitcnt vmcnt of the round itself; loads the compiler does not know about only make its own waits conservative (vmcnt
// retires in order).  M0 is compiler-reserved and not preserved around an asm statement: saved and put back in the statement.
__device__ __forceinline__ void glds16x4(const char *s0, const char *s1, const char *s2, const char *s3, uint32_t down_lds, uint32_t up_lds) {
    unsigned keep;
    const uint32_t d1 = down_lds + 1024u, u1 = up_lds + 1024u;
    asm volatile("s_mov_b32 %0, m0\n\t"
                 "s_mov_b32 m0, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"
                 "s_mov_b32 m0, %6\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\t"
                 "s_mov_b32 m0, %7\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, off\n\t"
                 "s_mov_b32 m0, %8\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, off\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(s0), "v"(s1), "v"(s2), "v"(s3), "s"(down_lds), "s"(d1), "s"(up_lds), "s"(u1) : "memory");
}
// running max over the magnitudes of 16-bit floats, two per word: v_and_b32 + v_pk_max_u16 per word, one accumulator
__device__ __forceinline__ uint32_t pk_max_u16(uint32_t a, uint32_t b) {
    uint32_t r;
    asm("v_pk_max_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ uint32_t pk_max_abs16(uint32_t m, u32x4 w) {
    const uint32_t k = 0x7fff7fffu;
    return pk_max_u16(pk_max_u16(m, pk_max_u16(w[0] & k, w[1] & k)), pk_max_u16(w[2] & k, w[3] & k));
}
// s_waitcnt vmcnt(n) for the counts the rounds use
__device__ __forceinline__ void wait_vm(int n) {
    if (n == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                   // the steady state: two later stages in flight
    else if (n == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// Block = 4 waves, block tile 128 x 128, wave tile 64 columns x 64 rows (2 x 2 MFMA tiles: 8 MFMAs per round and wave for
// 8 transposing reads + 4 ds_read_b128).  Two blocks per CU: one block's tile prologue / epilogue (a chain of memory
// latencies with nothing to overlap inside the block) runs beside the other block's rounds.
// Column permutation: the lane that gives the transposing read the address of k-row q, chunk p of a 16-lane group points at
// columns 16 (p & 1) + 8 gbit + 4 (p >> 1) of the 32-column window instead of 16 gbit + 4 p.  MFMA row index M = 8 g + 4 h + e of
// the accumulator layout is then output column 16 h + 4 g + e: a lane's 16 accumulators are 16 CONTIGUOUS output columns
// (16-byte base loads and output stores), and the 32 lanes of one LDS cycle still cover each 64-byte window exactly once.
template <int LDT>
__device__ __forceinline__ void lora_tile_fast(const LoraParams &p, const mecha_lora_key &key, const LoraChunk *chunks, int n_chunks, int key_index,
                                               int64_t tile_x, int64_t tile_y, unsigned char *ring) {
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int64_t bcol0 = tile_x * LORA_TILE, brow0 = tile_y * LORA_TILE;
    const int wcol = (wave & 1) * 64, wrow = (wave >> 1) * 64;
    const bool check_up = tile_x == 0, check_down = tile_y == 0;
    const char *base = static_cast<const char *>(p.base);
    const int64_t last_row = key.rows - 1;

    // ---- LDS-DMA source offsets of this lane (bytes).  Piece m = 2 wave + j (j = 0, 1) of each factor:
    //   down: rows 4m .. 4m+3 of the slab; lane = (row & 3) * 16 + 16-byte slot; slot s of row k holds the logical window
    //         (s >> 2) ^ (k & 3), i.e. columns 8 * ((((s >> 2) ^ (k & 3)) << 2) | (s & 3)) .. + 7 of the tile;
    //   up:   rows 16m .. 16m+15 of the tile, 64 B each (1 KiB contiguous in memory when the rank is 32: the DMA reads whole
    //         sectors; a [k-part][row] image with one row per lane asked for 16 of every 64 bytes, 64 requests per instruction).
    //         The image is [row][k-part] in 16-byte slots with the k-part XOR-ed by (row >> 2) & 3, so that the 16 lanes of a
    //         ds_read_b128 group (16 rows, one k-part) hit 16 different slots modulo 16.
    // Partial tiles read clamped rows / columns (pad, don't mask).
    const int drow = lane >> 4, dslot = lane & 15;
    int64_t dcol = bcol0 + 8 * ((((dslot >> 2) ^ drow) << 2) | (dslot & 3));
    if (dcol > key.cols - 8) dcol = key.cols - 8;
    const uint32_t d_lane0 = (uint32_t)(((int64_t)(8 * wave + drow) * key.cols + dcol) * 2);
    const uint32_t d_step = (uint32_t)(key.cols * 8);                                       // 4 rows further
    uint32_t urow[2];
#pragma unroll
    for (int j = 0; j < 2; j++) { const int64_t rr = brow0 + 16 * (2 * wave + j) + (lane >> 2); urow[j] = (uint32_t)(rr < last_row ? rr : last_row); }
    const uint32_t u_part = 16u * (uint32_t)((lane & 3) ^ ((lane >> 4) & 3));               // (row >> 2) & 3 of row 16 m + (lane >> 2)
    const uint32_t ring_lds = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lds_byte *)ring) + (uint32_t)wave * 2048u;

    // ---- LDS read offsets of this lane (bytes inside a stage)
    const int q = (lane & 15) >> 2, pp = lane & 3, gbit = (lane >> 4) & 1;
    const int a_lane = (8 * h + q) * 256 + 32 * (pp & 1) + 16 * gbit + 8 * (pp >> 1);       // + 16 s * 256 + window
    int a_win[2];
#pragma unroll
    for (int cg = 0; cg < 2; cg++) a_win[cg] = ((((wcol >> 5) + cg) ^ q) << 6);
    int b_lane[2];                                                                          // + 32 rg * 64
#pragma unroll
    for (int s = 0; s < 2; s++) b_lane[s] = STAGE_UP + (4 * (wrow + r) + ((2 * s + h) ^ ((r >> 2) & 3))) * 16;

    // ---- the ring
    uint32_t pflags[RING];
    float pscale[RING], pweight[RING];
    int next = 0;                                                         // chunks issued so far
    auto issue = [&](int slot) {
        const LoraChunk c = chunks[next];
        next++;
        pflags[slot] = c.flags; pscale[slot] = c.scale; pweight[slot] = c.weight;
        const char *d0 = c.down + d_lane0;
        const char *u0 = c.up + (urow[0] * c.rank2 + u_part);
        const char *u1 = c.up + (urow[1] * c.rank2 + u_part);
        const uint32_t stage = ring_lds + (uint32_t)slot * STAGE_BYTES;
        glds16x4(d0, d0 + d_step, u0, u1, stage, stage + STAGE_UP);
    };
    // prologue: RING - 1 stages in flight, then the base tile
#pragma unroll
    for (int i = 0; i < RING - 1; i++)
        if (next < n_chunks) issue(i);

    // ---- the base tile.  total[rg][cg] first holds the raw words (16-bit base: 8 dwords, f32: 16), then the fp32 values.
    // Rows / columns past the key's edge read clamped addresses and are zeroed when the words are unpacked.
    f32x16 total[2][2];
    bool in_row[2], in_col[2][2];
    int64_t row_at[2];                                                    // element index of this lane's rows, and its 8-column pieces inside a row
    int col_at[2][2];
#pragma unroll
    for (int rg = 0; rg < 2; rg++) {
        const int64_t row = brow0 + wrow + 32 * rg + r;
        in_row[rg] = row < key.rows;
        row_at[rg] = key.offset + (in_row[rg] ? row : last_row) * key.cols;
    }
#pragma unroll
    for (int cg = 0; cg < 2; cg++)
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int64_t c = bcol0 + wcol + 32 * cg + 16 * h + 8 * j;
            in_col[cg][j] = c < key.cols;                                 // cols % 8 == 0: a piece is inside or outside as a whole
            col_at[cg][j] = (int)(in_col[cg][j] ? c : key.cols - 8);
        }
    // the coalesced side of the 16-bit base / output tiles (see below): row 8 i + crow of a row group, 16-byte piece cpiece of the
    // wave's 128-byte row
    const int crow = lane >> 3, cpiece = lane & 7;
    const bool ccol_ok = bcol0 + wcol + 8 * cpiece < key.cols;
    const int64_t ccol_at = ccol_ok ? bcol0 + wcol + 8 * cpiece : key.cols - 8;
    if (base == nullptr) {
#pragma unroll
        for (int rg = 0; rg < 2; rg++)
#pragma unroll
            for (int cg = 0; cg < 2; cg++)
#pragma unroll
                for (int i = 0; i < 16; i++) total[rg][cg][i] = 0.0f;
    } else if (p.base_dt == MECHA_F32) {
#pragma unroll
        for (int rg = 0; rg < 2; rg++)
#pragma unroll
            for (int cg = 0; cg < 2; cg++)
#pragma unroll
                for (int j = 0; j < 2; j++)
#pragma unroll
                    for (int v = 0; v < 2; v++) {
                        const f32x4 w = *reinterpret_cast<const f32x4 *>(base + ((row_at[rg] + col_at[cg][j]) + 4 * v) * 4);
#pragma unroll
                        for (int e = 0; e < 4; e++) total[rg][cg][8 * j + 4 * v + e] = w[e];
                    }
    } else {
        // 16-bit base: a lane's accumulators want 16-byte pieces of 32 different rows, four instructions per 128-byte line.  The
        // loads are made the other way round -- instruction i of a row group: lane -> (row 8 i + lane / 8, piece lane % 8), whole
        // lines -- and turned into the accumulator layout through 4 KiB of LDS per wave (the ring stage the prologue leaves free;
        // pieces XOR-ed by the row so that both sides are conflict-free).  Only this wave touches its 4 KiB: no block barrier.
        unsigned char *xp = ring + (RING - 1) * STAGE_BYTES + wave * 4096;
#pragma unroll
        for (int rg = 0; rg < 2; rg++) {
            u32x4 tw[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int64_t row = brow0 + wrow + 32 * rg + 8 * i + crow;
                tw[i] = LORA_BASE_LOAD(reinterpret_cast<const u32x4 *>(base + (key.offset + (row < last_row ? row : last_row) * key.cols + ccol_at) * 2));
            }
#pragma unroll
            for (int i = 0; i < 4; i++) *reinterpret_cast<u32x4 *>(xp + (8 * i + crow) * 128 + ((cpiece ^ crow) << 4)) = tw[i];
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int cg = 0; cg < 2; cg++)
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    const u32x4 w = *reinterpret_cast<const u32x4 *>(xp + r * 128 + (((4 * cg + 2 * h + j) ^ (r & 7)) << 4));
#pragma unroll
                    for (int e = 0; e < 4; e++) total[rg][cg][4 * j + e] = __uint_as_float(w[e]);
                }
            __builtin_amdgcn_wave_barrier();
        }
    }
    f32x2 poison_in = {0.0f, 0.0f};
    const f32x2 zero2 = {0.0f, 0.0f};
    auto unpack_base = [&]() {                                            // raw words -> fp32 values, in place
#pragma unroll
        for (int rg = 0; rg < 2; rg++)
#pragma unroll
            for (int cg = 0; cg < 2; cg++) {
                float v[16];
                if (p.base_dt == MECHA_F32) {
#pragma unroll
                    for (int i = 0; i < 16; i++) v[i] = total[rg][cg][i];
                } else {
                    uint32_t w[8];
#pragma unroll
                    for (int i = 0; i < 8; i++) w[i] = __float_as_uint(total[rg][cg][i]);
                    if (p.base_dt == MECHA_BF16) {
#pragma unroll
                        for (int i = 0; i < 8; i++) { v[2 * i] = __uint_as_float(w[i] << 16); v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u); }
                    } else {
#pragma unroll
                        for (int i = 0; i < 8; i++) { v[2 * i] = f16_bits_to_f32(w[i] & 0xffffu); v[2 * i + 1] = f16_bits_to_f32(w[i] >> 16); }
                    }
                }
#pragma unroll
                for (int i = 0; i < 16; i++) v[i] = (in_row[rg] && in_col[cg][i >> 3]) ? v[i] : 0.0f;
#pragma unroll
                for (int i = 0; i < 16; i += 2) poison_in = __builtin_elementwise_fma(f32x2{v[i], v[i + 1]}, zero2, poison_in);
#pragma unroll
                for (int i = 0; i < 16; i++) total[rg][cg][i] = v[i];
            }
    };

    // Consumed here, not at the first fold: loads the compiler knows about, pending across the rounds, make it put its own
    // s_waitcnt vmcnt in front of whatever touches their registers -- in the unrolled ring that was a full drain every fourth
    // round.  The wait below costs one HBM latency per tile, beside the first stages' landing; the CU's other block computes.
    if (base != nullptr) unpack_base();
    uint32_t mag = 0u;                                                                      // running max of |factor| bit patterns, two per word (edge tiles)
    f32x16 acc[2][2];
#pragma unroll
    for (int rg = 0; rg < 2; rg++)
#pragma unroll
        for (int cg = 0; cg < 2; cg++)
#pragma unroll
            for (int i = 0; i < 16; i++) acc[rg][cg][i] = 0.0f;

    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    int done = 0;                                                         // chunks consumed so far
    // one round: wait for the stage in `slot`, refill the slot read in the previous round, multiply
    auto round = [&](int slot) {
        // vmcnt retires in order: allow the stages issued after this one
        wait_vm(DMA_PER_STAGE * (next - done - 1));
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const uint32_t flags = pflags[slot];
        const float scale = pscale[slot], weight = pweight[slot];
        if (next < n_chunks) issue((slot + RING - 1) % RING);
        const unsigned char *stage = ring + slot * STAGE_BYTES;
        u16x8 a[2][2], b[2][2];                                           // [k-step][column group], [k-step][row group]
#pragma unroll
        for (int s = 0; s < 2; s++) {
#pragma unroll
            for (int cg = 0; cg < 2; cg++) {
                const unsigned char *src = stage + a_lane + a_win[cg] + 16 * s * 256;
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(src));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(src + 4 * 256));
                a[s][cg] = __builtin_bit_cast(u16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
            }
#pragma unroll
            for (int rg = 0; rg < 2; rg++) b[s][rg] = *reinterpret_cast<const u16x8 *>(stage + b_lane[s] + 32 * rg * 64);
        }
        __builtin_amdgcn_sched_barrier(0);                                // all twelve reads in flight before the first MFMA waits
        // Consecutive networks with the same (scale, weight) share ONE accumulator chain and one fold (CH_FOLD closes the run):
        // sum_i fma(delta_i * s, w, .) and fma((sum_i delta_i) * s, w, .) differ by fp32 rounding of the partial sums only, inside
        // the parity bar of this method (BLAS-ordered sgemm in the reference: <= 1 ulp of the output dtype).  The chain always
        // accumulates; the accumulators are cleared where a run ends and at the start of the tile (C = 0 as an inline constant
        // in one branch made the compiler keep two accumulator register sets).
#pragma unroll
        for (int s = 0; s < 2; s++)
#pragma unroll
            for (int rg = 0; rg < 2; rg++)
#pragma unroll
                for (int cg = 0; cg < 2; cg++) acc[rg][cg] = mfma16<LDT>(a[s][cg], b[s][rg], acc[rg][cg]);
        if (check_down) {                                                 // this thread's 32 bytes of the down slab
            mag = pk_max_abs16(mag, *reinterpret_cast<const u32x4 *>(stage + t * 16));
            mag = pk_max_abs16(mag, *reinterpret_cast<const u32x4 *>(stage + 4096 + t * 16));
        }
        if (check_up) {
            mag = pk_max_abs16(mag, *reinterpret_cast<const u32x4 *>(stage + STAGE_UP + t * 16));
            mag = pk_max_abs16(mag, *reinterpret_cast<const u32x4 *>(stage + STAGE_UP + 4096 + t * 16));
        }
        if (flags & CH_FOLD) {                                            // the run is complete: delta * scale, then addcmul
            // fma(round(acc * scale), weight, total) (lycoris.py:263 then linear.py:76).  With a power-of-two scale (alpha / rank
            // usually is one) acc * scale and scale * weight are exact, so one fma gives the same bits; two per instruction
            // (v_pk_fma_f32).  Only results below the normal range could differ, where acc * scale would round and this does not.
            if (flags & CH_ONE_STEP) {
                const f32x2 sw = {scale * weight, scale * weight};
#pragma unroll
                for (int rg = 0; rg < 2; rg++)
#pragma unroll
                    for (int cg = 0; cg < 2; cg++)
#pragma unroll
                        for (int i = 0; i < 16; i += 2) {
                            const f32x2 f = __builtin_elementwise_fma(f32x2{acc[rg][cg][i], acc[rg][cg][i + 1]}, sw, f32x2{total[rg][cg][i], total[rg][cg][i + 1]});
                            total[rg][cg][i] = f.x; total[rg][cg][i + 1] = f.y;
                        }
            } else {
#pragma unroll
                for (int rg = 0; rg < 2; rg++)
#pragma unroll
                    for (int cg = 0; cg < 2; cg++)
#pragma unroll
                        for (int i = 0; i < 16; i++) total[rg][cg][i] = __builtin_fmaf(acc[rg][cg][i] * scale, weight, total[rg][cg][i]);
            }
#pragma unroll
            for (int rg = 0; rg < 2; rg++)
#pragma unroll
                for (int cg = 0; cg < 2; cg++)
#pragma unroll
                    for (int i = 0; i < 16; i++) acc[rg][cg][i] = 0.0f;
        }
        done++;
    };
    while (true) {
        if (done >= n_chunks) break;
        round(0);
        if (done >= n_chunks) break;
        round(1);
        if (done >= n_chunks) break;
        round(2);
        if (done >= n_chunks) break;
        round(3);
    }

    f32x2 poison_out = {0.0f, 0.0f};
    char *outp = static_cast<char *>(p.out);
    auto store_out = [&](auto dt_tag) {
        constexpr int ODT = decltype(dt_tag)::value;
        if constexpr (ODT != MECHA_F32) {
            // 16-bit output: the tile goes through LDS into whole-line stores (the mirror image of the base loads above).  Any ring
            // stage but the last one read is free now: every wave left its reads of the round before the last at that round's
            // barrier, and no DMA is in flight.
            unsigned char *xp = ring + ((n_chunks + RING - 2) % RING) * STAGE_BYTES + wave * 4096;
#pragma unroll
            for (int rg = 0; rg < 2; rg++) {
#pragma unroll
                for (int cg = 0; cg < 2; cg++) {
#pragma unroll
                    for (int i = 0; i < 16; i += 2) poison_out = __builtin_elementwise_fma(f32x2{total[rg][cg][i], total[rg][cg][i + 1]}, zero2, poison_out);
#pragma unroll
                    for (int j = 0; j < 2; j++) {
                        u32x4 w;
#pragma unroll
                        for (int e = 0; e < 4; e++) {
                            const float lo = total[rg][cg][8 * j + 2 * e], hi = total[rg][cg][8 * j + 2 * e + 1];
                            w[e] = ODT == MECHA_BF16 ? (f32_to_bf16_bits(lo) | (f32_to_bf16_bits(hi) << 16)) : (f32_to_f16_bits(lo) | (f32_to_f16_bits(hi) << 16));
                        }
                        *reinterpret_cast<u32x4 *>(xp + r * 128 + (((4 * cg + 2 * h + j) ^ (r & 7)) << 4)) = w;
                    }
                }
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const int64_t row = brow0 + wrow + 32 * rg + 8 * i + crow;
                    const u32x4 w = *reinterpret_cast<const u32x4 *>(xp + (8 * i + crow) * 128 + ((cpiece ^ crow) << 4));
                    if (row < key.rows && ccol_ok) LORA_OUT_STORE(w, reinterpret_cast<u32x4 *>(outp + (key.offset + row * key.cols + ccol_at) * 2));
                }
                __builtin_amdgcn_wave_barrier();
            }
            return;
        }
#pragma unroll
        for (int rg = 0; rg < 2; rg++)
#pragma unroll
            for (int cg = 0; cg < 2; cg++) {
#pragma unroll
                for (int i = 0; i < 16; i += 2) poison_out = __builtin_elementwise_fma(f32x2{total[rg][cg][i], total[rg][cg][i + 1]}, zero2, poison_out);
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    if (in_row[rg] && in_col[cg][j]) {
                        if constexpr (ODT == MECHA_F32) {
#pragma unroll
                            for (int v = 0; v < 2; v++) {
                                const int i0 = 8 * j + 4 * v;
                                *reinterpret_cast<f32x4 *>(outp + ((row_at[rg] + col_at[cg][j]) + 4 * v) * 4) =
                                    f32x4{total[rg][cg][i0], total[rg][cg][i0 + 1], total[rg][cg][i0 + 2], total[rg][cg][i0 + 3]};
                            }
                        } else {
                            u32x4 w;
#pragma unroll
                            for (int e = 0; e < 4; e++) {
                                const float lo = total[rg][cg][8 * j + 2 * e], hi = total[rg][cg][8 * j + 2 * e + 1];
                                w[e] = ODT == MECHA_BF16 ? (f32_to_bf16_bits(lo) | (f32_to_bf16_bits(hi) << 16)) : (f32_to_f16_bits(lo) | (f32_to_f16_bits(hi) << 16));
                            }
                            *reinterpret_cast<u32x4 *>(outp + (row_at[rg] + col_at[cg][j]) * 2) = w;
                        }
                    }
                }
            }
    };
    if (p.out_dt == MECHA_BF16) store_out(std::integral_constant<int, MECHA_BF16>{});
    else if (p.out_dt == MECHA_F16) store_out(std::integral_constant<int, MECHA_F16>{});
    else store_out(std::integral_constant<int, MECHA_F32>{});
    bool bad_in = (poison_in.x != poison_in.x) || (poison_in.y != poison_in.y);
    const bool bad_out = (poison_out.x != poison_out.x) || (poison_out.y != poison_out.y);
    const uint32_t inf_bits = LDT == MECHA_F16 ? 0x7c00u : 0x7f80u;
    bad_in |= ((mag & 0xffffu) >= inf_bits) | ((mag >> 16) >= inf_bits);
    if ((bad_in | bad_out) && p.flags != nullptr)
        atomicOr(p.flags + key_index, (bad_in ? MECHA_FLAG_INPUT_NONFINITE : 0u) | (bad_out ? MECHA_FLAG_OUTPUT_NONFINITE : 0u));
}

// the general path for one tile: any rank, any alignment (scalar edge accesses, every load guarded)
template <int LDT>
__device__ __forceinline__ void lora_tile_general(const LoraParams &p, const mecha_lora_key &key, const mecha_lora_net *nets, int key_index,
                                                  int64_t tile_x, int64_t tile_y, uint16_t (&lds)[2][SLAB_HALFS]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    // transposed product: MFMA rows <-> output columns, MFMA columns <-> output rows
    const int64_t bcol0 = tile_x * LORA_TILE;                             // block's first output column
    const int wcol = (wave & 1) * 64;                                     // this wave's columns inside the block tile
    const int64_t row0 = tile_y * LORA_TILE + (wave >> 1) * 64 + r;       // this lane's rows: row0 and row0 + 32
    const char *base = static_cast<const char *>(p.base);
    const bool vec4 = ((key.cols & 3) == 0) && ((key.offset & 3) == 0);   // rows start 4-element aligned (arenas are 16-B aligned)
    // every factor element is read by a whole row / column of tiles: the tiles on the key's first column check `up`,
    // those on its first row check `down`, once each
    const bool check_up = tile_x == 0, check_down = tile_y == 0;

    // total starts at the base value (fp32).  D^T layout of tile (rg, cg): reg g*4+q holds output column
    // bcol0 + wcol + 32 cg + 8g + 4h + q of row row0 + 32 rg.
    f32x16 total[2][2];
    bool bad_in = false;
#pragma unroll
    for (int rg = 0; rg < 2; rg++) {
        const int64_t row = row0 + 32 * rg;
#pragma unroll
        for (int cg = 0; cg < 2; cg++) {
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const int64_t c = bcol0 + wcol + 32 * cg + 8 * g + 4 * h;
                float v[4] = {0.0f, 0.0f, 0.0f, 0.0f};
                if (base != nullptr && row < key.rows && c < key.cols) {
                    const int valid = key.cols - c >= 4 ? 4 : (int)(key.cols - c);
                    load4(p.base_dt, vec4 && valid == 4, base, key.offset + row * key.cols + c, valid, v);
                }
#pragma unroll
                for (int q = 0; q < 4; q++) { bad_in |= nonfinite_f32(v[q]); total[rg][cg][g * 4 + q] = v[q]; }
            }
        }
    }

    // ---- the chunk sequence over (network, k0): the next chunk's global loads are in flight while the current one is
    // multiplied; one barrier per chunk (double-buffered LDS)
    struct Chunk { int n, k0; bool ok; };
    int n = 0, k0 = 0;
    const int n_nets = key.net_count;
    auto take = [&]() {
        Chunk c{n, k0, n < n_nets};
        if (c.ok) { k0 += KC; if (k0 >= nets[n].rank) { k0 = 0; n++; } }
        return c;
    };
    u16x8 dv[2], bv[2][2];                                                // down: two k-rows; up: [row group][k-step]
    auto fetch = [&](const Chunk &c) {
        if (!c.ok) return;
        const mecha_lora_net net = nets[c.n];
        const uint16_t *down = static_cast<const uint16_t *>(net.down), *up = static_cast<const uint16_t *>(net.up);
        load_down<LDT>(down, net.rank, key.cols, bcol0, c.k0, ((key.cols & 7) == 0) && ((reinterpret_cast<uintptr_t>(down) & 15u) == 0), check_down, dv, bad_in);
        const bool up_vec = ((net.rank & 7) == 0) && ((reinterpret_cast<uintptr_t>(up) & 15u) == 0);
#pragma unroll
        for (int rg = 0; rg < 2; rg++) load_up<LDT>(up, net.rank, row0 + 32 * rg, row0 + 32 * rg < key.rows, c.k0, h, up_vec, check_up, bv[rg], bad_in);
    };
    Chunk next = take();
    fetch(next);
    f32x16 acc[2][2];
#pragma unroll
    for (int rg = 0; rg < 2; rg++)
#pragma unroll
        for (int cg = 0; cg < 2; cg++)
#pragma unroll
            for (int i = 0; i < 16; i++) acc[rg][cg][i] = 0.0f;
    int buf = 0;
    while (next.ok) {
        const Chunk now = next;
        stage_down(lds[buf], dv);
        u16x8 b_now[2][2];
#pragma unroll
        for (int rg = 0; rg < 2; rg++) { b_now[rg][0] = bv[rg][0]; b_now[rg][1] = bv[rg][1]; }
        next = take();
        fetch(next);
        __syncthreads();                                                  // slab `buf` is complete; the other buffer's readers are done (barrier of the previous round)
        const int rank = nets[now.n].rank;
        {
            const u16x8 a0 = read_a(lds[buf], 0, wcol), a1 = read_a(lds[buf], 0, wcol + 32);
            if (now.k0 == 0) {                                            // a network's first k-step starts from C = 0 (an inline constant: no register clears)
                const f32x16 zero = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int rg = 0; rg < 2; rg++) {
                    acc[rg][0] = mfma16<LDT>(a0, b_now[rg][0], zero);
                    acc[rg][1] = mfma16<LDT>(a1, b_now[rg][0], zero);
                }
            } else {
#pragma unroll
                for (int rg = 0; rg < 2; rg++) {
                    acc[rg][0] = mfma16<LDT>(a0, b_now[rg][0], acc[rg][0]);
                    acc[rg][1] = mfma16<LDT>(a1, b_now[rg][0], acc[rg][1]);
                }
            }
        }
        if (now.k0 + 16 < rank) {
            const u16x8 a0 = read_a(lds[buf], 1, wcol), a1 = read_a(lds[buf], 1, wcol + 32);
#pragma unroll
            for (int rg = 0; rg < 2; rg++) {
                acc[rg][0] = mfma16<LDT>(a0, b_now[rg][1], acc[rg][0]);
                acc[rg][1] = mfma16<LDT>(a1, b_now[rg][1], acc[rg][1]);
            }
        }
        if (now.k0 + KC >= rank) {                                        // the network is complete: delta * scale, then addcmul
            const float scale = nets[now.n].scale, weight = nets[now.n].weight;
            // fma(round(acc * scale), weight, total) (lycoris.py:263 then linear.py:76).  With a power-of-two scale (alpha / rank
            // usually is one) acc * scale and scale * weight are exact, so one fma gives the same bits; two per instruction
            // (v_pk_fma_f32).  Only results below the normal range could differ, where acc * scale would round and this does not.
            const uint32_t sb = __float_as_uint(scale), swb = __float_as_uint(scale * weight);
            const bool one_step = ((sb & 0x007fffffu) == 0u) && (((sb >> 23) & 0xffu) - 1u < 254u) && (((swb >> 23) & 0xffu) - 1u < 254u);
            if (one_step) {
                const f32x2 sw = {scale * weight, scale * weight};
#pragma unroll
                for (int rg = 0; rg < 2; rg++)
#pragma unroll
                    for (int cg = 0; cg < 2; cg++)
#pragma unroll
                        for (int i = 0; i < 16; i += 2) {
                            const f32x2 t = __builtin_elementwise_fma(f32x2{acc[rg][cg][i], acc[rg][cg][i + 1]}, sw, f32x2{total[rg][cg][i], total[rg][cg][i + 1]});
                            total[rg][cg][i] = t.x; total[rg][cg][i + 1] = t.y;
                        }
            } else {
#pragma unroll
                for (int rg = 0; rg < 2; rg++)
#pragma unroll
                    for (int cg = 0; cg < 2; cg++)
#pragma unroll
                        for (int i = 0; i < 16; i++) total[rg][cg][i] = __builtin_fmaf(acc[rg][cg][i] * scale, weight, total[rg][cg][i]);
            }
        }
        buf ^= 1;
    }

    bool bad_out = false;
#pragma unroll
    for (int rg = 0; rg < 2; rg++) {
        const int64_t row = row0 + 32 * rg;
#pragma unroll
        for (int cg = 0; cg < 2; cg++) {
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const int64_t c = bcol0 + wcol + 32 * cg + 8 * g + 4 * h;
                if (row < key.rows && c < key.cols) {
                    float v[4];
#pragma unroll
                    for (int q = 0; q < 4; q++) { v[q] = total[rg][cg][g * 4 + q]; }
                    const int valid = key.cols - c >= 4 ? 4 : (int)(key.cols - c);
#pragma unroll
                    for (int q = 0; q < 4; q++) bad_out |= (q < valid) && nonfinite_f32(v[q]);
                    store4(p.out_dt, vec4 && valid == 4, p.out, key.offset + row * key.cols + c, valid, v);
                }
            }
        }
    }
    if ((bad_in | bad_out) && p.flags != nullptr)
        atomicOr(p.flags + key_index, (bad_in ? MECHA_FLAG_INPUT_NONFINITE : 0u) | (bad_out ? MECHA_FLAG_OUTPUT_NONFINITE : 0u));
}

// FAST: the kernel of the common case (its keys only); !FAST: the general path (the other keys only).  Two kernels rather
// than one branch: the fast path runs 8 waves per block, the general one 4, and together they do not fit the registers.
// Both are persistent: the grid is one resident set of blocks, dealt round-robin over the 8 XCDs (each with its own L2);
// XCD x works through the contiguous tile range [x, x + 1) * tiles_per_xcd, its blocks side by side (block j takes every
// nb-th tile), so a key's factors are fetched into one L2 and stay there while the key lasts (a contiguous run per block
// was measured first: 32 keys per XCD in flight, 83 MB of factors against 4 MB of L2, the factor reads went to HBM:
// 14.9 ms for the SDXL x16 batch).  A block walks the key table forward with a parallel search.  On a key change the key's network table goes to LDS with one parallel load
// (every later access is an LDS read: a global load inside the k-loop would have to wait out the LDS-DMA ring, vmcnt is
// in order) and the same threads evaluate the fast / general test.
constexpr int LORA_FAST_BLOCK = 256;            // the fast kernel: 4 waves, wave tile 64 x 64, two blocks per CU
// (round 2, on the 8-wave / 128 x 128 / one block per CU version: two of those blocks per CU -- __launch_bounds__(512, 4):
//  128 VGPRs, 26 of them spilled -- took 13.6 ms for the SDXL x16 batch against 11.5 ms)
template <int LDT, bool BATCHED, bool FAST>
__global__ __launch_bounds__(MECHA_BLOCK, 2) void lora_kernel(LoraParams p, int64_t n_tiles, int64_t tiles_per_xcd, int run) {
    __shared__ __attribute__((aligned(16))) unsigned char lds_raw[FAST ? RING * STAGE_BYTES : 2 * SLAB_HALFS * 2];
    __shared__ __attribute__((aligned(16))) mecha_lora_net s_nets[LORA_MAX_NETS];
    __shared__ LoraChunk s_chunks[FAST ? MAX_CHUNKS : 1];
    __shared__ int s_chunk_count;
    uint16_t (&lds)[2][SLAB_HALFS] = *reinterpret_cast<uint16_t (*)[2][SLAB_HALFS]>(lds_raw);
    // ---- this block's tiles: x0 + j, x0 + j + nb, ... of its XCD's range (the XCD's resident blocks work on neighbouring
    // tiles at any moment: one key's factors in the L2, not one key per block)
    const int64_t xcd = blockIdx.x & 7, j = blockIdx.x >> 3, nb = gridDim.x >> 3;
    const int64_t x0 = xcd * tiles_per_xcd, x1 = (x0 + tiles_per_xcd < n_tiles) ? x0 + tiles_per_xcd : n_tiles;
    if (x0 + j * run >= x1 && !(BATCHED && !FAST && p.n_general >= 0)) return;      // whole block
    mecha_lora_key key;
    int key_index = 0;
    int64_t key_end = 0;                                                  // first tile after the key
    bool fast = false;
    // last key at or after `from` whose tile_start is <= t.  One parallel load of a window of tile_starts (sorted) and a
    // block-wide count instead of a chain of dependent loads; the stride of the walk skips at most nb keys.
    auto seek_key = [&](int from, int64_t t) {
        while (true) {
            const int mine = from + (int)threadIdx.x;
            const int pred = mine < p.n_keys && p.keys[mine].tile_start <= t;
            const int n = __syncthreads_count(pred);
            if (n < (int)blockDim.x || from + n >= p.n_keys) return from + n - 1;
            from += n - 1;
        }
    };
    auto enter_key = [&](int ki) {                                        // block-uniform; called after a barrier
        key_index = ki;
        if constexpr (BATCHED) key = p.keys[ki]; else key = p.key0;
        const int64_t tiles_x = (key.cols + LORA_TILE - 1) / LORA_TILE, tiles_y = (key.rows + LORA_TILE - 1) / LORA_TILE;
        key_end = key.tile_start + tiles_x * tiles_y;
        int ok = ((key.cols & 7) == 0) && key.cols >= 8 && key.cols < ((int64_t)1 << 25) && ((key.offset & 7) == 0) && key.rows < ((int64_t)1 << 31);
        mecha_lora_net net{};
        if ((int)threadIdx.x < key.net_count) {
            if constexpr (BATCHED) net = p.nets[key.net_begin + threadIdx.x]; else net = p.net0[threadIdx.x];
            // group bits for the fast path: neighbours with the same (scale, weight) share an accumulator chain
            const int i = threadIdx.x;
            mecha_lora_net next = net;
            if constexpr (BATCHED) {
                if (i + 1 < key.net_count) next = p.nets[key.net_begin + i + 1];
            } else {
                if (i + 1 < key.net_count) next = p.net0[i + 1];
            }
            auto same = [](const mecha_lora_net &x, const mecha_lora_net &y) {
                return __float_as_uint(x.scale) == __float_as_uint(y.scale) && __float_as_uint(x.weight) == __float_as_uint(y.weight);
            };
            const uint32_t sb = __float_as_uint(net.scale), swb = __float_as_uint(net.scale * net.weight);
            const bool one_step = ((sb & 0x007fffffu) == 0u) && (((sb >> 23) & 0xffu) - 1u < 254u) && (((swb >> 23) & 0xffu) - 1u < 254u);
            net.reserved = ((i + 1 == key.net_count || !same(next, net)) ? NET_ENDS_GROUP : 0) | (one_step ? NET_ONE_STEP : 0);
            s_nets[threadIdx.x] = net;
            ok = ok && ((net.rank & 31) == 0) && (((reinterpret_cast<uintptr_t>(net.up) | reinterpret_cast<uintptr_t>(net.down)) & 15u) == 0) &&
                 (key.rows * (int64_t)net.rank < ((int64_t)1 << 30));
        }
        __syncthreads();
        // the key's chunk sequence: thread i writes the chunks of network i behind those of the networks before it
        if ((int)threadIdx.x < key.net_count) {
            int64_t first = 0;
            for (int m = 0; m < (int)threadIdx.x; m++) first += s_nets[m].rank >> 5;
            const int count = net.rank >> 5;
            if ((int)threadIdx.x + 1 == key.net_count) s_chunk_count = first + count <= MAX_CHUNKS ? (int)(first + count) : 0;
            if (first + count > MAX_CHUNKS) ok = 0;
            else if constexpr (FAST) {
                for (int c = 0; c < count; c++) {
                    LoraChunk ch;
                    ch.down = static_cast<const char *>(net.down) + (int64_t)c * KC * key.cols * 2;
                    ch.up = static_cast<const char *>(net.up) + c * KC * 2;
                    ch.rank2 = (uint32_t)net.rank * 2u;
                    ch.flags = (c + 1 == count && (net.reserved & NET_ENDS_GROUP) ? CH_FOLD : 0u) | ((net.reserved & NET_ONE_STEP) ? CH_ONE_STEP : 0u);
                    ch.scale = net.scale; ch.weight = net.weight;
                    s_chunks[first + c] = ch;
                }
            }
        } else if (threadIdx.x == 0) s_chunk_count = 0;                   // a key without networks
        fast = __syncthreads_and(ok) != 0;
    };
    // `run` consecutive tiles per block and turn: fewer key changes per block (a change is a chain of dependent loads and two
    // barriers), at the price of run x as many tiles -- and keys' factors -- in flight per XCD.  SDXL x16 rank-32 batch / one
    // network per key: run 1 9.41 / 4.28 ms, 2 9.23 / 4.13 (the default; MECHA_LORA_RUN), 4 9.65 / 4.02, 8 9.93 / 4.00
    if constexpr (BATCHED && !FAST) {
        // The few keys off the fast path, by the host's list: walking every tile of the batch to find them cost 0.5 ms of the
        // SDXL x16 batch's 9.9 (a key change per visited tile, each a chain of dependent loads and two barriers).
        if (p.n_general >= 0) {
            for (int g = 0; g < p.n_general; g++) {
                __syncthreads();
                enter_key(p.general_keys[g]);
                if (fast) continue;                                      // the kernel's own test decides
                const int64_t tiles_x = (key.cols + LORA_TILE - 1) / LORA_TILE;
                for (int64_t local = blockIdx.x; local < key_end - key.tile_start; local += gridDim.x) {
                    __syncthreads();
                    lora_tile_general<LDT>(p, key, s_nets, key_index, local % tiles_x, local / tiles_x, lds);
                }
            }
            return;
        }
    }
    for (int64_t t0 = x0 + j * run; t0 < x1; t0 += nb * run)
    for (int64_t t = t0; t < t0 + run && t < x1; t++) {
        __syncthreads();                                                  // the previous tile's readers of the LDS stages and of s_nets are done
        if (t >= key_end) {
            if constexpr (BATCHED) enter_key(seek_key(key_index, t)); else enter_key(0);
        }
        if (fast != FAST) continue;                                       // the other kernel's key
        const int64_t local = t - key.tile_start;
        const int64_t tiles_x = (key.cols + LORA_TILE - 1) / LORA_TILE;
        const int64_t tile_y = local / tiles_x, tile_x = local % tiles_x;
        if constexpr (FAST) lora_tile_fast<LDT>(p, key, s_chunks, s_chunk_count, key_index, tile_x, tile_y, lds_raw);
        else lora_tile_general<LDT>(p, key, s_nets, key_index, tile_x, tile_y, lds);
    }
}

// the host's copy of the kernel's test, for the launch decision
bool net_is_fast(const mecha_lora_net &net, int64_t rows) {
    return ((net.rank & 31) == 0) && (((reinterpret_cast<uintptr_t>(net.up) | reinterpret_cast<uintptr_t>(net.down)) & 15u) == 0) &&
           (rows * (int64_t)net.rank < ((int64_t)1 << 30));
}
bool key_is_fast(const mecha_lora_key &key) {
    return ((key.cols & 7) == 0) && key.cols >= 8 && key.cols < ((int64_t)1 << 25) && ((key.offset & 7) == 0) && key.rows < ((int64_t)1 << 31);
}

// blocks of `kernel` that fit one CU at once (asked once per kernel and device)
template <typename K>
int resident_blocks_per_cu(K kernel, int block) {
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel, block, 0) != hipSuccess || n < 1) { (void)hipGetLastError(); n = 1; }
    return n;
}

template <int LDT, bool BATCHED>
int launch_lora(const LoraParams &p, int64_t tiles, bool any_fast, bool any_general, hipStream_t s) {
    MECHA_REQUIRE(tiles + 8 <= 0x7fffffff, "too many tiles");
    const int64_t per_xcd = (tiles + 7) / 8;
    int dev = 0, cus = 256;
    MECHA_CHECK_HIP(hipGetDevice(&dev));
    MECHA_CHECK_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    static const int occ_fast = resident_blocks_per_cu(lora_kernel<LDT, BATCHED, true>, LORA_FAST_BLOCK);
    static const int occ_general = resident_blocks_per_cu(lora_kernel<LDT, BATCHED, false>, MECHA_BLOCK);
    static const int run = []() { const char *e = getenv("MECHA_LORA_RUN"); const int v = e ? atoi(e) : 2; return v >= 1 && v <= 64 ? v : 2; }();
    auto grid_for = [&](int occ) {
        int64_t per_xcd_blocks = ((int64_t)cus * occ + 7) / 8;
        if (per_xcd_blocks > per_xcd) per_xcd_blocks = per_xcd;
        return dim3((unsigned)(per_xcd_blocks * 8));
    };
    if (any_fast) MECHA_LAUNCH(MK_LORA, (lora_kernel<LDT, BATCHED, true>), grid_for(occ_fast), dim3(LORA_FAST_BLOCK), 0, s, p, tiles, per_xcd, run);
    if (any_general) MECHA_LAUNCH(MK_LORA, (lora_kernel<LDT, BATCHED, false>), grid_for(occ_general), dim3(MECHA_BLOCK), 0, s, p, tiles, per_xcd, 1);
    MECHA_CHECK_HIP(hipGetLastError());
    return MECHA_OK;
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
    p.n_general = -1;
    p.n_keys = 1;
    const int64_t tiles = ((cols + LORA_TILE - 1) / LORA_TILE) * ((rows + LORA_TILE - 1) / LORA_TILE);
    hipStream_t s = static_cast<hipStream_t>(stream);
    bool fast = key_is_fast(p.key0);
    int64_t chunks = 0;
    for (int i = 0; i < n && fast; i++) { fast = net_is_fast(p.net0[i], rows); chunks += p.net0[i].rank >> 5; }
    fast = fast && chunks <= MAX_CHUNKS;
    return lora_dtype == MECHA_F16 ? launch_lora<MECHA_F16, false>(p, tiles, fast, !fast, s) : launch_lora<MECHA_BF16, false>(p, tiles, fast, !fast, s);
}

int mecha_batched_lora_accumulate(const void *base_arena, void *out_arena, const mecha_lora_key *keys, const mecha_lora_key *keys_host,
                                  int n_keys, const mecha_lora_net *nets, const mecha_lora_net *nets_host, int base_dtype, int out_dtype,
                                  int lora_dtype, uint32_t *key_flags, void *stream) {
    MECHA_REQUIRE(n_keys >= 0, "n_keys must be >= 0");
    if (n_keys == 0) return MECHA_OK;
    if (int e = check_common(out_arena, base_dtype, out_dtype, lora_dtype, base_arena)) return e;
    MECHA_REQUIRE(keys != nullptr && keys_host != nullptr && nets != nullptr, "NULL key / network table");
    int64_t tiles = 0;
    bool any_fast = nets_host == nullptr, any_general = nets_host == nullptr;     // without the host copy both kernels run
    LoraParams p{};
    p.n_general = nets_host != nullptr ? 0 : -1;
    for (int i = 0; i < n_keys; i++) {
        const mecha_lora_key &k = keys_host[i];
        if (nets_host != nullptr) {
            bool fast = key_is_fast(k);
            int64_t chunks = 0;
            for (int j = 0; j < k.net_count && fast && k.net_begin >= 0; j++) {
                fast = net_is_fast(nets_host[k.net_begin + j], k.rows);
                chunks += nets_host[k.net_begin + j].rank >> 5;
            }
            fast = fast && chunks <= MAX_CHUNKS;
            any_fast |= fast; any_general |= !fast;
            if (!fast && p.n_general >= 0) {
                if (p.n_general < LORA_MAX_GENERAL) p.general_keys[p.n_general++] = i;
                else p.n_general = -1;                                   // too many for the list: the general kernel walks all tiles
            }
        }
        MECHA_REQUIRE(k.rows >= 1 && k.cols >= 1 && k.offset >= 0 && k.net_begin >= 0 && k.net_count >= 0 && k.net_count <= LORA_MAX_NETS,
                      "key %d: bad shape, offset or network range", i);
        MECHA_REQUIRE(k.tile_start == tiles, "key %d: tile_start must be the running sum of ceil(rows/%d) * ceil(cols/%d) (expected %lld)", i,
                      LORA_TILE, LORA_TILE, (long long)tiles);
        tiles += ((k.cols + LORA_TILE - 1) / LORA_TILE) * ((k.rows + LORA_TILE - 1) / LORA_TILE);
    }
    p.base = base_arena; p.out = out_arena; p.base_dt = base_dtype; p.out_dt = out_dtype; p.flags = key_flags;
    p.keys = keys; p.nets = nets; p.n_keys = n_keys;
    hipStream_t s = static_cast<hipStream_t>(stream);
    return lora_dtype == MECHA_F16 ? launch_lora<MECHA_F16, true>(p, tiles, any_fast, any_general, s)
                                   : launch_lora<MECHA_BF16, true>(p, tiles, any_fast, any_general, s);
}

}  // extern "C"
