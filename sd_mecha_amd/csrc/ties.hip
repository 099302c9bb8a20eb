// ties.hip -- TIES / DARE: exact top-k magnitude threshold by radix select, sign election, disjoint mean.
//
// Reference: extensions/builtin/merge_methods/ties.py:98-132 (ties_sum, ties_sum_deltas, filter_top_k),
// ties.py:10-50 (ties_sum_with_dropout), linear.py:101 (subtract), linear.py:76 (add_difference).
//
// The threshold of delta m is the kth-smallest |delta_m| (kth = max(int((1-k)*numel), 1), ties kept), exact.
// Two selection pipelines (one key, or every key of an arena in the batched form -- same kernels):
//
//  A. bracket select (default; two full passes over the inputs):
//       sample  : one workgroup per (key, delta) sorts 4096 sampled |delta| in LDS and picks a bracket [lo, hi] of bit
//                 patterns around the sample's estimate of the kth value (keys of <= 4096 elements are solved here)
//       count   : ONE pass over the inputs: exact counts of |delta| < lo and == lo in registers, and a 2048-bin
//                 LDS histogram of the few elements inside (lo, hi]
//       resolve : per (key, delta): kth is below the bracket / at lo / in bin b / above.  A bin one bit pattern wide is
//                 an exact threshold; otherwise the bin is "ambiguous" (expected ~4e-5 of the elements)
//       merge   : the fused merge pass with the provisional threshold = lower edge of the ambiguous bin; elements that
//                 fall inside an ambiguous bin are also appended to a small list
//       select  : per (key, delta) exact radix select of the remaining rank inside its list -> final threshold
//       fixup   : outputs of listed elements that turn out to lie below the final threshold are recomputed
//     Brackets that miss (kth outside [lo, hi], or an over-full bin: heavy ties) fall back, per (key, delta), to B; the
//     fallback kernels are always enqueued and exit at once when a device counter says nothing failed (no host sync).
//     DARE (keep masks): pairs whose kth value is one of the dropped zeros are settled from the masks alone (presolve); the others
//     take this same pipeline on delta * mask -- sample, count, collect and fixup read the keep bits.
//  B. radix select (fallback; DARE with the model-stock factor): hist<1> -> scan<1> -> hist<2> -> scan<2> -> hist<3> -> scan<3>,
//     most-significant-digit radix select on the fp32 bit pattern of |delta| (31 value bits split 11 / 11 / 9), LDS
//     histograms flushed with one global atomic per non-empty bin; three more passes over the inputs.
//
// merge recomputes the deltas, applies |delta| >= threshold, elects the sign, averages the agreeing deltas, and fuses
// add_difference + output cast + finite checks.  No fp32 delta is ever written to memory.
// Roofline: HBM.  Algorithmic bytes per element for the fused graph = in_size * (n_models + 1) + out_size (SURVEY.md
// §8d: 10 B for base + 3 models in bf16); the count pass re-reads the inputs (implementation traffic).
//
// Storage dtypes are runtime (wave-uniform switches around the 16-byte loads and stores); what is compiled in is the
// register capacity CAP (deltas held per lane: 2, 4 or 8) and the DARE mask path.
#include <stdlib.h>
#include <type_traits>
#include "common.h"

namespace {

constexpr int TIES_TILE = 4096;            // elements per tile: 256 lanes x 8 elements x 2 vectors
constexpr int TIES_BINS = 2048;
constexpr int TIES_GROUP = 8;              // most deltas handled per hist/count launch (LDS: 8 x 2048 x 4 B = 64 KiB)
constexpr int TIES_MAX_DELTAS = MECHA_MAX_MODELS;
constexpr int TIES_SAMPLE = 4096;          // sampled elements per (key, delta); keys up to this size are sorted whole
constexpr int AMB_BASE_CAP = 512;          // ambiguous-list capacity per (key, delta) = AMB_BASE_CAP + numel / AMB_DIV
constexpr int AMB_DIV = 2048;

enum : uint32_t { MODE_RESOLVED = 0, MODE_AMBIG = 1, MODE_FAILED = 2, MODE_PENDING = 3 };

struct SelectState {                       // per (key, delta), 64 bytes
    uint32_t prefix;                       // B: digits selected so far, in place
    uint32_t threshold_bits;               // final: fp32 bit pattern of the threshold
    unsigned long long kth_rem;            // B: rank still to find inside the matching prefix (1-indexed)
    unsigned long long kth;                // rank of the threshold among |delta| (1-indexed)
    uint32_t mode;
    uint32_t lo, hi, shift;                // A: bracket (inclusive bit patterns) and bin shift
    uint32_t amb_lo, amb_hi;               // A: the ambiguous bin (inclusive)
    uint32_t amb_rank, amb_count;          // A: rank inside the bin (1-indexed), elements in it
    unsigned long long amb_offset;         // A: first slot of this pair's ambiguous list
};
struct SelectCounts { unsigned long long below, eq_lo; };
struct TiesGlobals { uint32_t n_failed; uint32_t n_pending; unsigned long long n_amb; unsigned long long n_hit; };

// first element of the tile inside the arena, its key, its element count; the key's offset and size ride along (32 bytes) so that
// what a flat-grid block needs from the key -- the DARE kernel its mask bit addresses -- is one load away, not two dependent ones
struct TileRec { long long elem; int seg; int valid; long long seg_offset, seg_numel; };

struct TiesParams {
    const void *base;                      // nullptr: inputs are deltas
    const void *in[TIES_MAX_DELTAS];       // models (base mode) or deltas
    const uint32_t *keep_bits;             // DARE masks or nullptr
    int64_t mask_words;                    // words per delta row of keep_bits (row form)
    int mask_stream;                       // 1: keep_bits is ONE Bernoulli stream; delta m of a key of n elements uses bits [m n, (m + 1) n)
    int presolved;                         // 1: ties_dare_presolve_kernel ran: pairs it settled (threshold 0) are MODE_RESOLVED, the others MODE_PENDING
    int n_deltas;
    int in_dt, out_dt;
    int raw;                               // model_stock (ties.py:137-143): the stock kernels take the deltas as they are
    int vec_ok;                            // every pointer 16-byte aligned
    const mecha_segment *segs;             // batched: device segment table; else nullptr
    mecha_segment seg0;                    // per key: the one segment
    int n_segs;
    const long long *tile_start;           // batched: device prefix of tiles per segment [n_segs + 1]
    const struct TileRec *tile_rec;        // batched: one record per tile (flat-grid kernels); nullptr per key
    long long total_tiles;
    SelectState *state;                    // [n_segs * n_deltas]
    uint32_t *hist;                        // [n_segs * n_deltas * TIES_BINS]
    // bracket select
    int bracket;                           // 1: pipeline A is running (the radix kernels then serve only MODE_FAILED pairs)
    SelectCounts *counts;                  // [n_segs * n_deltas]
    uint32_t *fill;                        // [n_segs * n_deltas] slots used in each ambiguous list
    TiesGlobals *globals;
    long long *amb_elem;                   // ambiguous lists: element index inside the arena / tensor
    uint32_t *amb_bits;                    //                  |delta| bit pattern
    uint32_t *amb_sd;                      //                  owning (key, delta) pair
    long long *hit_elem;                   // hit queue: first element of a vector that held an ambiguous value
    uint32_t *hit_seg;                     //            its key
    unsigned long long amb_capacity;       // slots in each of the five arrays above
};

__device__ __forceinline__ mecha_segment get_segment(const TiesParams &p, int s) { return p.segs == nullptr ? p.seg0 : p.segs[s]; }

// A block walks a contiguous range of tiles; the cursor remembers the key the current tile belongs to
// and only searches again when the walk leaves that key.
struct TileCursor {
    int s = -1;
    long long first = 0, end = 0;          // tile range [first, end) of segment s
    mecha_segment seg{0, 0};
};
__device__ __forceinline__ void cursor_seek(const TiesParams &p, long long t, TileCursor &c) {
    if (c.s >= 0 && t < c.end) return;
    if (p.tile_start == nullptr) { c.s = 0; c.first = 0; c.end = p.total_tiles; c.seg = p.seg0; return; }
    if (c.s >= 0 && c.s + 1 < p.n_segs) {                       // common case: the walk entered the next key
        const long long next_end = p.tile_start[c.s + 2];
        if (t < next_end) { c.s += 1; c.first = c.end; c.end = next_end; c.seg = p.segs[c.s]; return; }
    }
    int lo = (c.s >= 0) ? c.s : 0, hi = p.n_segs - 1;
    while (lo < hi) {
        int mid = (lo + hi + 1) >> 1;
        if (p.tile_start[mid] <= t) lo = mid; else hi = mid - 1;
    }
    c.s = lo; c.first = p.tile_start[lo]; c.end = p.tile_start[lo + 1]; c.seg = p.segs[lo];
}

// 8 consecutive elements as fp32, dtype chosen at run time (wave-uniform).
__device__ __forceinline__ void load_rt(int dt, bool vec, const void *ptr, int64_t e, int valid, float (&v)[8]) {
    if (vec) {
        if (dt == MECHA_BF16) load8<MECHA_BF16, true>(ptr, e, v);      // streamed: far larger than any cache between passes
        else if (dt == MECHA_F16) load8<MECHA_F16, true>(ptr, e, v);
        else load8<MECHA_F32, true>(ptr, e, v);
    } else {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            float x = 0.0f;
            if (j < valid) x = (dt == MECHA_BF16) ? load1<MECHA_BF16>(ptr, e + j) : (dt == MECHA_F16) ? load1<MECHA_F16>(ptr, e + j) : load1<MECHA_F32>(ptr, e + j);
            v[j] = x;
        }
    }
}
// The same in two steps for the two hot kernels (count, merge): all the loads of a vector position are issued back to back
// as raw words, and each input is converted when it is consumed.  W32: 32-bit storage (two 16-byte words per lane).
template <bool W32> struct Raw8 { u32x4 w[W32 ? 2 : 1]; };
template <bool W32>
__device__ __forceinline__ Raw8<W32> load_raw(bool vec, const void *ptr, int64_t e, int valid) {
    Raw8<W32> r;
    if (vec) {
        if constexpr (W32) {
            const u32x4 *q = reinterpret_cast<const u32x4 *>(reinterpret_cast<const float *>(ptr) + e);
            r.w[0] = __builtin_nontemporal_load(q);
            r.w[1] = __builtin_nontemporal_load(q + 1);
        } else {
            r.w[0] = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(reinterpret_cast<const uint16_t *>(ptr) + e));
        }
    } else {
        uint32_t t[W32 ? 8 : 4];
#pragma unroll
        for (int j = 0; j < (W32 ? 8 : 4); j++) t[j] = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            if (j < valid) {
                if constexpr (W32) t[j] = reinterpret_cast<const uint32_t *>(ptr)[e + j];
                else t[j >> 1] |= (uint32_t)reinterpret_cast<const uint16_t *>(ptr)[e + j] << (16 * (j & 1));
            }
        }
        r.w[0] = u32x4{t[0], t[1], t[2], t[3]};
        if constexpr (W32) r.w[1] = u32x4{t[4], t[5], t[6], t[7]};
    }
    return r;
}
template <bool W32>
__device__ __forceinline__ void unpack_raw(int dt, const Raw8<W32> &r, float (&v)[8]) {
    if constexpr (W32) {
        v[0] = __uint_as_float(r.w[0].x); v[1] = __uint_as_float(r.w[0].y); v[2] = __uint_as_float(r.w[0].z); v[3] = __uint_as_float(r.w[0].w);
        v[4] = __uint_as_float(r.w[1].x); v[5] = __uint_as_float(r.w[1].y); v[6] = __uint_as_float(r.w[1].z); v[7] = __uint_as_float(r.w[1].w);
    } else {
        if (dt == MECHA_BF16) unpack16<MECHA_BF16>(r.w[0], v);
        else unpack16<MECHA_F16>(r.w[0], v);
    }
}

__device__ __forceinline__ float load1_rt(int dt, const void *ptr, int64_t e) {
    return (dt == MECHA_BF16) ? load1<MECHA_BF16>(ptr, e) : (dt == MECHA_F16) ? load1<MECHA_F16>(ptr, e) : load1<MECHA_F32>(ptr, e);
}
__device__ __forceinline__ void store1_rt(int dt, void *ptr, int64_t e, float v) {
    if (dt == MECHA_BF16) store1<MECHA_BF16>(ptr, e, v);
    else if (dt == MECHA_F16) store1<MECHA_F16>(ptr, e, v);
    else store1<MECHA_F32>(ptr, e, v);
}
__device__ __forceinline__ void store_rt(int dt, bool vec, void *ptr, int64_t e, int valid, const float (&v)[8]) {
    if (vec) {
        if (dt == MECHA_BF16) store8<MECHA_BF16, true>(ptr, e, v);
        else if (dt == MECHA_F16) store8<MECHA_F16, true>(ptr, e, v);
        else store8<MECHA_F32, true>(ptr, e, v);
    } else {
#pragma unroll
        for (int j = 0; j < 8; j++)
            if (j < valid) store1_rt(dt, ptr, e + j, v[j]);
    }
}

// DARE: delta * mask (ties.py:33).  `local` = element index inside the key (a multiple of 8), `numel` = the key's size.
// Row form: mask bit i of row m (rows start at word boundaries, so the 8 bits sit in one word).  Stream form: the reference
// seeds a fresh generator PER KEY (ties.py:29-31) and draws the deltas' masks one after the other (ties.py:33), so delta m of a
// key of n elements takes draws [m n, (m + 1) n) of the one stream every key shares: bit m n + i, at any bit offset.
__device__ __forceinline__ void apply_keep_bits(const TiesParams &p, int m, int64_t local, int64_t numel, float (&v)[8]) {
    const int64_t s = (p.mask_stream ? (int64_t)m * numel : (int64_t)m * p.mask_words * 32) + local;
    const uint32_t sh = (uint32_t)(s & 31);
    // both words loaded unconditionally (a load under a branch brings its own full wait); the second index only moves where the
    // 8 bits straddle a word: stream form only, and the stream ends with a pad word
    const uint32_t w0 = p.keep_bits[s >> 5], w1 = p.keep_bits[(s >> 5) + (sh > 24u ? 1 : 0)];
    const uint32_t bits8 = __builtin_amdgcn_alignbit(w1, w0, sh);
#pragma unroll
    for (int j = 0; j < 8; j++) v[j] = v[j] * (((bits8 >> j) & 1u) ? 1.0f : 0.0f);
}

// the keep bit of element `local` of delta m of a key of `numel` elements (see apply_keep_bits)
__device__ __forceinline__ bool keep_bit(const TiesParams &p, int m, int64_t local, int64_t numel) {
    const int64_t s = (p.mask_stream ? (int64_t)m * numel : (int64_t)m * p.mask_words * 32) + local;
    return ((p.keep_bits[s >> 5] >> (uint32_t)(s & 31)) & 1u) != 0u;
}

__device__ __forceinline__ unsigned long long ties_kth(double one_minus_k, int64_t numel) {
    long long kth = (long long)(one_minus_k * (double)numel);       // max(int((1 - k) * numel), 1)  (ties.py:129)
    return (unsigned long long)(kth < 1 ? 1 : kth);
}

// ------------------------------------------------------------------------------------------------ A: count
// Bin of a bit pattern inside the bracket (lo <= bits <= hi).  The bracket is cut into 1024 cells of 2^shift patterns,
// aligned to absolute multiples of 2^shift.  Each cell owns two bins, in value order: bin 2q holds the one pattern
// whose low `shift` bits are zero, bin 2q + 1 the rest of the cell.  Deltas of half-precision weights are multiples
// of a coarse power of two, so nearly all the mass (and the kth value) sits on patterns with many trailing zero bits:
// those get a bin of their own, which is an exact threshold however many elements tie there, while the odd bins
// collect only the sparse remainder, few enough for the ambiguous list.
// `t` = bits - lo, with lo a multiple of 2^shift (ties_sample_kernel aligns it).
__device__ __forceinline__ uint32_t bracket_bin(uint32_t t, uint32_t shift, uint32_t low_mask) {
    return ((t + low_mask) >> shift) + (t >> shift);              // ceil(t / 2^shift) + floor(t / 2^shift) = 2q + (remainder != 0)
}

// LDS layout of the count pass, per delta: COUNT_LOW words that count the elements below the bracket (per-lane words, summed at
// the flush), the TIES_BINS bins of the bracket, COUNT_HIGH write-only words for the elements above it.
constexpr int COUNT_LOW = 66, COUNT_HIGH = 66;
constexpr int COUNT_STRIDE = COUNT_LOW + TIES_BINS + COUNT_HIGH;
static_assert(COUNT_STRIDE % 2 == 0, "the count kernel folds word offsets into the bias as pairs of words");

// Clamped binning.  92 % of the elements lie outside the bracket, and a wave nearly always holds some that lie inside, so the
// pass is branch-free per element; what it costs is VALU issue (as long as the memory time).  Instead of testing
// "inside?" and selecting a dump word (compare, select, sign extraction and a per-lane below counter: 10.5 instructions per
// element and delta), the pattern is CLAMPED into [low_clamp, high_clamp] -- per-lane patterns a few cells outside the bracket,
// chosen so that every lane's clamp values fall into LDS words of their own -- and binned like any other: |x|, clamp, two shifted
// copies and an add-shift, 6-7 instructions.  The words the low clamps hit are real counters (the "below" count), those of the high
// clamps are never read.  Needs room for 33 cells of patterns on either side of the bracket and a bracket that ends on a cell
// boundary (ties_sample_kernel sees to it); brackets at the ends of the pattern range take the compare-and-select form.
//   t = pattern - (lo - 33 cells);  word = floor(t / cell) + ceil(t / cell)   (bracket_bin(), shifted by COUNT_LOW = 66 words)
//   (in the kernel t also carries the LDS address of the delta's words, as address / 8 cells: the sum, times 4, IS the address)
//   lane L: low clamp  = lo - (1 + L / 2) cells - (L & 1)  -> words 64 - L .. (distinct per lane; cells of one pattern: pairs of lanes share)
//           high clamp = hi + 1 + (L / 2) cells + (L & 1)  -> words 66 + bins + L
// (a + b) << 2 in one instruction: the byte offset of word a + b.  (The compiler distributes the shift over the sum: three.)
__device__ __forceinline__ uint32_t add_lshl2(uint32_t a, uint32_t b) {
    uint32_t r;
    asm("v_add_lshl_u32 %0, %1, %2, 2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
typedef __attribute__((address_space(3))) uint32_t lds_u32;
__device__ __forceinline__ void lds_count(uint32_t lds_byte_address) {               // += 1 on the LDS word at that address
    __hip_atomic_fetch_add(reinterpret_cast<lds_u32 *>(lds_byte_address), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ bool count_fast_ok(uint32_t lo, uint32_t hi, uint32_t shift) {
    const unsigned long long margin = (33ull << shift) + 1ull;
    // float clamps (see the kernel) must be normal numbers, and the high ones at most the infinity pattern; cells of at most 2^18
    // patterns, because the kernel folds the LDS address of the delta's histogram into `t` as a number of cells
    return shift <= 18u && (unsigned long long)lo >= margin + 0x00800000ull && (unsigned long long)hi + 1ull + margin <= 0x7f800000ull &&
           (((unsigned long long)hi + 1ull - lo) & ((1ull << shift) - 1ull)) == 0ull && ((unsigned long long)hi + 1ull - lo) <= (1024ull << shift);
}

// Three 16-bit deltas (the SDXL x4 job; the second group of a FLUX x8 stripe) fit 5 waves per SIMD.
// BASE: the inputs are models and a base is subtracted (else they are deltas).
#ifndef MECHA_COUNT_OCC3
#define MECHA_COUNT_OCC3 5                   // waves per SIMD of the three-delta 16-bit instantiation (tuning switch)
#endif
// Four fp32 streams (+ base) hold 2 x 5 x 8 registers of raw words in the two slots alone: at 4 waves per SIMD (128 registers) that
// instantiation spilled 344 of them; it gets 2 waves' worth.
#ifndef MECHA_COUNT_OCC4_F32
#define MECHA_COUNT_OCC4_F32 2               // (tuning switch)
#endif
constexpr int count_waves_per_simd(int cap, bool w32, bool dare) {
    return (cap == 3 && !w32 && !dare) ? MECHA_COUNT_OCC3 : ((cap == 4 && w32) ? MECHA_COUNT_OCC4_F32 : 4);
}
// DARE: delta * mask before the binning (ties.py:33): a dropped delta is a zero, below every bracket that has room for the clamps.
template <int CAP, bool W32, bool BASE, bool DARE = false>
__global__ __launch_bounds__(MECHA_BLOCK, count_waves_per_simd(CAP, W32, DARE)) void ties_count_kernel(TiesParams p, int d0, int nd) {
    if (DARE && p.presolved && p.globals->n_pending == 0) return;           // every pair was settled from its mask
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_hist[];     // nd * COUNT_STRIDE words
    for (int i = threadIdx.x; i < nd * COUNT_STRIDE; i += MECHA_BLOCK) lds_hist[i] = 0;
    __syncthreads();
    const long long chunk = (p.total_tiles + gridDim.x - 1) / gridDim.x;
    const long long t0 = (long long)blockIdx.x * chunk;
    const long long t1 = (t0 + chunk < p.total_tiles) ? t0 + chunk : p.total_tiles;
    const uint32_t lane = threadIdx.x & 63;
    TileCursor cur;
    uint32_t lo[CAP], width[CAP], shift[CAP], low_mask[CAP];          // block-uniform
    uint32_t lob[CAP];                                                 // fast form: lo - 33 cells - (LDS byte address of the delta's words / 8) cells
    const uint32_t lds_base = (uint32_t)(size_t)(lds_u32 *)lds_hist;   // 16-byte aligned; COUNT_STRIDE is even
    uint32_t low_clamp[CAP], high_clamp[CAP];                          // fast form: per-lane patterns
    bool active[CAP], fast[CAP];
#pragma unroll
    for (int d = 0; d < CAP; d++) { active[d] = fast[d] = false; lo[d] = width[d] = shift[d] = low_mask[d] = lob[d] = low_clamp[d] = high_clamp[d] = 0; }

    auto flush = [&](int s) {
        __syncthreads();
        const int wave = threadIdx.x >> 6;
        if (wave < nd) {                                                 // nd <= 4 = waves per block: wave d sums delta d's below words
            uint32_t *low = lds_hist + wave * COUNT_STRIDE;
            uint32_t tot = low[lane];
            low[lane] = 0;
            if (lane < COUNT_LOW - 64) { tot += low[64 + lane]; low[64 + lane] = 0; }
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) tot += __shfl_xor(tot, off, 64);
            if (tot && lane == 0) atomicAdd(&p.counts[(int64_t)s * p.n_deltas + d0 + wave].below, (unsigned long long)tot);
        }
#pragma unroll
        for (int d = 0; d < CAP; d++) {
            if (d < nd) {
                // bins past the bracket's last cell hold elements above it (the high clamps' words start there): zeroed, not reported
                const int limit = fast[d] ? (int)(2u * ((width[d] + 1u) >> shift[d])) : TIES_BINS;
                uint32_t *w = lds_hist + d * COUNT_STRIDE + COUNT_LOW;
                for (int b = threadIdx.x; b < TIES_BINS + COUNT_HIGH; b += MECHA_BLOCK) {
                    const uint32_t c = w[b];
                    if (c) {
                        if (b < limit) atomicAdd(p.hist + ((int64_t)s * p.n_deltas + d0 + d) * TIES_BINS + b, c);
                        w[b] = 0;
                    }
                }
            }
        }
        __syncthreads();
    };

    // Rolling pipeline over half tiles ("positions": 256 lanes x 8 elements): the loads of position q + 1 are issued BEFORE position q
    // is consumed, so a wave always has one position's loads in flight while it does another's arithmetic.
    // The load side runs one position ahead with a cursor of its own (`cur`); the compute side follows with the key it is in
    // (`comp_s`) and flushes the block's histograms when that changes.  Two register slots, addressed by compile-time index.
    int issue_s = -1;                                                    // DARE: the load side's key, and whether any of its pairs is pending
    bool issue_active = true;
    int valid_u[2], seg_u[2];
    int64_t local_u[2];                                                  // DARE: the position's first element inside its key (mask bit address)
    uint32_t mw_u[2][DARE ? CAP : 1][2];                                 // DARE: the mask words of the position, loaded with its data
    int64_t comp_numel = 0;                                              // DARE: size of the key the compute side is in
    Raw8<W32> rb_u[2], rv_u[2][CAP];
    auto issue = [&](long long q, auto slot_c) {
        constexpr int slot = decltype(slot_c)::value;
        const long long t = q >> 1;
        cursor_seek(p, t, cur);
        const mecha_segment seg = cur.seg;
        const int64_t local = (t - cur.first) * TIES_TILE + (int64_t)(q & 1) * (TIES_TILE / 2) + (int64_t)threadIdx.x * 8;
        const int valid = (local >= seg.numel) ? 0 : ((seg.numel - local >= 8) ? 8 : (int)(seg.numel - local));
        if constexpr (DARE) local_u[slot] = local;
        valid_u[slot] = valid;
        seg_u[slot] = cur.s;
        if (valid == 0) return;
        const bool vec = p.vec_ok && valid == 8;
        const int64_t e = seg.offset + local;
        if constexpr (BASE) rb_u[slot] = load_raw<W32>(vec, p.base, e, valid);
#pragma unroll
        for (int d = 0; d < CAP; d++)
            if (d < nd) rv_u[slot][d] = load_raw<W32>(vec, p.in[d0 + d], e, valid);
        if constexpr (DARE) {
            // with the data, not when it is consumed: a load issued there would wait behind the next position's (vmcnt is in order)
#pragma unroll
            for (int d = 0; d < CAP; d++)
                if (d < nd) {
                    const int64_t sb = (p.mask_stream ? (int64_t)(d0 + d) * seg.numel : (int64_t)(d0 + d) * p.mask_words * 32) + local;
                    mw_u[slot][d][0] = p.keep_bits[sb >> 5];
                    mw_u[slot][d][1] = p.keep_bits[(sb >> 5) + (((sb & 31) > 24) ? 1 : 0)];      // unconditional load, see apply_keep_bits
                }
        }
    };
    int comp_s = -1;
    auto consume = [&](auto slot_c) {
        constexpr int slot = decltype(slot_c)::value;
        const int s_now = seg_u[slot];                                   // block-uniform
        if (s_now != comp_s) {
            if (comp_s >= 0) flush(comp_s);
            comp_s = s_now;
            if constexpr (DARE) comp_numel = get_segment(p, s_now).numel;
#pragma unroll
            for (int d = 0; d < CAP; d++) {
                if (d < nd) {
                    const SelectState &st = p.state[(int64_t)s_now * p.n_deltas + d0 + d];
                    active[d] = st.mode == MODE_PENDING;
                    lo[d] = st.lo; width[d] = st.hi - st.lo; shift[d] = st.shift; low_mask[d] = (1u << st.shift) - 1u;
                    fast[d] = count_fast_ok(st.lo, st.hi, st.shift);
                    lob[d] = st.lo - ((33u + lds_base / 8u + (uint32_t)d * (COUNT_STRIDE / 2)) << st.shift);
                    const uint32_t odd = st.shift ? (lane & 1u) : 0u;
                    low_clamp[d] = st.lo - (((lane >> 1) + 1u) << st.shift) - odd;
                    high_clamp[d] = st.hi + 1u + ((lane >> 1) << st.shift) + odd;
                }
            }
        }
        bool any = false;
#pragma unroll
        for (int d = 0; d < CAP; d++) any |= (d < nd) && active[d];
        const int valid = valid_u[slot];
        if (!any || valid == 0) return;                                  // every pair of this key was solved by the sample / past the key
        float b[8];
        if constexpr (BASE) unpack_raw<W32>(p.in_dt, rb_u[slot], b);
        const bool ragged = __ballot(valid < 8) != 0ull;                 // wave-uniform: some lane is at the end of the key
#pragma unroll
        for (int d = 0; d < CAP; d++) {
            if (d < nd && active[d]) {
                float v[8];
                unpack_raw<W32>(p.in_dt, rv_u[slot][d], v);
                if constexpr (BASE) {
#pragma unroll
                    for (int j = 0; j < 8; j++) v[j] = v[j] - b[j];                         // subtract (linear.py:101)
                }
                if constexpr (DARE) {                                                       // delta * mask (ties.py:33), see apply_keep_bits
                    const int64_t sb = (p.mask_stream ? (int64_t)(d0 + d) * comp_numel : (int64_t)(d0 + d) * p.mask_words * 32) + local_u[slot];
                    const uint32_t sh = (uint32_t)(sb & 31);
                    const uint32_t bits8 = __builtin_amdgcn_alignbit(mw_u[slot][d][1], mw_u[slot][d][0], sh);
#pragma unroll
                    for (int j = 0; j < 8; j++) v[j] = ((bits8 >> j) & 1u) ? v[j] : 0.0f;
                }
                uint32_t *hist_d = lds_hist + d * COUNT_STRIDE;                              // the compare-and-select form indexes it
                if (fast[d] && !ragged) {
                    if constexpr (BASE) {
                        // |x| and the clamp in one v_med3_f32: patterns of non-negative floats order like the floats, the clamps are
                        // normal numbers (count_fast_ok), and a delta computed here is never a signalling NaN.  A quiet NaN (from
                        // non-finite inputs: the key is flagged and dropped) comes out as the low clamp.
                        const float lc = __uint_as_float(low_clamp[d]), hc = __uint_as_float(high_clamp[d]);
#pragma unroll
                        for (int j = 0; j < 8; j++) {
                            const uint32_t t = __float_as_uint(__builtin_amdgcn_fmed3f(__builtin_fabsf(v[j]), lc, hc)) - lob[d];
                            lds_count(add_lshl2(t >> shift[d], (t + low_mask[d]) >> shift[d]));
                        }
                    } else {
#pragma unroll
                        for (int j = 0; j < 8; j++) {                                           // the same on the integer patterns
                            const int a = (int)(__float_as_uint(v[j]) & 0x7fffffffu);
                            const int c = min(max(a, (int)low_clamp[d]), (int)high_clamp[d]);
                            const uint32_t t = (uint32_t)c - lob[d];
                            lds_count(add_lshl2(t >> shift[d], (t + low_mask[d]) >> shift[d]));
                        }
                    }
                } else {
                    if (ragged) {
#pragma unroll
                        for (int j = 0; j < 8; j++)
                            if (j >= valid) v[j] = __uint_as_float(0x7fffffffu);                  // past the key: above every bracket
                    }
#pragma unroll
                    for (int j = 0; j < 8; j++) {
                        const uint32_t tt = (__float_as_uint(v[j]) & 0x7fffffffu) - lo[d];         // |delta| pattern relative to lo
                        // below the bracket: this lane's counting word; inside: its bin; above: this lane's write-only word
                        const uint32_t bin = COUNT_LOW + bracket_bin(tt, shift[d], low_mask[d]);
                        const uint32_t out = (tt >> 31) ? lane : COUNT_LOW + TIES_BINS + lane;
                        atomicAdd(&hist_d[tt <= width[d] ? bin : out], 1u);
                    }
                }
            }
        }
    };
    const long long q0 = 2 * t0, q1 = 2 * t1;
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    if constexpr (DARE) {
        // most pairs of a DARE job are settled from their masks: the walk jumps over the keys none of whose pairs is pending
        // (their data is not read, their positions are not visited)
        long long qi = q0;
        auto skip = [&]() {
            while (qi < q1) {
                cursor_seek(p, qi >> 1, cur);
                if (cur.s != issue_s) {
                    issue_s = cur.s;
                    issue_active = false;
#pragma unroll
                    for (int d = 0; d < CAP; d++)
                        if (d < nd) issue_active |= p.state[(int64_t)cur.s * p.n_deltas + d0 + d].mode == MODE_PENDING;
                }
                if (issue_active) break;
                qi = 2 * cur.end < q1 ? 2 * cur.end : q1;
            }
        };
        skip();
        bool have0 = qi < q1, have1 = false;
        if (have0) issue(qi++, S0{});
        while (have0) {
            skip();
            have1 = qi < q1;
            if (have1) issue(qi++, S1{});
            consume(S0{});
            skip();
            have0 = qi < q1;
            if (have0) issue(qi++, S0{});
            if (have1) consume(S1{});
        }
    } else {
        if (q0 < q1) issue(q0, S0{});
        for (long long q = q0; q < q1; q += 2) {                         // q0 and q1 are even: positions come in pairs
            issue(q + 1, S1{});
            consume(S0{});
            if (q + 2 < q1) issue(q + 2, S0{});
            consume(S1{});
        }
    }
    if (comp_s >= 0) flush(comp_s);
}

// ------------------------------------------------------------------------------------------------ A: resolve
// exclusive prefix of the 2048 bins over the block; returns (bin, count below it) of the bin holding `rank`, bin = -1 if none
__device__ __forceinline__ void find_bin(const uint32_t *h, unsigned long long rank, int &bin_out, unsigned long long &below_out,
                                         uint32_t &count_out, unsigned long long *wave_tot, int *sel, unsigned long long *sel_below,
                                         uint32_t *sel_count) {
    constexpr int PER = TIES_BINS / MECHA_BLOCK;
    uint32_t c[PER];
    unsigned long long local = 0;
#pragma unroll
    for (int j = 0; j < PER; j++) { c[j] = h[threadIdx.x * PER + j]; local += c[j]; }
    unsigned long long incl = local;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        unsigned long long n = __shfl_up(incl, off, 64);
        if (lane >= off) incl += n;
    }
    if (lane == 63) wave_tot[wave] = incl;
    if (threadIdx.x == 0) *sel = -1;
    __syncthreads();
    unsigned long long before = incl - local;
    for (int w = 0; w < wave; w++) before += wave_tot[w];
    if (rank > before && rank <= before + local) {                  // exactly one thread's bin range contains the rank
        unsigned long long cum = before;
#pragma unroll
        for (int j = 0; j < PER; j++) {
            if (rank > cum && rank <= cum + c[j]) { *sel = threadIdx.x * PER + j; *sel_below = cum; *sel_count = c[j]; }
            cum += c[j];
        }
    }
    __syncthreads();
    bin_out = *sel; below_out = *sel_below; count_out = *sel_count;
}

// ------------------------------------------------------------------------------------------------ A: sample
// Value of 0-indexed rank `rank` among the block's 4096 sampled bit patterns (16 per thread, in registers; 0xffffffff = padding,
// never selected: ranks stay below the number of real samples): MSD radix select in three passes of 11 / 11 / 9 bits through one
// 2048-bin LDS histogram.  Replaces a bitonic sort of the 4096 patterns in LDS (78 compare-exchange stages with a barrier each:
// 0.75 ms for the 7 587 (key, delta) pairs of the SDXL x4 job, VALU- and LDS-issue-bound) -- only two order statistics are needed.
__device__ __forceinline__ uint32_t sample_select(const uint32_t (&k)[16], uint32_t rank, uint32_t *hist, unsigned long long *wave_tot, int *sel,
                                                  unsigned long long *sel_below, uint32_t *sel_count) {
    uint32_t prefix = 0u, mask = 0x80000000u;
    unsigned long long r = (unsigned long long)rank + 1ull;              // find_bin takes 1-indexed ranks
    const int shifts[3] = {20, 9, 0};
    const uint32_t nbins[3] = {2048u, 2048u, 512u};
#pragma unroll
    for (int pass = 0; pass < 3; pass++) {
        for (int i = threadIdx.x; i < TIES_BINS; i += MECHA_BLOCK) hist[i] = 0;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 16; i++)
            if ((k[i] & mask) == prefix) atomicAdd(&hist[(k[i] >> shifts[pass]) & (nbins[pass] - 1u)], 1u);
        __syncthreads();
        int bin; unsigned long long below; uint32_t count;
        find_bin(hist, r, bin, below, count, wave_tot, sel, sel_below, sel_count);
        if (bin < 0) bin = 0;                                            // cannot happen: rank < number of samples under the prefix
        prefix |= (uint32_t)bin << shifts[pass];
        mask |= (nbins[pass] - 1u) << shifts[pass];
        r -= below;
        __syncthreads();                                                 // find_bin's shared words are reused by the next pass
    }
    return prefix;
}

// One workgroup per (key, delta): TIES_SAMPLE sampled |delta| bit patterns, the bracket from two of their order statistics.
__global__ __launch_bounds__(MECHA_BLOCK) void ties_sample_kernel(TiesParams p, double one_minus_k, float *thresholds_out) {
    __shared__ uint32_t hist[TIES_BINS];
    __shared__ unsigned long long wave_tot[MECHA_BLOCK / 64];
    __shared__ int sel;
    __shared__ unsigned long long sel_below;
    __shared__ uint32_t sel_count;
    const int64_t sd = blockIdx.x;
    const int s = (int)(sd / p.n_deltas), d = (int)(sd % p.n_deltas);
    if (p.presolved && p.state[sd].mode == MODE_RESOLVED) return;        // DARE: settled from its mask alone (threshold 0)
    const mecha_segment seg = get_segment(p, s);
    const bool whole = seg.numel <= TIES_SAMPLE;
    uint32_t k[16];
    if (whole) {
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const int idx = threadIdx.x + i * MECHA_BLOCK;
            uint32_t bits = 0xffffffffu;                                  // padding: beyond every rank
            if (idx < seg.numel) {
                float x = load1_rt(p.in_dt, p.in[d], seg.offset + idx);
                if (p.base != nullptr) x = x - load1_rt(p.in_dt, p.base, seg.offset + idx);
                if (p.keep_bits != nullptr) {                             // DARE: delta * mask (ties.py:33)
                    const int64_t sb = (p.mask_stream ? (int64_t)d * seg.numel : (int64_t)d * p.mask_words * 32) + idx;
                    if (((p.keep_bits[sb >> 5] >> (sb & 31)) & 1u) == 0u) x = 0.0f;
                }
                bits = __float_as_uint(x) & 0x7fffffffu;
            }
            k[i] = bits;
        }
    } else {
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int vpos = 2 * threadIdx.x + u;                         // 512 stratified runs of 8 contiguous elements
            const int64_t local = (int64_t)(((unsigned long long)vpos * (unsigned long long)seg.numel) / 512ull) & ~7ll;
            const int64_t e = seg.offset + local;
            float v[8], b[8];
            load_rt(p.in_dt, p.vec_ok != 0, p.in[d], e, 8, v);
            if (p.base != nullptr) {
                load_rt(p.in_dt, p.vec_ok != 0, p.base, e, 8, b);
#pragma unroll
                for (int j = 0; j < 8; j++) v[j] = v[j] - b[j];
            }
            if (p.keep_bits != nullptr) apply_keep_bits(p, d, local, seg.numel, v);
#pragma unroll
            for (int j = 0; j < 8; j++) k[u * 8 + j] = __float_as_uint(v[j]) & 0x7fffffffu;
        }
    }
    // block-uniform: every thread derives the same ranks
    SelectState st{};
    st.kth = ties_kth(one_minus_k, seg.numel);
    if (whole) {
        const uint32_t t = sample_select(k, (uint32_t)(st.kth - 1), hist, wave_tot, &sel, &sel_below, &sel_count);
        st.mode = MODE_RESOLVED;
        st.threshold_bits = t;
        if (threadIdx.x == 0 && thresholds_out != nullptr) thresholds_out[sd] = __uint_as_float(t);
    } else {
        const double q = ((double)st.kth - 0.5) / (double)seg.numel;
        const double r = q * TIES_SAMPLE;
        const double margin = 6.0 * sqrt(TIES_SAMPLE * q * (1.0 - q)) + 16.0;
        const long long lo_rank = (long long)floor(r - margin), hi_rank = (long long)ceil(r + margin);
        st.mode = MODE_PENDING;
        st.lo = (lo_rank < 0) ? 0u : sample_select(k, (uint32_t)lo_rank, hist, wave_tot, &sel, &sel_below, &sel_count);
        st.hi = (hi_rank >= TIES_SAMPLE) ? 0x7f800000u : sample_select(k, (uint32_t)hi_rank, hist, wave_tot, &sel, &sel_below, &sel_count);
        if (st.hi > 0x7f800000u) st.hi = 0x7f800000u;                     // NaN patterns stay above every bracket
        if (st.lo > st.hi) st.lo = st.hi;
        // bins over lo <= bits <= hi, see bracket_bin(): 1024 cells of 2^shift patterns, two bins per cell
        const uint32_t span = st.hi - st.lo;
        st.shift = (span >= 1024u) ? (uint32_t)(32 - __clz((int)span)) - 10u : 0u;      // span < 2^(shift + 10)
        st.lo = (st.lo >> st.shift) << st.shift;                          // cells start at lo
        const unsigned long long top = ((((unsigned long long)(st.lo >> st.shift)) + 1024ull) << st.shift) - 1ull;
        if (top < st.hi) st.hi = (uint32_t)top;                           // trims less than one cell off the bracket
        // the bracket ends with its last cell (hi + 1 - lo a whole number of cells): what the count pass's clamped binning needs
        // (count_fast_ok); widening by less than a cell costs nothing.  Not past the infinity pattern: such brackets stay ragged.
        const unsigned long long cell_end = (((unsigned long long)st.hi >> st.shift) << st.shift) + ((1ull << st.shift) - 1ull);
        if (cell_end <= 0x7f800000ull) st.hi = (uint32_t)cell_end;
    }
    if (threadIdx.x == 0) {
        p.state[sd] = st;
        p.counts[sd] = SelectCounts{0ull, 0ull};
        p.fill[sd] = 0u;
    }
}

// amb_fails: a pair whose kth value lies in a mixed bin is not listed for the select / fixup kernels but handed to the radix passes
// like a miss (the DARE form, whose fixup kernels do not exist)
__global__ __launch_bounds__(MECHA_BLOCK) void ties_resolve_kernel(TiesParams p, float *thresholds_out, int force_fail, int amb_fails = 0) {
    __shared__ unsigned long long wave_tot[MECHA_BLOCK / 64];
    __shared__ int sel;
    __shared__ unsigned long long sel_below;
    __shared__ uint32_t sel_count;
    const int64_t sd = blockIdx.x;
    SelectState st = p.state[sd];
    if (st.mode != MODE_PENDING) return;
    uint32_t *h = p.hist + sd * TIES_BINS;
    const SelectCounts sc = p.counts[sd];
    const int s = (int)(sd / p.n_deltas);
    const int64_t numel = get_segment(p, s).numel;
    const unsigned long long cap = (unsigned long long)AMB_BASE_CAP + (unsigned long long)numel / AMB_DIV;
    uint32_t mode = MODE_FAILED;
    int bin = -1; unsigned long long below = 0; uint32_t count = 0;
    if (!force_fail && st.kth > sc.below) {
        if (st.kth <= sc.below + sc.eq_lo) { mode = MODE_RESOLVED; st.threshold_bits = st.lo; }
        else {
            find_bin(h, st.kth - sc.below - sc.eq_lo, bin, below, count, wave_tot, &sel, &sel_below, &sel_count);
            if (bin >= 0) {
                const uint32_t cell = (((st.lo >> st.shift) + ((uint32_t)bin >> 1)) << st.shift);   // see bracket_bin()
                if ((bin & 1) == 0) { mode = MODE_RESOLVED; st.threshold_bits = cell; }
                else {
                    st.amb_lo = (cell + 1u > st.lo) ? cell + 1u : st.lo;
                    const unsigned long long top = (unsigned long long)cell + ((1ull << st.shift) - 1ull);
                    st.amb_hi = (top > st.hi) ? st.hi : (uint32_t)top;
                    st.amb_rank = (uint32_t)(st.kth - sc.below - sc.eq_lo - below);
                    st.amb_count = count;
                    if (st.amb_lo == st.amb_hi) { mode = MODE_RESOLVED; st.threshold_bits = st.amb_lo; }
                    else if (count <= cap && !amb_fails) mode = MODE_AMBIG;
                }
            }
        }
    }
    for (int i = threadIdx.x; i < TIES_BINS; i += MECHA_BLOCK) h[i] = 0;      // ready for the fallback / the next call
    if (threadIdx.x == 0) {
        st.mode = mode;
        if (mode == MODE_RESOLVED && thresholds_out != nullptr) thresholds_out[sd] = __uint_as_float(st.threshold_bits);
        if (mode == MODE_FAILED) atomicAdd(&p.globals->n_failed, 1u);
        p.state[sd] = st;
    }
}

// amb_offset = exclusive prefix of amb_count over the ambiguous pairs (one block)
__global__ __launch_bounds__(1024) void ties_amb_offsets_kernel(TiesParams p, int64_t n_sd) {
    __shared__ unsigned long long part[1024];
    const int64_t per = (n_sd + 1023) / 1024;
    const int64_t lo = (threadIdx.x * per < n_sd) ? threadIdx.x * per : n_sd, hi = (lo + per < n_sd) ? lo + per : n_sd;
    unsigned long long sum = 0;
    for (int64_t i = lo; i < hi; i++) sum += (p.state[i].mode == MODE_AMBIG) ? p.state[i].amb_count : 0u;
    part[threadIdx.x] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long run = 0;
        for (int i = 0; i < 1024; i++) { unsigned long long v = part[i]; part[i] = run; run += v; }
        p.globals->n_amb = run;
    }
    __syncthreads();
    unsigned long long run = part[threadIdx.x];
    for (int64_t i = lo; i < hi; i++) {
        if (p.state[i].mode == MODE_AMBIG) { p.state[i].amb_offset = run; run += p.state[i].amb_count; }
    }
}

// ------------------------------------------------------------------------------------------------ B: hist / scan
template <int PASS> struct PassCfg;
template <> struct PassCfg<1> { static constexpr int shift = 20, bins = 2048; [[maybe_unused]] static constexpr uint32_t prefix_mask = 0u; };
template <> struct PassCfg<2> { static constexpr int shift = 9, bins = 2048; static constexpr uint32_t prefix_mask = 0x7ff00000u; };
template <> struct PassCfg<3> { static constexpr int shift = 0, bins = 512; static constexpr uint32_t prefix_mask = 0x7ffffe00u; };

template <int PASS, int CAP, bool DARE>
__global__ __launch_bounds__(MECHA_BLOCK) void ties_hist_kernel(TiesParams p, int d0, int nd) {
    using C = PassCfg<PASS>;
    if (p.bracket && p.globals->n_failed == 0) return;             // nothing fell back
    if (p.presolved && p.globals->n_pending == 0) return;          // DARE: every pair was settled from its mask
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_hist[];     // nd * TIES_BINS words
    for (int i = threadIdx.x; i < nd * TIES_BINS; i += MECHA_BLOCK) lds_hist[i] = 0;
    __syncthreads();

    const long long chunk = (p.total_tiles + gridDim.x - 1) / gridDim.x;
    const long long t0 = (long long)blockIdx.x * chunk;
    const long long t1 = (t0 + chunk < p.total_tiles) ? t0 + chunk : p.total_tiles;
    TileCursor cur;

    // Exact zeros are counted per lane in registers and added to bin 0 at the flush: a DARE-masked delta is 90 % zeros (and deltas
    // of fine-tunes hold runs of them), and 64 lanes adding to ONE LDS word serialise (30.8 ms per pass for SDXL x3 with masks
    // against ~5 ms for dense deltas).  A zero can only match the all-zero prefix, whose bin in every pass is 0.
    uint32_t zeros[CAP];
#pragma unroll
    for (int d = 0; d < CAP; d++) zeros[d] = 0;
    auto flush = [&](int s) {
#pragma unroll
        for (int d = 0; d < CAP; d++) {
            if (d < nd) {                                               // block-uniform
                uint32_t tot = zeros[d];
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) tot += __shfl_xor(tot, off, 64);
                if (tot && (threadIdx.x & 63) == 0) {
                    atomicAdd(&lds_hist[d * TIES_BINS], tot);
                    // pass 1 also keeps the number of exact zeros per pair: a rank inside them IS the threshold 0 (ties_scan_kernel<1>)
                    if constexpr (PASS == 1) atomicAdd(&p.counts[(int64_t)s * p.n_deltas + d0 + d].eq_lo, (unsigned long long)tot);
                }
            }
            zeros[d] = 0;
        }
        __syncthreads();
        for (int i = threadIdx.x; i < nd * TIES_BINS; i += MECHA_BLOCK) {
            const uint32_t c = lds_hist[i];
            if (c) {
                const int d = i / TIES_BINS, b = i % TIES_BINS;
                atomicAdd(p.hist + ((int64_t)s * p.n_deltas + d0 + d) * TIES_BINS + b, c);
                lds_hist[i] = 0;
            }
        }
        __syncthreads();
    };

    uint32_t want[CAP];
    bool active[CAP];
#pragma unroll
    for (int d = 0; d < CAP; d++) { want[d] = 0; active[d] = false; }
    for (long long t = t0; t < t1; t++) {
        const int prev = cur.s;
        cursor_seek(p, t, cur);
        if (cur.s != prev) {
            if (prev >= 0) flush(prev);
#pragma unroll
            for (int d = 0; d < CAP; d++) {
                if (d < nd) {
                    const SelectState &st = p.state[(int64_t)cur.s * p.n_deltas + d0 + d];
                    // pipeline A: only the pairs whose bracket missed; pipeline B: every pair (pass 1), then those pass 1 did not settle
                    active[d] = p.bracket ? st.mode == MODE_FAILED : ((PASS == 1 && !p.presolved) || st.mode != MODE_RESOLVED);
                    if constexpr (PASS > 1) want[d] = st.prefix;
                }
            }
        }
        bool any = false;
#pragma unroll
        for (int d = 0; d < CAP; d++) any |= (d < nd) && active[d];
        if (!any) { t = cur.end - 1; continue; }                     // nothing to find in this key: on to the next one
        const mecha_segment seg = cur.seg;
#pragma unroll 1
        for (int u = 0; u < 2; u++) {
            const int64_t local = (t - cur.first) * TIES_TILE + (int64_t)u * (TIES_TILE / 2) + (int64_t)threadIdx.x * 8;
            if (local >= seg.numel) continue;
            const int valid = (seg.numel - local >= 8) ? 8 : (int)(seg.numel - local);
            const bool vec = p.vec_ok && valid == 8;
            const int64_t e = seg.offset + local;
            float b[8];
            if (p.base != nullptr) load_rt(p.in_dt, vec, p.base, e, valid, b);
            float v[CAP][8];
#pragma unroll
            for (int d = 0; d < CAP; d++)
                if (d < nd && active[d]) load_rt(p.in_dt, vec, p.in[d0 + d], e, valid, v[d]);
#pragma unroll
            for (int d = 0; d < CAP; d++) {
                if (d < nd && active[d]) {
                    if (p.base != nullptr) {
#pragma unroll
                        for (int j = 0; j < 8; j++) v[d][j] = v[d][j] - b[j];              // subtract (linear.py:101)
                    }
                    if constexpr (DARE) apply_keep_bits(p, d0 + d, local, seg.numel, v[d]);
#pragma unroll
                    for (int j = 0; j < 8; j++) {
                        const uint32_t bits = __float_as_uint(v[d][j]) & 0x7fffffffu;        // |delta|
                        bool hit = j < valid;
                        if constexpr (PASS > 1) hit = hit && ((bits & C::prefix_mask) == want[d]);
                        if (hit && bits == 0u) zeros[d] += 1u;
                        else if (hit) atomicAdd(&lds_hist[d * TIES_BINS + ((bits >> C::shift) & (C::bins - 1))], 1u);
                    }
                }
            }
        }
    }
    if (cur.s >= 0) flush(cur.s);
}

// One block per (key, delta): find the bin holding rank kth_rem, update the state, zero the histogram.
template <int PASS>
__global__ __launch_bounds__(MECHA_BLOCK) void ties_scan_kernel(TiesParams p, double one_minus_k, float *thresholds_out) {
    using C = PassCfg<PASS>;
    __shared__ unsigned long long wave_tot[MECHA_BLOCK / 64];
    __shared__ int sel;
    __shared__ unsigned long long sel_below;
    __shared__ uint32_t sel_count;
    if (p.bracket && p.globals->n_failed == 0) return;
    if (p.presolved && p.globals->n_pending == 0) return;
    const int64_t sd = blockIdx.x;
    const int s = (int)(sd / p.n_deltas);
    uint32_t *h = p.hist + sd * TIES_BINS;
    SelectState st = p.state[sd];
    if (p.bracket && st.mode != MODE_FAILED) return;
    if constexpr (PASS > 1) {
        if (!p.bracket && st.mode == MODE_RESOLVED) return;                  // settled by pass 1 (its histogram is already clear)
    } else {
        if (!p.bracket && p.presolved && st.mode == MODE_RESOLVED) return;   // settled from the masks alone
    }
    if constexpr (PASS == 1) {
        st.kth = ties_kth(one_minus_k, get_segment(p, s).numel);
        st.prefix = 0; st.threshold_bits = 0; st.kth_rem = st.kth;
        if (!p.bracket) st.mode = MODE_PENDING;
    }
    int bin; unsigned long long below; uint32_t count;
    find_bin(h, st.kth_rem, bin, below, count, wave_tot, &sel, &sel_below, &sel_count);
    for (int i = threadIdx.x; i < TIES_BINS; i += MECHA_BLOCK) h[i] = 0;     // ready for the next pass / the next call
    if constexpr (PASS == 1) {
        // the kth-smallest |delta| is an exact zero (a DARE-masked delta is mostly zeros): that is the threshold, the other two
        // passes have nothing to find for this pair
        if (!p.bracket && st.kth <= p.counts[sd].eq_lo) {
            if (threadIdx.x == 0) {
                st.mode = MODE_RESOLVED; st.threshold_bits = 0; st.prefix = 0;
                if (thresholds_out != nullptr) thresholds_out[sd] = 0.0f;
                p.state[sd] = st;
            }
            return;
        }
    }
    if (threadIdx.x == 0) {
        if (bin < 0) { bin = 0; below = 0; }                     // unreachable for kth <= numel; keeps the state defined
        st.prefix |= ((uint32_t)bin) << C::shift;
        st.kth_rem -= below;
        if constexpr (PASS == 3) {
            st.threshold_bits = st.prefix;
            st.mode = MODE_RESOLVED;
            if (thresholds_out != nullptr) thresholds_out[sd] = __uint_as_float(st.prefix);
        }
        p.state[sd] = st;
    }
}

// ------------------------------------------------------------------------------------------------ DARE: thresholds from the masks alone
// A DARE-masked delta holds at least (numel - kept) exact zeros, kept = set bits of its mask.  If the rank kth lies inside them,
// the kth-smallest |delta * mask| IS 0 (ties.py:33, :128-132) and no pass over the deltas is needed to know it: with the usual
// settings (probability 0.9, k 0.2 .. 1.0: kth <= 0.8 numel <= dropped) that settles every pair from 1/16 of a byte per element.
// One block per (key, delta); the pairs it cannot settle go through the radix passes.
constexpr int64_t PRESOLVE_SLICE_WORDS = 1 << 16;          // mask words per block (2 M elements of one delta)
// counts[sd].below += set bits of this block's slice of the pair's mask
__global__ __launch_bounds__(MECHA_BLOCK) void ties_dare_presolve_kernel(TiesParams p) {
    const int64_t sd = blockIdx.x;
    const int s = (int)(sd / p.n_deltas), m = (int)(sd % p.n_deltas);
    const int64_t numel = get_segment(p, s).numel;
    const int64_t b0 = p.mask_stream ? (int64_t)m * numel : (int64_t)m * p.mask_words * 32, b1 = b0 + numel;    // bit range
    const int64_t w0 = b0 >> 5, w1 = (b1 + 31) >> 5;
    const int64_t lo = w0 + (int64_t)blockIdx.y * PRESOLVE_SLICE_WORDS;
    if (lo >= w1) return;
    const int64_t hi = lo + PRESOLVE_SLICE_WORDS < w1 ? lo + PRESOLVE_SLICE_WORDS : w1;
    unsigned long long kept = 0;
    for (int64_t w = lo + threadIdx.x; w < hi; w += MECHA_BLOCK) {
        uint32_t word = p.keep_bits[w];
        if (w == w0) word &= ~0u << (b0 & 31);
        if (w == w1 - 1 && (b1 & 31)) word &= (1u << (b1 & 31)) - 1u;
        kept += __popc(word);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) kept += __shfl_down(kept, off, 64);
    if ((threadIdx.x & 63) == 0 && kept) atomicAdd(&p.counts[sd].below, kept);
}

__global__ __launch_bounds__(MECHA_BLOCK) void ties_dare_presolve_decide_kernel(TiesParams p, int64_t n_sd, double one_minus_k, float *thresholds_out) {
    const int64_t sd = (int64_t)blockIdx.x * MECHA_BLOCK + threadIdx.x;
    if (sd >= n_sd) return;
    const int64_t numel = get_segment(p, (int)(sd / p.n_deltas)).numel;
    SelectState st{};
    st.kth = ties_kth(one_minus_k, numel);
    st.kth_rem = st.kth;
    if (st.kth <= (unsigned long long)numel - p.counts[sd].below) {
        st.mode = MODE_RESOLVED;
        if (thresholds_out != nullptr) thresholds_out[sd] = 0.0f;
    } else {
        st.mode = MODE_PENDING;
        atomicAdd(&p.globals->n_pending, 1u);
    }
    p.counts[sd].below = 0;
    p.state[sd] = st;
}

// ------------------------------------------------------------------------------------------------ merge
struct MergeExtra {
    float alpha;            // add_difference weight (base mode)
    float rescale_div;      // DARE: divide by (1 - probability), or 1
    int vote_sgn;
    int has_rescale;
    int skip_full_tiles;    // 1: ties_merge_full_kernel runs too and owns the full tiles of the keys it accepts
    int full_any_thr;       // 1: ... of EVERY key (its DARE form needs no threshold >= 2^-90)
    float rescale_rcp;      // RN(1 / rescale_div)
    int fast_rescale;       // the three-instruction division by rescale_div is safe (a normal divisor, significand not all ones)
};

__device__ __forceinline__ float sign_f(float x) { return (float)((x > 0.0f) - (x < 0.0f)); }   // torch.sign: sign(+-0) = 0

// provisional threshold of a pair: the final one, or the lower edge of its ambiguous bin
__device__ __forceinline__ float prov_threshold(const SelectState &st) {
    return __uint_as_float(st.mode == MODE_AMBIG ? st.amb_lo : st.threshold_bits);
}

// The merge pass only notes WHICH 8-element vectors held a value inside an ambiguous bin (hit queue: a handful of
// entries per key).  This kernel re-reads those vectors and appends the ambiguous values to their pairs' lists.
__global__ __launch_bounds__(MECHA_BLOCK) void ties_amb_collect_kernel(TiesParams p) {
    const unsigned long long n = p.globals->n_hit < p.amb_capacity ? p.globals->n_hit : p.amb_capacity;
    const int lane = threadIdx.x & 63;
    // every lane of a wave takes the same number of turns (the slots are handed out wave by wave, below)
    for (unsigned long long i0 = (unsigned long long)blockIdx.x * MECHA_BLOCK; i0 < n; i0 += (unsigned long long)gridDim.x * MECHA_BLOCK) {
        const unsigned long long i = i0 + threadIdx.x;
        const bool live = i < n;
        const int s = live ? (int)p.hit_seg[i] : 0;
        const int64_t e = live ? p.hit_elem[i] : 0;
        const mecha_segment seg = get_segment(p, s);
        const int64_t left = seg.offset + seg.numel - e;
        const int valid = !live ? 0 : (left >= 8 ? 8 : (int)left);
        for (int m = 0; m < p.n_deltas; m++) {
            const int64_t sd = (int64_t)s * p.n_deltas + m;
            const SelectState &st = p.state[sd];
            const bool amb = live && st.mode == MODE_AMBIG;
            uint32_t vals[8], match = 0u;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                vals[j] = 0u;
                if (amb && j < valid) {
                    float x = load1_rt(p.in_dt, p.in[m], e + j);
                    if (p.base != nullptr) x = x - load1_rt(p.in_dt, p.base, e + j);
                    if (p.keep_bits != nullptr && !keep_bit(p, m, e + j - seg.offset, seg.numel)) x = 0.0f;      // DARE: delta * mask
                    vals[j] = __float_as_uint(x) & 0x7fffffffu;
                    if (vals[j] >= st.amb_lo && vals[j] <= st.amb_hi) match |= 1u << j;
                }
            }
            // One atomic per wave and pair, not one per value: continuous-valued fp32 deltas put ~a thousand values into a pair's
            // list, and a thousand atomics on ONE word took 0.8 ms (a single key of 4 x 134 M fp32 deltas: as long as its merge pass).
            const int cnt = __builtin_popcount(match);
            const int sd_key = cnt > 0 ? (int)sd : -1;                   // (pairs < 2^31: the workspace holds a SelectState for each)
            uint32_t first = 0u;
            unsigned long long todo = __ballot(cnt > 0);
            while (todo) {
                const int leader = __ffsll((long long)todo) - 1;
                const int sd_l = __shfl(sd_key, leader, 64);
                const bool mine = sd_key == sd_l;
                const int v = mine ? cnt : 0;
                int incl = v;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const int o = __shfl_up(incl, off, 64);
                    if (lane >= off) incl += o;
                }
                const int total = __shfl(incl, 63, 64);
                uint32_t b = 0u;
                if (lane == leader) b = atomicAdd(&p.fill[sd_l], (uint32_t)total);
                b = (uint32_t)__shfl((int)b, leader, 64);
                if (mine) first = b + (uint32_t)(incl - v);
                todo &= ~__ballot(mine);
            }
            if (cnt > 0) {
                uint32_t slot = first;
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    if ((match >> j) & 1u) {
                        const unsigned long long at = st.amb_offset + slot;
                        if (slot < st.amb_count && at < p.amb_capacity) {    // (slot >= amb_count cannot happen: the count pass saw the same elements)
                            p.amb_elem[at] = e + j; p.amb_bits[at] = vals[j]; p.amb_sd[at] = (uint32_t)sd;
                        }
                        slot++;
                    }
                }
            }
        }
    }
}

template <int CAP, bool DARE, bool W32>
__global__ __launch_bounds__(MECHA_BLOCK) void ties_merge_kernel(TiesParams p, MergeExtra x, void *out, uint32_t *flags) {
    const long long chunk = (p.total_tiles + gridDim.x - 1) / gridDim.x;
    const long long t0 = (long long)blockIdx.x * chunk;
    const long long t1 = (t0 + chunk < p.total_tiles) ? t0 + chunk : p.total_tiles;
    const bool in_regs = p.n_deltas <= CAP;        // one sweep over the inputs; otherwise the second sweep re-reads them
    // correctly rounded 1 / count for the division of the disjoint mean, see `fast_div` below ([0] = +inf)
    __shared__ float recip[MECHA_MAX_MODELS + 1];
    // hits of this block, pushed to the global queue with ONE atomic when the block is done: continuous-valued fp32 deltas give
    // thousands of hits per key, and as many atomics on the queue's one counter bounded the whole pass (0.86 ms for 4 x 134 M)
    constexpr int HQ = 128;
    __shared__ long long hq_elem[HQ];
    __shared__ uint32_t hq_seg[HQ];
    __shared__ uint32_t hq_n;
    __shared__ unsigned long long hq_base;
    if (threadIdx.x <= MECHA_MAX_MODELS) recip[threadIdx.x] = 1.0f / (float)threadIdx.x;
    if (threadIdx.x == 0) hq_n = 0u;
    __syncthreads();
    TileCursor cur;
    float thr[CAP];
    float amb_hi[CAP];                             // upper edge of the ambiguous bin as a float, -1 = none
    bool fast_div = false;
#pragma unroll
    for (int d = 0; d < CAP; d++) { thr[d] = 0.0f; amb_hi[d] = -1.0f; }
    for (long long t = t0; t < t1; t++) {
        const int prev = cur.s;
        cursor_seek(p, t, cur);
        const SelectState *st = p.state + (int64_t)cur.s * p.n_deltas;
        if (cur.s != prev && in_regs) {
            fast_div = true;
#pragma unroll
            for (int d = 0; d < CAP; d++) {
                if (d < p.n_deltas) {
                    thr[d] = prov_threshold(st[d]);
                    amb_hi[d] = (st[d].mode == MODE_AMBIG) ? __uint_as_float(st[d].amb_hi) : -1.0f;
                    // every kept |delta| of this key is >= its threshold, so a non-zero sum of agreeing deltas is too:
                    // far from the denormal range, where the fma-based division below stays exact
                    fast_div = fast_div && (thr[d] >= 0x1p-90f);
                }
            }
        }
        const mecha_segment seg = cur.seg;
        // the same test as in ties_merge_full_kernel: that kernel merges every full tile of this key.  Only the key's last tile can
        // be ragged: the walk jumps to it (or past the key) instead of visiting 400 full tiles per block one by one
        if (x.skip_full_tiles && in_regs && (fast_div || x.full_any_thr)) {
            const long long last = cur.end - 1;
            const bool last_ragged = (cur.end - cur.first) * TIES_TILE > seg.numel;
            if (t < last || !last_ragged) { t = (last_ragged ? last : cur.end) - 1; continue; }
        }
#pragma unroll 1
        for (int u = 0; u < 2; u++) {
            const int64_t local = (t - cur.first) * TIES_TILE + (int64_t)u * (TIES_TILE / 2) + (int64_t)threadIdx.x * 8;
            if (local >= seg.numel) continue;
            const int valid = (seg.numel - local >= 8) ? 8 : (int)(seg.numel - local);
            const bool vec = p.vec_ok && valid == 8;
            const int64_t e = seg.offset + local;
            float b[8];
            f32x2 poison = {0.0f, 0.0f};           // fma(x, 0, poison): stays 0 while every input is finite, NaN after inf / NaN (v_pk_fma_f32)
            bool hit = false;                      // some |delta| of this lane lies in an ambiguous bin
            // every load of this position is issued before the first one is consumed
            Raw8<W32> rb, rv[CAP];
            if (p.base != nullptr) rb = load_raw<W32>(vec, p.base, e, valid);
            if (in_regs) {
#pragma unroll
                for (int d = 0; d < CAP; d++)
                    if (d < p.n_deltas) rv[d] = load_raw<W32>(vec, p.in[d], e, valid);
            }
            if (p.base != nullptr) {
                unpack_raw<W32>(p.in_dt, rb, b);
#pragma unroll
                for (int j = 0; j < 8; j += 2) poison = __builtin_elementwise_fma(f32x2{b[j], b[j + 1]}, f32x2{0.0f, 0.0f}, poison);
            }
            // filtered delta f_m = delta_m * (|delta_m| >= thr_m)   (ties.py:128-132); a select gives the same values
            // up to the sign of a zero, which neither the vote (sign(+-0) = 0) nor the sums (from +0) can see
            auto filter = [&](int m, float t_m, float amb_hi_m, float (&f)[8]) {
#pragma unroll
                for (int j = 0; j < 8; j += 2) poison = __builtin_elementwise_fma(f32x2{f[j], f[j + 1]}, f32x2{0.0f, 0.0f}, poison);
                if (p.base != nullptr) {
#pragma unroll
                    for (int j = 0; j < 8; j++) f[j] = f[j] - b[j];
                }
                if constexpr (DARE) apply_keep_bits(p, m, local, seg.numel, f);
                if (amb_hi_m >= 0.0f) {            // wave-uniform and rare: this pair's threshold is still provisional
#pragma unroll
                    for (int j = 0; j < 8; j++) hit |= (__builtin_fabsf(f[j]) >= t_m) && (__builtin_fabsf(f[j]) <= amb_hi_m);
                }
#pragma unroll
                for (int j = 0; j < 8; j++) f[j] = (__builtin_fabsf(f[j]) >= t_m) ? f[j] : 0.0f;
            };
            float vote[8], acc[8];                 // vote: an fp32 sum (vote_sgn) or an integer count kept in the same registers
            int cnt[8];
#pragma unroll
            for (int j = 0; j < 8; j++) { vote[j] = 0.0f; acc[j] = 0.0f; cnt[j] = 0; }
            // up to 4 deltas: the filtered deltas stay in registers between the sweeps; beyond, sweep 2 converts and filters
            // the raw words again (4 VGPRs per delta instead of 8 keeps the 8-delta kernel at a usable occupancy)
            constexpr bool KEEP_F = CAP <= 4;
            float f[KEEP_F ? CAP : 1][8];
            // sweep 1: the vote  (ties.py:120: sum of signs, or of deltas if vote_sgn), index order from +0
            for (int g = 0; g < p.n_deltas; g += CAP) {
                if (!in_regs) {
#pragma unroll
                    for (int d = 0; d < CAP; d++)
                        if (g + d < p.n_deltas) rv[d] = load_raw<W32>(vec, p.in[g + d], e, valid);
                }
#pragma unroll
                for (int d = 0; d < CAP; d++)
                    if (g + d < p.n_deltas) {
                        float (&fd)[8] = f[KEEP_F ? d : 0];
                        unpack_raw<W32>(p.in_dt, rv[d], fd);
                        if (in_regs) filter(g + d, thr[d], amb_hi[d], fd);
                        else filter(g + d, prov_threshold(st[g + d]), (st[g + d].mode == MODE_AMBIG) ? __uint_as_float(st[g + d].amb_hi) : -1.0f, fd);
                        if (x.vote_sgn) {
#pragma unroll
                            for (int j = 0; j < 8; j++) vote[j] += fd[j];
                        } else {
                            // sign(f) as an integer: clamp the bit pattern of f + 0 (the add turns -0 into +0) to [-1, 1]
#pragma unroll
                            for (int j = 0; j < 8; j++) {
                                const int sb = (int)__float_as_uint(fd[j] + 0.0f);
                                vote[j] = __int_as_float(__float_as_int(vote[j]) + __builtin_elementwise_min(__builtin_elementwise_max(sb, -1), 1));   // v_med3_i32
                            }
                        }
                    }
            }
            float fs[8];
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int vi = __float_as_int(vote[j]);
                fs[j] = x.vote_sgn ? sign_f(vote[j]) : (float)__builtin_elementwise_min(__builtin_elementwise_max(vi, -1), 1);
            }
            // sweep 2: disjoint mean over the deltas whose sign agrees (ties.py:122-124, :110).  g = f * fs is positive
            // exactly for the agreeing non-zero deltas; summing |f| = g and restoring the sign afterwards rounds like
            // the signed sum.  fs = 0 (agreeing deltas are the zeros) gives 0 / 0 -> nan_to_num -> 0 like the reference's
            // 0 / count.
            for (int g = 0; g < p.n_deltas; g += CAP) {
                if (!in_regs) {
#pragma unroll
                    for (int d = 0; d < CAP; d++)
                        if (g + d < p.n_deltas) rv[d] = load_raw<W32>(vec, p.in[g + d], e, valid);
                }
#pragma unroll
                for (int d = 0; d < CAP; d++)
                    if (g + d < p.n_deltas) {
                        float (&fd)[8] = f[KEEP_F ? d : 0];
                        if (!KEEP_F || !in_regs) {         // the filtered delta again, from the raw words
                            unpack_raw<W32>(p.in_dt, rv[d], fd);
                            filter(g + d, in_regs ? thr[d] : prov_threshold(st[g + d]), -1.0f, fd);
                        }
#pragma unroll
                        for (int j = 0; j < 8; j++) {
                            const float gj = __builtin_fmaxf(fd[j] * fs[j], 0.0f);
                            acc[j] += gj;
                            // gj is +-0 or positive: 1 iff gj > 0, from the bit pattern (v_med3_i32; -0 clamps to 0)
                            cnt[j] += __builtin_elementwise_min(__builtin_elementwise_max(__float_as_int(gj), 0), 1);
                        }
                    }
            }
            const bool bad_in = (poison.x != poison.x) | (poison.y != poison.y);
            float r[8];
            // filtered_delta / param_counts (ties.py:124), correctly rounded.  The count is an integer in 0 .. n_models,
            // so with y = RN(1 / count) from the table:  q0 = RN(a y), r = a - count q0 (exact in one fma),
            // q = RN(q0 + r y) is the correctly rounded quotient (Markstein; 9.6e9 random cases per divisor 1..32 checked
            // against IEEE division on the CPU) as long as nothing is denormal -- which `fast_div` guarantees; 4 instructions
            // instead of the ~10 of the general division.  count = 0: 0 * inf = NaN, like the reference's 0 / 0.
            auto finish = [&](auto fast) {
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const float a = acc[j] * fs[j];
                    float q;
                    if constexpr (decltype(fast)::value) {
                        const float y = recip[cnt[j]], q0 = a * y;
                        q = __builtin_fmaf(__builtin_fmaf(-(float)cnt[j], q0, a), y, q0);
                    } else {
                        q = a / (float)cnt[j];
                    }
                    q = __builtin_amdgcn_class(q, 0x1f8) ? q : 0.0f;                // nan_to_num(nan=0, posinf=0, neginf=0): keep finite classes
                    if constexpr (DARE) {
                        if (x.has_rescale) q = q / x.rescale_div;                   // ties.py:50
                    }
                    if (p.base != nullptr) q = __builtin_fmaf(q, x.alpha, b[j]);    // base.addcmul(ties, alpha)  (linear.py:76)
                    r[j] = q;
                }
            };
            if (fast_div) finish(std::true_type{}); else finish(std::false_type{});
            f32x2 out_poison = {0.0f, 0.0f};       // lanes past the end of a key hold zeros: they cannot raise it
#pragma unroll
            for (int j = 0; j < 8; j += 2) out_poison = __builtin_elementwise_fma(f32x2{r[j], r[j + 1]}, f32x2{0.0f, 0.0f}, out_poison);
            const bool bad_out = (out_poison.x != out_poison.x) | (out_poison.y != out_poison.y);
            store_rt(p.out_dt, vec, out, e, valid, r);
            if ((bad_in | bad_out) && flags != nullptr)
                atomicOr(flags + (p.tile_start == nullptr ? 0 : cur.s),
                         (bad_in ? MECHA_FLAG_INPUT_NONFINITE : 0u) | (bad_out ? MECHA_FLAG_OUTPUT_NONFINITE : 0u));
            if (hit) {
                const uint32_t mine = atomicAdd(&hq_n, 1u);
                if (mine < (uint32_t)HQ) { hq_elem[mine] = e; hq_seg[mine] = (uint32_t)cur.s; }
                else {                                                   // the block's buffer is full: straight to the queue
                    const unsigned long long slot = atomicAdd(&p.globals->n_hit, 1ull);
                    if (slot < p.amb_capacity) { p.hit_elem[slot] = e; p.hit_seg[slot] = (uint32_t)cur.s; }
                }
            }
        }
    }
    __syncthreads();
    const uint32_t buffered = hq_n < (uint32_t)HQ ? hq_n : (uint32_t)HQ;
    if (buffered == 0u) return;                                          // block-uniform
    if (threadIdx.x == 0) hq_base = atomicAdd(&p.globals->n_hit, (unsigned long long)buffered);
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < buffered; i += MECHA_BLOCK) {
        const unsigned long long slot = hq_base + i;
        if (slot < p.amb_capacity) { p.hit_elem[slot] = hq_elem[i]; p.hit_seg[slot] = hq_seg[i]; }
    }
}

// ------------------------------------------------------------------------------------------------ merge, full tiles
// The common case as straight-line code: add_difference(base, ties_sum(models - base)) with 1..8 models in 16-bit storage,
// sign election by count, every threshold of the key >= 2^-90 (no kept delta is a zero, the fma division is exact) and a
// tile that lies whole inside its key.  Everything else (ragged last tiles, tiny keys, zero thresholds, DARE, vote_sgn, fp32
// storage, 9+ models) stays with ties_merge_kernel above, which skips the tiles merged here (MergeExtra::skip_full_tiles).
// Same values as that kernel, bit for bit, with about half the vector instructions per element (it was VALU-bound: 79
// instructions per element against 5.2 TB/s worth of loads and stores):
//   * s_m = sign of the filtered delta as an integer (v_med3_i32 of the bit pattern; no -0 to normalise when thr > 0);
//     vote = s_1 + .. + s_N in one v_add3; elected sign = med3(vote, -1, 1);
//   * the agreeing deltas are selected by s_m == elected and summed signed, in index order from +0: the same roundings as the
//     sum of magnitudes times the sign (addition is symmetric under negation);
//   * the agree count needs no pass of its own: agreeing - disagreeing = |vote| and agreeing + disagreeing = kept, so
//     count = (vote * elected + kept) / 2, kept counted by the filter's own compare (elected = 0: every selected value is a
//     zero, the quotient is +0 or 0 * inf = NaN -> 0, like the reference's 0 / count);
//   * subtraction, sums and the finishing fmas on two values per instruction (v_pk_add_f32 / v_pk_fma_f32).
// E elements of 16-bit storage per lane: 8 (one 16-byte access) or 4 (one 8-byte access)
template <int E> struct RawWords { uint32_t w[E / 2]; };
template <int E>
__device__ __forceinline__ RawWords<E> load_words_nt(const uint16_t *ptr) {
    RawWords<E> r;
    if constexpr (E == 8) {
        const u32x4 t = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(ptr));
        r.w[0] = t.x; r.w[1] = t.y; r.w[2] = t.z; r.w[3] = t.w;
    } else {
        typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
        const u32x2 t = __builtin_nontemporal_load(reinterpret_cast<const u32x2 *>(ptr));
        r.w[0] = t.x; r.w[1] = t.y;
    }
    return r;
}
template <int DT, int E>
__device__ __forceinline__ void unpack_pairs(const RawWords<E> &r, f32x2 (&v)[E / 2]) {
#pragma unroll
    for (int q = 0; q < E / 2; q++) {
        if constexpr (DT == MECHA_BF16) v[q] = f32x2{__uint_as_float(r.w[q] << 16), __uint_as_float(r.w[q] & 0xffff0000u)};
        else v[q] = f32x2{f16_bits_to_f32(r.w[q] & 0xffffu), f16_bits_to_f32(r.w[q] >> 16)};
    }
}
template <int ODT, int E>
__device__ __forceinline__ void store_words_nt(void *out, int64_t e, const float (&r)[E]) {
    if constexpr (ODT == MECHA_F32) {
#pragma unroll
        for (int h = 0; h < E / 4; h++) {
            const f32x4 v = {r[4 * h], r[4 * h + 1], r[4 * h + 2], r[4 * h + 3]};
            __builtin_nontemporal_store(v, reinterpret_cast<f32x4 *>(static_cast<float *>(out) + e) + h);
        }
    } else {
        uint32_t w[E / 2];
#pragma unroll
        for (int q = 0; q < E / 2; q++) {
            if constexpr (ODT == MECHA_BF16) w[q] = f32_to_bf16_bits(r[2 * q]) | (f32_to_bf16_bits(r[2 * q + 1]) << 16);
            else w[q] = f32_to_f16_bits(r[2 * q]) | (f32_to_f16_bits(r[2 * q + 1]) << 16);
        }
        if constexpr (E == 8) {
            const u32x4 v = {w[0], w[1], w[2], w[3]};
            __builtin_nontemporal_store(v, reinterpret_cast<u32x4 *>(static_cast<uint16_t *>(out) + e));
        } else {
            typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
            const u32x2 v = {w[0], w[1]};
            __builtin_nontemporal_store(v, reinterpret_cast<u32x2 *>(static_cast<uint16_t *>(out) + e));
        }
    }
}
// clamp to [-1, 1]: the sign of an integer, or of a float's bit pattern (no -0 here).  As inline assembly: written as min(max())
// the compiler recognises a sign function and emits two compares and two selects instead of the one v_med3_i32.
__device__ __forceinline__ int sign_i32(int v) {
    int r;
    asm("v_med3_i32 %0, %1, -1, 1" : "=v"(r) : "v"(v));
    return r;
}

// E: elements per lane.  8 (16-byte accesses) up to 4 deltas; 4 (8-byte accesses) for 5..8 deltas: with 8 elements the state of 8
// input streams costs 180-228 VGPRs (2 waves per SIMD, 0.61 of 8 TB/s on the FLUX x8 stripe) whatever the source order -- the
// scheduler hoists every model's unpack and subtraction to the front; half the elements is half the registers.
// (5 waves per SIMD by launch bounds -- 96 VGPRs, 2 spilled with 3 models, 11 with 7 -- measured slower: 5.9 against 5.5 ms
//  on the SDXL x4 job, 6.7 against 4.7 on the FLUX stripe)
// DARE: the same with the keep masks of ties_sum_with_dropout (delta * mask before the filter, ties.py:33; the result divided by
// 1 - probability, ties.py:50).  A masked delta is mostly zeros, so thresholds are usually 0 and kept values may be anything:
// -0 is normalised before the sign is taken, and a sum in the denormal range sends its vector through the IEEE divisions.
template <int N, int DT, int ODT, int E, bool DARE>
__global__ __launch_bounds__(MECHA_BLOCK) void ties_merge_full_kernel(TiesParams p, MergeExtra x, void *out, uint32_t *flags) {
    constexpr int P = E / 2;                       // pairs per lane
    constexpr int BLOCKS_PER_TILE = TIES_TILE / (MECHA_BLOCK * E);
    // FLAT grid: one block per 256 lanes x E elements (half a tile at E = 8), no loop -- a wave loads, stores and ends.
    // Loads and stores share gfx9's one in-order vmcnt counter, so a looping wave waits for its previous stores whenever it
    // waits for its next loads; with a write stream every looped shape stops at 5.1-5.2 TB/s where this one reaches 6.0
    // (tools/stream_bench3.hip, profiles/r02_stream_bench3.txt).  The block finds its tile through one 16-byte record.
    __shared__ float recip[16];                    // correctly rounded 1 / count, [0] = +inf
    if (threadIdx.x < 16) recip[threadIdx.x] = 1.0f / (float)threadIdx.x;
    const long long t = (long long)(blockIdx.x / BLOCKS_PER_TILE);
    const int u = (int)(blockIdx.x % BLOCKS_PER_TILE);
    TileRec rec;
    // (read through the constant address space: block-uniform, so scalar loads -- the compiler otherwise selects between this
    //  pointer and the kernel argument's and reads seg_offset with a FLAT vector load, whose wait drains the data loads)
    if (p.tile_rec != nullptr) {
        typedef const __attribute__((address_space(4))) long long *const_i64;
        typedef const __attribute__((address_space(4))) int *const_i32;
        const TileRec *g = p.tile_rec + t;
        rec.elem = *(const_i64)&g->elem; rec.seg = *(const_i32)&g->seg; rec.valid = *(const_i32)&g->valid;
        rec.seg_offset = *(const_i64)&g->seg_offset; rec.seg_numel = *(const_i64)&g->seg_numel;
    }
    else {
        const long long left = p.seg0.numel - t * TIES_TILE;
        rec = TileRec{t * TIES_TILE, 0, (int)(left < TIES_TILE ? left : TIES_TILE), p.seg0.offset, p.seg0.numel};
    }
    if (rec.valid < TIES_TILE) return;                                                 // ties_merge_kernel's tile
    const int64_t e = rec.elem + (int64_t)u * (MECHA_BLOCK * E) + (int64_t)threadIdx.x * E;
    const uint16_t *base = static_cast<const uint16_t *>(p.base);
    // every load is issued before the key's thresholds (a second, dependent read) are needed
    RawWords<E> rv[N];
    const RawWords<E> rb = load_words_nt<E>(base + e);
#pragma unroll
    for (int m = 0; m < N; m++) rv[m] = load_words_nt<E>(static_cast<const uint16_t *>(p.in[m]) + e);
    // DARE: this lane's E keep bits per model.  The mask words are loaded here, behind the data and without a branch: loads under
    // a condition each came with a full wait of their own, three round trips after the data had landed (7.2 ms for the SDXL job)
    uint32_t mbits[N], mw0[DARE ? N : 1], mw1[DARE ? N : 1], msh[DARE ? N : 1];
    if constexpr (DARE) {
        const int64_t local = e - rec.seg_offset;
#pragma unroll
        for (int m = 0; m < N; m++) {
            const int64_t sbit = (p.mask_stream ? (int64_t)m * rec.seg_numel : (int64_t)m * p.mask_words * 32) + local;
            msh[m] = (uint32_t)(sbit & 31);
            mw0[m] = p.keep_bits[sbit >> 5];
            mw1[m] = p.keep_bits[(sbit >> 5) + (msh[m] > 32u - E ? 1 : 0)];             // see apply_keep_bits
        }
    }
    float thr[N], amb_hi[N];
    bool key_ok = true, any_amb = false;
    {
        const SelectState *st = p.state + (int64_t)rec.seg * N;
#pragma unroll
        for (int d = 0; d < N; d++) {
            thr[d] = prov_threshold(st[d]);
            amb_hi[d] = (st[d].mode == MODE_AMBIG) ? __uint_as_float(st[d].amb_hi) : -1.0f;
            key_ok = key_ok && (thr[d] >= 0x1p-90f);
            any_amb = any_amb || (amb_hi[d] >= 0.0f);
        }
    }
    if (!DARE && !key_ok) return;                                                       // ties_merge_kernel's key (same test there)
    // DARE: thresholds may be 0, and zeros (dropped, or genuine +-0) pass `|d| >= 0` but neither vote nor count where a sign is
    // elected.  The keep test runs against max(thr, smallest denormal): |d| >= 2^-149 is d != 0, and above a positive
    // threshold nothing changes -- one compare per element instead of two.
    float thr_k[N];
#pragma unroll
    for (int d = 0; d < N; d++) thr_k[d] = DARE ? __builtin_fmaxf(thr[d], __uint_as_float(1u)) : thr[d];
    if constexpr (DARE) {
#pragma unroll
        for (int m = 0; m < N; m++) mbits[m] = __builtin_amdgcn_alignbit(mw1[m], mw0[m], msh[m]);
    }
    __syncthreads();                                                                    // recip[]
    const f32x2 zero2 = {0.0f, 0.0f};
    const f32x2 alpha2 = {x.alpha, x.alpha};
    {
        {
            f32x2 b2[P];
            unpack_pairs<DT, E>(rb, b2);
            f32x2 poison = zero2;                  // fma(x, 0, poison): 0 while every input is finite, NaN after inf / NaN
#pragma unroll
            for (int q = 0; q < P; q++) poison = __builtin_elementwise_fma(b2[q], zero2, poison);
            int kept[E];
#pragma unroll
            for (int j = 0; j < E; j++) kept[j] = 0;
            bool hit = false;
            f32x2 acc[P];
            int cnt[E];
            if constexpr (N <= 4 || E == 4) {
                // one sweep: the filtered deltas and their signs stay in registers
                float f[N][E];
                int sgn[N][E];
#pragma unroll
                for (int m = 0; m < N; m++) {
                    f32x2 v2[P];
                    unpack_pairs<DT, E>(rv[m], v2);
#pragma unroll
                    for (int q = 0; q < P; q++) poison = __builtin_elementwise_fma(v2[q], zero2, poison);
#pragma unroll
                    for (int q = 0; q < P; q++) v2[q] = v2[q] - b2[q];                     // subtract (linear.py:101)
                    if (any_amb && amb_hi[m] >= 0.0f) {                                    // wave-uniform and rare
                        // (DARE: a dropped delta is a zero, never an ambiguous value -- an ambiguous bin starts above its cell's first pattern)
#pragma unroll
                        for (int j = 0; j < E; j++)
                            hit |= (!DARE || ((mbits[m] >> j) & 1u)) && (__builtin_fabsf(v2[j >> 1][j & 1]) >= thr[m]) && (__builtin_fabsf(v2[j >> 1][j & 1]) <= amb_hi[m]);
                    }
#pragma unroll
                    for (int j = 0; j < E; j++) {
                        float d = v2[j >> 1][j & 1];
                        // DARE: delta * mask (ties.py:33); a dropped delta is +0 and fails the keep test below like any zero (thr_k), so
                        // zeros stay out of `kept`, and -0 never reaches sign_i32
                        if constexpr (DARE) d = ((mbits[m] >> j) & 1u) ? d : 0.0f;
                        const bool keep = __builtin_fabsf(d) >= thr_k[m];                  // filter_top_k (ties.py:128-132)
                        kept[j] += keep ? 1 : 0;
                        f[m][j] = keep ? d : 0.0f;
                        sgn[m][j] = sign_i32(__float_as_int(f[m][j]));
                    }
                }
#pragma unroll
                for (int q = 0; q < P; q++) {
#pragma unroll
                    for (int h = 0; h < 2; h++) {
                        const int j = 2 * q + h;
                        int vote = sgn[0][j];
#pragma unroll
                        for (int m = 1; m < N; m++) vote += sgn[m][j];
                        const int fs = sign_i32(vote);
                        cnt[j] = (vote * fs + kept[j]) >> 1;
#pragma unroll
                        for (int m = 0; m < N; m++) f[m][j] = (sgn[m][j] == fs) ? f[m][j] : 0.0f;
                    }
                    acc[q] = zero2;
#pragma unroll
                    for (int m = 0; m < N; m++) acc[q] = acc[q] + f32x2{f[m][2 * q], f[m][2 * q + 1]};   // index order, from +0 (ties.py:123)
                }
            } else {
                // 5..8 models: two sweeps over the raw words (4 registers per model instead of 16 for delta and sign)
                int vote[E], fs[E];
#pragma unroll
                for (int j = 0; j < E; j++) vote[j] = 0;
#pragma unroll
                for (int m = 0; m < N; m++) {
                    f32x2 v2[P];
                    unpack_pairs<DT, E>(rv[m], v2);
#pragma unroll
                    for (int q = 0; q < P; q++) poison = __builtin_elementwise_fma(v2[q], zero2, poison);
#pragma unroll
                    for (int q = 0; q < P; q++) v2[q] = v2[q] - b2[q];
                    if (any_amb && amb_hi[m] >= 0.0f) {
#pragma unroll
                        for (int j = 0; j < E; j++) hit |= (__builtin_fabsf(v2[j >> 1][j & 1]) >= thr[m]) && (__builtin_fabsf(v2[j >> 1][j & 1]) <= amb_hi[m]);
                    }
#pragma unroll
                    for (int j = 0; j < E; j++) {
                        float d = v2[j >> 1][j & 1];
                        if constexpr (DARE) d = ((mbits[m] >> j) & 1u) ? d : 0.0f;
                        const bool keep = __builtin_fabsf(d) >= thr_k[m];
                        kept[j] += keep ? 1 : 0;
                        vote[j] += sign_i32(__float_as_int(keep ? d : 0.0f));
                    }
                }
#pragma unroll
                for (int j = 0; j < E; j++) {
                    fs[j] = sign_i32(vote[j]);
                    cnt[j] = (vote[j] * fs[j] + kept[j]) >> 1;
                }
#pragma unroll
                for (int q = 0; q < P; q++) acc[q] = zero2;
#pragma unroll
                for (int m = 0; m < N; m++) {
                    f32x2 v2[P];
                    unpack_pairs<DT, E>(rv[m], v2);
#pragma unroll
                    for (int q = 0; q < P; q++) v2[q] = v2[q] - b2[q];
#pragma unroll
                    for (int q = 0; q < P; q++) {
                        f32x2 g;
#pragma unroll
                        for (int h = 0; h < 2; h++) {
                            float d = v2[q][h];
                            if constexpr (DARE) d = ((mbits[m] >> (2 * q + h)) & 1u) ? d : 0.0f;
                            // kept and of the elected sign (elected 0: only a zero delta could match, and that one is not kept)
                            const bool agree = (__builtin_fabsf(d) >= thr_k[m]) & (sign_i32(__float_as_int(d)) == fs[2 * q + h]);
                            g[h] = agree ? d : 0.0f;
                        }
                        acc[q] = acc[q] + g;
                    }
                }
            }
            // DARE: the three-instruction divisions below (by the count, then by 1 - probability) are exact for sums that are 0 or
            // lie in [2^-57, 2^59) (the quotient by a count <= 8 then lies in [2^-60, 2^60]); one test of the binary exponents of the
            // lane's E sums (frexp: 0 for a zero, inf and NaN, which come out as 0 either way) instead of two range tests per
            // element, and the IEEE divisions behind ONE rarely taken branch.
            bool slow = false;
            if constexpr (DARE) {
                int emin = 0, emax = 0;
#pragma unroll
                for (int j = 0; j < E; j++) {
                    const int ex = __builtin_amdgcn_frexp_expf(acc[j >> 1][j & 1]);
                    emin = ex < emin ? ex : emin;
                    emax = ex > emax ? ex : emax;
                }
                slow = emin < -56 || emax > 59 || (x.has_rescale && !x.fast_rescale);
            }
            f32x2 r2[P];
#pragma unroll
            for (int q = 0; q < P; q++) {
                // filtered_delta / param_counts (ties.py:124), correctly rounded: y = RN(1 / count), q0 = RN(a y),
                // q = RN(q0 + (a - count q0) y)  (see ties_merge_kernel)
                const f32x2 y = {recip[cnt[2 * q]], recip[cnt[2 * q + 1]]};
                const f32x2 negc = {-(float)cnt[2 * q], -(float)cnt[2 * q + 1]};
                const f32x2 q0 = acc[q] * y;
                f32x2 qq = __builtin_elementwise_fma(__builtin_elementwise_fma(negc, q0, acc[q]), y, q0);
                if constexpr (DARE) {
                    // with a zero threshold a sum may lie in the denormal range, where the fma form is no longer exact: IEEE division
                    // then; the result is divided by 1 - probability (ties.py:46-50) the same way
                    if (slow) {
#pragma unroll
                        for (int h = 0; h < 2; h++) {
                            const uint32_t ab = __float_as_uint(acc[q][h]) & 0x7fffffffu;
                            if (ab - 1u < 0x12800000u - 1u) qq[h] = acc[q][h] / (float)cnt[2 * q + h];      // 0 < |a| < 2^-90
                        }
                    }
                }
                qq.x = __builtin_amdgcn_class(qq.x, 0x1f8) ? qq.x : 0.0f;              // nan_to_num(nan=0, posinf=0, neginf=0)
                qq.y = __builtin_amdgcn_class(qq.y, 0x1f8) ? qq.y : 0.0f;
                if constexpr (DARE) {
                    if (x.has_rescale) {
                        const f32x2 rc = {x.rescale_rcp, x.rescale_rcp}, nd = {-x.rescale_div, -x.rescale_div};
                        const f32x2 s0 = qq * rc;
                        f32x2 sq = __builtin_elementwise_fma(__builtin_elementwise_fma(nd, s0, qq), rc, s0);
                        if (slow) {
#pragma unroll
                            for (int h = 0; h < 2; h++) {
                                const uint32_t ab = __float_as_uint(qq[h]) & 0x7fffffffu;
                                if (!x.fast_rescale || (ab - 1u < 0x21800000u - 1u) || ab > 0x5d800000u) sq[h] = qq[h] / x.rescale_div;   // outside [2^-60, 2^60]
                            }
                        }
                        qq = sq;
                    }
                }
                r2[q] = __builtin_elementwise_fma(qq, alpha2, b2[q]);                  // base.addcmul(ties, alpha)  (linear.py:76)
            }
            f32x2 out_poison = zero2;
#pragma unroll
            for (int q = 0; q < P; q++) out_poison = __builtin_elementwise_fma(r2[q], zero2, out_poison);
            float r[E];
#pragma unroll
            for (int q = 0; q < P; q++) { r[2 * q] = r2[q].x; r[2 * q + 1] = r2[q].y; }
            store_words_nt<ODT, E>(out, e, r);
            const bool bad = (poison.x != poison.x) | (poison.y != poison.y);
            const bool bad_o = (out_poison.x != out_poison.x) | (out_poison.y != out_poison.y);
            if constexpr (E == 4) {
                // the hit queue holds 8-element vectors (ties_amb_collect_kernel scans 8 from the recorded element): a lane pair
                // shares one, recorded once by its even lane (whose element index is a multiple of 8)
                if (any_amb) {
                    const bool pair_hit = hit | (__shfl_xor((int)hit, 1, 64) != 0);
                    hit = pair_hit && ((threadIdx.x & 1) == 0);
                }
            }
            if (hit) {
                const unsigned long long slot = atomicAdd(&p.globals->n_hit, 1ull);
                if (slot < p.amb_capacity) { p.hit_elem[slot] = e; p.hit_seg[slot] = (uint32_t)rec.seg; }
            }
            if ((bad | bad_o) && flags != nullptr)
                atomicOr(flags + (p.tile_start == nullptr ? 0 : rec.seg), (bad ? MECHA_FLAG_INPUT_NONFINITE : 0u) | (bad_o ? MECHA_FLAG_OUTPUT_NONFINITE : 0u));
        }
    }
}

// ------------------------------------------------------------------------------------------------ A: select + fixup
// One block per ambiguous pair: exact radix select (11 / 11 / 10 bits) of rank amb_rank inside its list.
__global__ __launch_bounds__(MECHA_BLOCK) void ties_amb_select_kernel(TiesParams p, float *thresholds_out, int64_t n_sd) {
    __shared__ uint32_t hist[TIES_BINS];
    __shared__ uint32_t s_prefix, s_rank;
    __shared__ unsigned long long wave_tot[MECHA_BLOCK / 64];
    __shared__ int sel;
    __shared__ unsigned long long sel_below;
    __shared__ uint32_t sel_count;
    // a few hundred blocks walk the pairs (ambiguous ones are rare: a block per pair spent 0.14 ms on 7 587 early exits)
    for (int64_t sd = blockIdx.x; sd < n_sd; sd += gridDim.x) {
    const SelectState st = p.state[sd];
    if (st.mode != MODE_AMBIG) continue;                                  // block-uniform
    const uint32_t n = p.fill[sd] < st.amb_count ? p.fill[sd] : st.amb_count;
    const uint32_t *bits = p.amb_bits + st.amb_offset;
    if (threadIdx.x == 0) { s_prefix = 0; s_rank = st.amb_rank; }
    const int shifts[3] = {21, 10, 0};
    const uint32_t masks[3] = {0u, 0xffe00000u, 0xfffffc00u};
    const uint32_t nbins[3] = {2048u, 2048u, 1024u};
    for (int pass = 0; pass < 3; pass++) {
        for (int i = threadIdx.x; i < TIES_BINS; i += MECHA_BLOCK) hist[i] = 0;
        __syncthreads();
        const uint32_t prefix = s_prefix;
        for (uint32_t i = threadIdx.x; i < n; i += MECHA_BLOCK) {
            const uint32_t b = bits[i];
            if ((b & masks[pass]) == prefix) atomicAdd(&hist[(b >> shifts[pass]) & (nbins[pass] - 1u)], 1u);
        }
        __syncthreads();
        // (the scan over the bins by the whole block: one thread walking 2048 bins three times took 0.1 ms per pair)
        int bin; unsigned long long below; uint32_t count;
        const uint32_t rank = s_rank;
        find_bin(hist, (unsigned long long)rank, bin, below, count, wave_tot, &sel, &sel_below, &sel_count);
        if (bin < 0) { bin = (int)nbins[pass] - 1; below = 0; }               // rank beyond the list (cannot happen): the old scan's last bin
        __syncthreads();
        if (threadIdx.x == 0) {
            s_prefix = prefix | ((uint32_t)bin << shifts[pass]);
            s_rank = rank - (uint32_t)below;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        p.state[sd].threshold_bits = s_prefix;
        if (thresholds_out != nullptr) thresholds_out[sd] = __uint_as_float(s_prefix);
    }
    __syncthreads();                                                      // s_prefix / s_rank are rewritten for the next pair
    }
}

// Listed elements whose |delta| is below the final threshold were provisionally kept: recompute their output element.
__global__ __launch_bounds__(MECHA_BLOCK) void ties_fixup_kernel(TiesParams p, MergeExtra x, void *out, uint32_t *flags) {
    const unsigned long long n = p.globals->n_amb < p.amb_capacity ? p.globals->n_amb : p.amb_capacity;
    for (unsigned long long i = (unsigned long long)blockIdx.x * MECHA_BLOCK + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * MECHA_BLOCK) {
        const int64_t sd = p.amb_sd[i];
        if (p.amb_bits[i] >= p.state[sd].threshold_bits) continue;            // the provisional decision was right
        const int s = (int)(sd / p.n_deltas);
        const int64_t e = p.amb_elem[i];
        const SelectState *st = p.state + (int64_t)s * p.n_deltas;
        const float b = (p.base != nullptr) ? load1_rt(p.in_dt, p.base, e) : 0.0f;
        const mecha_segment seg = get_segment(p, s);
        auto filtered = [&](int m) {
            float f = load1_rt(p.in_dt, p.in[m], e);
            if (p.base != nullptr) f = f - b;
            if (p.keep_bits != nullptr && !keep_bit(p, m, e - seg.offset, seg.numel)) f = 0.0f;               // DARE: delta * mask (ties.py:33)
            return f * ((__builtin_fabsf(f) >= __uint_as_float(st[m].threshold_bits)) ? 1.0f : 0.0f);
        };
        float vote = 0.0f;
        for (int m = 0; m < p.n_deltas; m++) { const float f = filtered(m); vote += x.vote_sgn ? f : sign_f(f); }
        const float fs = sign_f(vote);
        float acc = 0.0f, cnt = 0.0f;
        for (int m = 0; m < p.n_deltas; m++) {
            const float f = filtered(m);
            const float agree = (sign_f(f) == fs) ? 1.0f : 0.0f;
            acc += f * agree; cnt += agree;
        }
        float q = acc / cnt;
        if (nonfinite_f32(q)) q = 0.0f;
        if (x.has_rescale) q = q / x.rescale_div;                            // DARE: / (1 - probability) (ties.py:46-50)
        if (p.base != nullptr) q = __builtin_fmaf(q, x.alpha, b);
        store1_rt(p.out_dt, out, e, q);
        if (nonfinite_f32(q) && flags != nullptr) atomicOr(flags + (p.tile_start == nullptr ? 0 : s), MECHA_FLAG_OUTPUT_NONFINITE);
    }
}

// tile_start[s] = number of tiles before segment s (one block; n_segs is a few thousand)
__global__ __launch_bounds__(1024) void ties_tile_index_kernel(const mecha_segment *segs, int n_segs, long long *tile_start) {
    __shared__ long long part[1024];
    const int per = (n_segs + 1023) / 1024;
    const int lo = (threadIdx.x * per < n_segs) ? threadIdx.x * per : n_segs;
    const int hi = (lo + per < n_segs) ? lo + per : n_segs;
    long long sum = 0;
    for (int i = lo; i < hi; i++) sum += (segs[i].numel + TIES_TILE - 1) / TIES_TILE;
    part[threadIdx.x] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        long long run = 0;
        for (int i = 0; i < 1024; i++) { long long v = part[i]; part[i] = run; run += v; }
        tile_start[n_segs] = run;
    }
    __syncthreads();
    long long run = part[threadIdx.x];
    for (int i = lo; i < hi; i++) { tile_start[i] = run; run += (segs[i].numel + TIES_TILE - 1) / TIES_TILE; }
}

// One record per tile for the flat-grid kernels (one block per segment; a block of the flat merge kernel then needs ONE
// 32-byte read to know where it is, instead of a binary search over the tile prefix).
__global__ __launch_bounds__(MECHA_BLOCK) void ties_tile_records_kernel(const mecha_segment *segs, const long long *tile_start, TileRec *rec) {
    const int s = blockIdx.x;
    const mecha_segment seg = segs[s];
    const long long first = tile_start[s], n = tile_start[s + 1] - first;
    for (long long i = threadIdx.x; i < n; i += MECHA_BLOCK) {
        const long long left = seg.numel - i * TIES_TILE;
        rec[first + i] = TileRec{seg.offset + i * TIES_TILE, s, (int)(left < TIES_TILE ? left : TIES_TILE), seg.offset, seg.numel};
    }
}

// ------------------------------------------------------------------------------------------------ model-stock factor
// ties_sum_extended with apply_stock == False (ties.py:90 -- the flag is inverted upstream: False APPLIES the factor):
//   t = n c / (1 + (n - 1) c),  c = mean over consecutive pairs of cos(g_i, g_{i+1}) along the LAST dimension
//   (ties.py:146-159, torch.nn.CosineSimilarity(dim=-1, eps)), g_i = the filtered deltas that agree with the elected sign;
//   out = nan_to_num(sum_i g_i * t / count).
// Per-key, delta form, 2..8 deltas.  Three kernels after the thresholds are known: row sums (dot products and squared
// norms, fp64: torch's fp32 reduction order is not fixed), the factor per row, the apply pass.
struct StockParams {
    int64_t row_len, n_rows;    // per key form: the key's last dimension and its number of rows
    const int64_t *row_lens;    // batched: device [n_segs] last dimension of each key ...
    const int64_t *row_starts;  //          ... and the first row of each key in `sums` / `t` (prefix of numel / row_len)
    double *sums;               // [n_rows][2 n - 1]: n squared norms, then n - 1 dot products
    float *t;                   // [n_rows]
    float cos_eps;
};
constexpr int STOCK_MAX = 8;

// row_starts[s] = rows of the keys before s; row_starts[n_segs] = all rows (one block; n_segs is a few thousand)
__global__ __launch_bounds__(1024) void ties_stock_rows_kernel(const mecha_segment *segs, const int64_t *row_lens, int n_segs, int64_t *row_starts) {
    __shared__ long long part[1024];
    const int per = (n_segs + 1023) / 1024;
    const int lo = (threadIdx.x * per < n_segs) ? threadIdx.x * per : n_segs;
    const int hi = (lo + per < n_segs) ? lo + per : n_segs;
    long long sum = 0;
    for (int i = lo; i < hi; i++) sum += segs[i].numel / row_lens[i];
    part[threadIdx.x] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        long long run = 0;
        for (int i = 0; i < 1024; i++) { long long v = part[i]; part[i] = run; run += v; }
        row_starts[n_segs] = run;
    }
    __syncthreads();
    long long run = part[threadIdx.x];
    for (int i = lo; i < hi; i++) { row_starts[i] = run; run += segs[i].numel / row_lens[i]; }
}

// g[d][j] for the 8 elements at `local` of key s (delta = model - base where there is a base; filtered by threshold, DARE mask,
// and agreement with the elected sign), the agree count, and the base values
template <bool DARE>
__device__ __forceinline__ void stock_agreeing(const TiesParams &p, int vote_sgn, int s, const mecha_segment &seg, int64_t local, int valid, bool vec,
                                               float (&g)[STOCK_MAX][8], float (&cnt)[8], float (&b)[8], bool &bad_in) {
    const int64_t e = seg.offset + local;
    float vote[8];
#pragma unroll
    for (int j = 0; j < 8; j++) { vote[j] = 0.0f; cnt[j] = 0.0f; b[j] = 0.0f; }
    if (p.base != nullptr) {
        load_rt(p.in_dt, vec, p.base, e, valid, b);
#pragma unroll
        for (int j = 0; j < 8; j++) bad_in |= nonfinite_f32(b[j]);
    }
#pragma unroll
    for (int d = 0; d < STOCK_MAX; d++) {
        if (d < p.n_deltas) {
            load_rt(p.in_dt, vec, p.in[d], e, valid, g[d]);
#pragma unroll
            for (int j = 0; j < 8; j++) bad_in |= nonfinite_f32(g[d][j]);
            if (p.base != nullptr) {
#pragma unroll
                for (int j = 0; j < 8; j++) g[d][j] = g[d][j] - b[j];                             // subtract (linear.py:101)
            }
            if constexpr (DARE) apply_keep_bits(p, d, local, seg.numel, g[d]);
            if (p.raw) continue;                                                                // model_stock: no filter, no election
            const float thr = __uint_as_float(p.state[(int64_t)s * p.n_deltas + d].threshold_bits);
#pragma unroll
            for (int j = 0; j < 8; j++) {
                g[d][j] = g[d][j] * ((__builtin_fabsf(g[d][j]) >= thr) ? 1.0f : 0.0f);          // filter_top_k (ties.py:128-132)
                vote[j] += vote_sgn ? g[d][j] : sign_f(g[d][j]);
            }
        }
    }
    if (p.raw) {
#pragma unroll
        for (int j = 0; j < 8; j++) cnt[j] = (float)p.n_deltas;
        return;
    }
#pragma unroll
    for (int d = 0; d < STOCK_MAX; d++) {
        if (d < p.n_deltas) {
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const float agree = (sign_f(g[d][j]) == sign_f(vote[j])) ? 1.0f : 0.0f;          // ties.py:122
                g[d][j] = g[d][j] * agree;
                cnt[j] += agree;
            }
        }
    }
}

// Both stock kernels walk tiles like the radix passes (a block owns a contiguous run of tiles, the cursor tracks the key): one
// key with p.seg0, or every key of an arena.
template <bool DARE>
__global__ __launch_bounds__(MECHA_BLOCK) void ties_stock_reduce_kernel(TiesParams p, StockParams sp, int vote_sgn, uint32_t *flags) {
    const int n = p.n_deltas, width = 2 * n - 1;
    const long long chunk = (p.total_tiles + gridDim.x - 1) / gridDim.x;
    const long long t0 = (long long)blockIdx.x * chunk;
    const long long t1 = (t0 + chunk < p.total_tiles) ? t0 + chunk : p.total_tiles;
    TileCursor cur;
    for (long long t = t0; t < t1; t++) {
        cursor_seek(p, t, cur);
        const mecha_segment seg = cur.seg;
        const int64_t row_len = sp.row_lens != nullptr ? sp.row_lens[cur.s] : sp.row_len;
        const int64_t row_base = sp.row_starts != nullptr ? sp.row_starts[cur.s] : 0;
#pragma unroll 1
        for (int u = 0; u < 2; u++) {
            const int64_t local = (t - cur.first) * TIES_TILE + (int64_t)u * (TIES_TILE / 2) + (int64_t)threadIdx.x * 8;
            const int valid = (local >= seg.numel) ? 0 : ((seg.numel - local >= 8) ? 8 : (int)(seg.numel - local));
            bool bad_in = false;
            float g[STOCK_MAX][8], cnt[8], b[8];
            if (valid > 0) stock_agreeing<DARE>(p, vote_sgn, cur.s, seg, local, valid, p.vec_ok && valid == 8, g, cnt, b, bad_in);
            double acc[2 * STOCK_MAX - 1];
            int64_t row = local / row_len;
            int64_t row_end = (row + 1) * row_len;
            auto clear = [&]() {
#pragma unroll
                for (int i = 0; i < 2 * STOCK_MAX - 1; i++) acc[i] = 0.0;
            };
            auto flush = [&](int64_t r) {
#pragma unroll
                for (int i = 0; i < 2 * STOCK_MAX - 1; i++)
                    if (i < width && acc[i] != 0.0) atomicAdd(&sp.sums[(row_base + r) * width + i], acc[i]);
            };
            clear();
            // whole wave inside one row (rows of >= 512 elements, mostly): reduce across lanes first, one lane adds
            const int64_t first_row = __shfl(row, 0, 64);
            const bool lane_one_row = valid == 8 && (local + valid <= row_end);
            const bool uniform = __ballot(lane_one_row && row == first_row) == ~0ull;        // all 64 lanes active, full, in one row
#pragma unroll
            for (int j = 0; j < 8; j++) {
                if (j < valid) {
                    if (local + j >= row_end) { flush(row); clear(); row += 1; row_end += row_len; if (local + j >= row_end) { row = (local + j) / row_len; row_end = (row + 1) * row_len; } }
#pragma unroll
                    for (int d = 0; d < STOCK_MAX; d++) {
                        if (d < n) {
                            acc[d] += (double)g[d][j] * (double)g[d][j];
                            if (d + 1 < n) acc[n + d] += (double)g[d][j] * (double)g[d + 1][j];
                        }
                    }
                }
            }
            if (uniform) {
#pragma unroll
                for (int i = 0; i < 2 * STOCK_MAX - 1; i++) {
                    if (i < width) {
#pragma unroll
                        for (int off = 32; off > 0; off >>= 1) acc[i] += __shfl_xor(acc[i], off, 64);
                    }
                }
                if ((threadIdx.x & 63) == 0) flush(row);
            } else if (valid > 0) {
                flush(row);
            }
            if (bad_in && flags != nullptr) atomicOr(flags + (p.tile_start == nullptr ? 0 : cur.s), MECHA_FLAG_INPUT_NONFINITE);
        }
    }
}

__global__ __launch_bounds__(MECHA_BLOCK) void ties_stock_t_kernel(StockParams sp, int n, const int64_t *n_rows_dev) {
    const int64_t n_rows = n_rows_dev != nullptr ? *n_rows_dev : sp.n_rows;
    for (int64_t r = (int64_t)blockIdx.x * MECHA_BLOCK + threadIdx.x; r < n_rows; r += (int64_t)gridDim.x * MECHA_BLOCK) {
        const double *s = sp.sums + r * (2 * n - 1);
        double mean = 0.0;
        for (int i = 0; i + 1 < n; i++) {
            // cosine_similarity: x.y / (max(|x|, eps) max(|y|, eps))
            const double nx = fmax(sqrt(s[i]), (double)sp.cos_eps), ny = fmax(sqrt(s[i + 1]), (double)sp.cos_eps);
            mean += s[n + i] / (nx * ny);
        }
        const float c = (float)(mean / (double)(n - 1));
        sp.t[r] = ((float)n * c) / (1.0f + (float)(n - 1) * c);                                  // ties.py:158
    }
}

template <bool DARE>
__global__ __launch_bounds__(MECHA_BLOCK) void ties_stock_apply_kernel(TiesParams p, StockParams sp, MergeExtra x, void *out, uint32_t *flags) {
    const long long chunk = (p.total_tiles + gridDim.x - 1) / gridDim.x;
    const long long t0 = (long long)blockIdx.x * chunk;
    const long long t1 = (t0 + chunk < p.total_tiles) ? t0 + chunk : p.total_tiles;
    TileCursor cur;
    for (long long t = t0; t < t1; t++) {
        cursor_seek(p, t, cur);
        const mecha_segment seg = cur.seg;
        const int64_t row_len = sp.row_lens != nullptr ? sp.row_lens[cur.s] : sp.row_len;
        const int64_t row_base = sp.row_starts != nullptr ? sp.row_starts[cur.s] : 0;
#pragma unroll 1
        for (int u = 0; u < 2; u++) {
            const int64_t local = (t - cur.first) * TIES_TILE + (int64_t)u * (TIES_TILE / 2) + (int64_t)threadIdx.x * 8;
            if (local >= seg.numel) continue;
            const int valid = (seg.numel - local >= 8) ? 8 : (int)(seg.numel - local);
            const bool vec = p.vec_ok && valid == 8;
            bool bad_in = false, bad_out = false;
            float g[STOCK_MAX][8], cnt[8], b[8], r[8];
            stock_agreeing<DARE>(p, x.vote_sgn, cur.s, seg, local, valid, vec, g, cnt, b, bad_in);
#pragma unroll
            for (int j = 0; j < 8; j++) {
                float acc = 0.0f;
#pragma unroll
                for (int d = 0; d < STOCK_MAX; d++)
                    if (d < p.n_deltas) acc += g[d][j];                                              // filtered_delta.sum(dim=0)
                const float tt = (j < valid) ? sp.t[row_base + (local + j) / row_len] : 0.0f;
                float q;
                if (p.raw) {
                    q = tt * (acc / cnt[j]);                                                         // t * (sum(deltas) / len(deltas))  (ties.py:141-143)
                    // torch.nan_to_num with its defaults: nan -> 0, +-inf -> the largest finite fp32 of that sign
                    if (q != q) q = 0.0f;
                    else if (nonfinite_f32(q)) q = __builtin_copysignf(3.4028234663852886e38f, q);
                } else {
                    q = (acc * tt) / cnt[j];                                                         // ties.py:92
                    if (nonfinite_f32(q)) q = 0.0f;                                                  // nan_to_num(nan=0, posinf=0, neginf=0)
                }
                if (x.has_rescale) q = q / x.rescale_div;                                            // ties.py:50
                if (p.base != nullptr) q = __builtin_fmaf(q, x.alpha, b[j]);                         // base.addcmul(ties, alpha)  (linear.py:76)
                bad_out |= (j < valid) && nonfinite_f32(q);
                r[j] = q;
            }
            store_rt(p.out_dt, vec, out, seg.offset + local, valid, r);
            if ((bad_in | bad_out) && flags != nullptr)
                atomicOr(flags + (p.tile_start == nullptr ? 0 : cur.s), (bad_in ? MECHA_FLAG_INPUT_NONFINITE : 0u) | (bad_out ? MECHA_FLAG_OUTPUT_NONFINITE : 0u));
        }
    }
}

// ------------------------------------------------------------------------------------------------ dropout (n-supermario)
// dropout (ties.py:202-238) at odd whole `overlap` (its default 1.0; the other settings go through torch.lerp on numpy arrays
// upstream and raise): every delta keeps an element where ITS host-drawn Bernoulli mask is set,
//   out = sum_m (mask_m ? delta_m : 0)  /  max(sum_m mask_m, 1)  /  (1 - probability) ** rescale
// Sums in index order from +0 (final_delta[mask] += delta[mask], :226-227); two fp32 divisions, as the reference's two `/`.
struct DropoutParams {
    const void *in[MECHA_MAX_MODELS];
    const uint32_t *keep_bits;
    int64_t mask_words;
    int n;
    float rescalar;
};
template <int IN_DT, int OUT_DT>
__global__ __launch_bounds__(MECHA_BLOCK) void dropout_kernel(DropoutParams p, void *out, int64_t numel, bool vec_ok, uint32_t *flags) {
    const int64_t e = ((int64_t)blockIdx.x * MECHA_BLOCK + threadIdx.x) * 8;          // flat grid: one vector position per lane
    if (e >= numel) return;
    const int valid = (numel - e >= 8) ? 8 : (int)(numel - e);
    const bool vec = vec_ok && valid == 8;
    float acc[8], cnt[8];
#pragma unroll
    for (int j = 0; j < 8; j++) { acc[j] = 0.0f; cnt[j] = 0.0f; }
    bool bad_in = false;
    for (int m = 0; m < p.n; m++) {
        float v[8];
        if (vec) load8<IN_DT, true>(p.in[m], e, v);
        else {
#pragma unroll
            for (int j = 0; j < 8; j++) v[j] = (j < valid) ? load1<IN_DT>(p.in[m], e + j) : 0.0f;
        }
        const uint32_t word = p.keep_bits[(int64_t)m * p.mask_words + (e >> 5)];
        const uint32_t bits8 = (word >> (e & 31)) & 0xffu;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            bad_in |= nonfinite_f32(v[j]);
            if ((bits8 >> j) & 1u) { acc[j] += v[j]; cnt[j] += 1.0f; }
        }
    }
    float r[8];
    bool bad_out = false;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        r[j] = (acc[j] / __builtin_fmaxf(cnt[j], 1.0f)) / p.rescalar;
        bad_out |= (j < valid) && nonfinite_f32(r[j]);
    }
    if (vec) store8<OUT_DT, true>(out, e, r);
    else {
#pragma unroll
        for (int j = 0; j < 8; j++)
            if (j < valid) store1<OUT_DT>(out, e + j, r[j]);
    }
    if ((bad_in | bad_out) && flags != nullptr)
        atomicOr(flags, (bad_in ? MECHA_FLAG_INPUT_NONFINITE : 0u) | (bad_out ? MECHA_FLAG_OUTPUT_NONFINITE : 0u));
}

// ------------------------------------------------------------------------------------------------ host
size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct WorkspaceLayout {
    size_t tile_start_off, tile_rec_off, state_off, counts_off, fill_off, globals_off, hist_off, amb_elem_off, amb_bits_off, amb_sd_off,
        hit_elem_off, hit_seg_off, total;
    size_t amb_capacity;
};
WorkspaceLayout layout(int n_deltas, int n_segs, int64_t total_elems) {
    WorkspaceLayout w;
    const size_t n_sd = (size_t)n_segs * n_deltas;
    w.amb_capacity = n_sd * AMB_BASE_CAP + (size_t)n_deltas * ((size_t)(total_elems > 0 ? total_elems : 0) / AMB_DIV + 1);
    size_t off = 0;
    w.tile_start_off = off; off = align_up(off + sizeof(long long) * ((size_t)n_segs + 1), 256);
    // batched only (n_segs > 1 or an arena): at most total_elems / TILE + n_segs tiles
    const size_t max_tiles = (size_t)(total_elems > 0 ? total_elems : 0) / TIES_TILE + (size_t)n_segs;
    w.tile_rec_off = off;   off = align_up(off + sizeof(TileRec) * max_tiles, 256);
    w.state_off = off;      off = align_up(off + sizeof(SelectState) * n_sd, 256);
    w.counts_off = off;     off = align_up(off + sizeof(SelectCounts) * n_sd, 256);
    w.fill_off = off;       off = align_up(off + sizeof(uint32_t) * n_sd, 256);
    w.globals_off = off;    off = align_up(off + sizeof(TiesGlobals), 256);
    w.hist_off = off;       off = align_up(off + sizeof(uint32_t) * n_sd * TIES_BINS, 256);
    w.amb_elem_off = off;   off = align_up(off + sizeof(long long) * w.amb_capacity, 256);
    w.amb_bits_off = off;   off = align_up(off + sizeof(uint32_t) * w.amb_capacity, 256);
    w.amb_sd_off = off;     off = align_up(off + sizeof(uint32_t) * w.amb_capacity, 256);
    w.hit_elem_off = off;   off = align_up(off + sizeof(long long) * w.amb_capacity, 256);
    w.hit_seg_off = off;    off = align_up(off + sizeof(uint32_t) * w.amb_capacity, 256);
    w.total = off;
    return w;
}

size_t stock_bytes(int n_deltas, int64_t n_rows) {
    return align_up(sizeof(double) * (size_t)n_rows * (2 * n_deltas - 1), 256) + align_up(sizeof(float) * (size_t)n_rows, 256);
}

struct TiesCall {
    const void *base;
    const void *const *inputs;
    int n_deltas;
    const uint32_t *keep_bits;
    void *out;
    const mecha_segment *segs_dev;          // batched
    const mecha_segment *segs_host;         // batched
    int n_segs;
    int64_t numel;                          // per key
    int64_t total_elems;                    // batched: arena length the workspace was sized for
    int in_dt, out_dt;
    double k;
    float alpha;
    int vote_sgn;
    float probability;
    int rescale;
    int mask_stream;                        // keep_bits is one shared Bernoulli stream (TiesParams::mask_stream)
    int model_stock;                        // ties_sum_extended: apply the model-stock factor (per-key delta form only)
    float cos_eps;
    int64_t row_len;                        // size of the key's last dimension
    const int64_t *row_lens_dev;            // batched + model_stock: every key's last dimension (DEVICE [n_segs]) ...
    const int64_t *row_lens_host;           // ... and the HOST copy
    float *thresholds_out;
    void *workspace;
    size_t workspace_bytes;
    uint32_t *flags;
    void *stream;
};

// Blocks of a persistent kernel that are resident at one time (asked once per kernel instantiation and LDS size): a grid of
// exactly that many walks its tile ranges in one round, a larger one trails a partly filled second round behind the first
// (the count pass: 4.60 -> 4.38 ms).
template <typename K>
unsigned resident_grid(K kernel, size_t lds, int &cached, long long total_tiles) {
    if (cached == 0) {
        int per_cu = 0, dev = 0, cus = 256;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, MECHA_BLOCK, lds) != hipSuccess || per_cu < 1) { (void)hipGetLastError(); per_cu = 4; }
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) { (void)hipGetLastError(); cus = 256; }
        cached = per_cu * cus;
    }
    return (unsigned)(total_tiles < cached ? total_tiles : cached);
}

template <int PASS, bool DARE>
void launch_hist(const TiesParams &p, unsigned looped_grid, hipStream_t s) {
    constexpr int id = PASS == 1 ? MK_TIES_HIST1 : (PASS == 2 ? MK_TIES_HIST2 : MK_TIES_HIST3);
    static int resident[3][TIES_GROUP + 1] = {};                     // [register capacity][deltas of the launch]
    (void)looped_grid;
    for (int d0 = 0; d0 < p.n_deltas; d0 += TIES_GROUP) {
        const int nd = (p.n_deltas - d0 < TIES_GROUP) ? p.n_deltas - d0 : TIES_GROUP;
        const size_t lds = (size_t)nd * TIES_BINS * sizeof(uint32_t);
        const unsigned grid = nd <= 2 ? resident_grid(ties_hist_kernel<PASS, 2, DARE>, lds, resident[0][nd], p.total_tiles)
                            : nd <= 4 ? resident_grid(ties_hist_kernel<PASS, 4, DARE>, lds, resident[1][nd], p.total_tiles)
                                      : resident_grid(ties_hist_kernel<PASS, 8, DARE>, lds, resident[2][nd], p.total_tiles);
        if (nd <= 2) MECHA_LAUNCH(id, (ties_hist_kernel<PASS, 2, DARE>), dim3(grid), dim3(MECHA_BLOCK), lds, s, p, d0, nd);
        else if (nd <= 4) MECHA_LAUNCH(id, (ties_hist_kernel<PASS, 4, DARE>), dim3(grid), dim3(MECHA_BLOCK), lds, s, p, d0, nd);
        else MECHA_LAUNCH(id, (ties_hist_kernel<PASS, 8, DARE>), dim3(grid), dim3(MECHA_BLOCK), lds, s, p, d0, nd);
    }
}

#ifndef MECHA_COUNT_BLOCKS_PER_CU
#define MECHA_COUNT_BLOCKS_PER_CU 0          // 0: as many as the kernel's occupancy (tuning switch)
#endif
void launch_count(const TiesParams &p, unsigned looped_grid, hipStream_t s) {
    // 4 histograms per block (33 KiB of LDS, 4 blocks per CU).  7 fit in 64 KiB and read the base once instead of twice for a
    // FLUX x8 job, but at 2 blocks per CU the pass is latency-bound: measured on one FLUX stripe 6.18 ms in one launch of 7
    // against 2 x 2.48 ms in groups of 4 + 3 (groups of 3: 5.56 ms, of 2: 5.77 ms).
    constexpr int COUNT_GROUP = 4;
    for (int d0 = 0; d0 < p.n_deltas; d0 += COUNT_GROUP) {
        const int nd = (p.n_deltas - d0 < COUNT_GROUP) ? p.n_deltas - d0 : COUNT_GROUP;
        // LDS: per delta its below words, histogram and write-only words (COUNT_STRIDE)
        const int cap = nd <= 2 ? 2 : (nd <= 3 ? 3 : 4);
        const size_t lds = (size_t)nd * COUNT_STRIDE * sizeof(uint32_t);
        const bool w32 = p.in_dt == MECHA_F32, has_base = p.base != nullptr;
        // persistent blocks over contiguous tile ranges: exactly as many as are resident at once (5 per CU for three 16-bit deltas,
        // else 4), so that no second, partly filled round of blocks trails the first
        const unsigned per_cu = MECHA_COUNT_BLOCKS_PER_CU ? MECHA_COUNT_BLOCKS_PER_CU : (unsigned)count_waves_per_simd(cap, w32, p.keep_bits != nullptr);
        const unsigned grid = (unsigned)(p.total_tiles < (long long)(256u * per_cu) ? p.total_tiles : 256u * per_cu);
        (void)looped_grid;
#define MECHA_COUNT_LAUNCH_B(CAP_, W32_, BASE_)                                                                                                       \
    do {                                                                                                                                          \
        if (p.keep_bits != nullptr) MECHA_LAUNCH(MK_TIES_COUNT, (ties_count_kernel<CAP_, W32_, BASE_, true>), dim3(grid), dim3(MECHA_BLOCK), lds, s, p, d0, nd); \
        else MECHA_LAUNCH(MK_TIES_COUNT, (ties_count_kernel<CAP_, W32_, BASE_>), dim3(grid), dim3(MECHA_BLOCK), lds, s, p, d0, nd);                 \
    } while (0)
#define MECHA_COUNT_LAUNCH(CAP_, W32_) do { if (has_base) MECHA_COUNT_LAUNCH_B(CAP_, W32_, true); else MECHA_COUNT_LAUNCH_B(CAP_, W32_, false); } while (0)
        if (cap == 2) { if (w32) MECHA_COUNT_LAUNCH(2, true); else MECHA_COUNT_LAUNCH(2, false); }
        else if (cap == 3) { if (w32) MECHA_COUNT_LAUNCH(3, true); else MECHA_COUNT_LAUNCH(3, false); }
        else { if (w32) MECHA_COUNT_LAUNCH(4, true); else MECHA_COUNT_LAUNCH(4, false); }
#undef MECHA_COUNT_LAUNCH
#undef MECHA_COUNT_LAUNCH_B
    }
}

template <bool DARE>
void launch_radix(const TiesParams &p_in, unsigned grid, double one_minus_k, float *thresholds_out, hipStream_t s, int64_t max_numel = 0) {
    TiesParams p = p_in;
    const unsigned scan_grid = (unsigned)((int64_t)p.n_segs * p.n_deltas);
    if constexpr (DARE) {
        if (!p.bracket) {
            const int64_t slices = ((max_numel + 31) / 32 + 1 + PRESOLVE_SLICE_WORDS - 1) / PRESOLVE_SLICE_WORDS;
            MECHA_LAUNCH(MK_TIES_SAMPLE, ties_dare_presolve_kernel, dim3(scan_grid, (unsigned)slices), dim3(MECHA_BLOCK), 0, s, p);
            MECHA_LAUNCH(MK_TIES_SAMPLE, ties_dare_presolve_decide_kernel, dim3((scan_grid + MECHA_BLOCK - 1) / MECHA_BLOCK), dim3(MECHA_BLOCK), 0, s, p,
                         (int64_t)scan_grid, one_minus_k, thresholds_out);
            p.presolved = 1;
        }
    }
    launch_hist<1, DARE>(p, grid, s);
    MECHA_LAUNCH(MK_TIES_SCAN, (ties_scan_kernel<1>), dim3(scan_grid), dim3(MECHA_BLOCK), 0, s, p, one_minus_k, thresholds_out);
    launch_hist<2, DARE>(p, grid, s);
    MECHA_LAUNCH(MK_TIES_SCAN, (ties_scan_kernel<2>), dim3(scan_grid), dim3(MECHA_BLOCK), 0, s, p, one_minus_k, thresholds_out);
    launch_hist<3, DARE>(p, grid, s);
    MECHA_LAUNCH(MK_TIES_SCAN, (ties_scan_kernel<3>), dim3(scan_grid), dim3(MECHA_BLOCK), 0, s, p, one_minus_k, thresholds_out);
}

template <int N, int DT, bool DARE>
void launch_merge_full(const TiesParams &p, const MergeExtra &x, unsigned /*looped_grid*/, void *out, uint32_t *flags, hipStream_t s) {
    constexpr int E = N <= 4 ? 8 : 4;                             // elements per lane (see the kernel; 4 for three models: 5.6-5.8 against 5.5 ms)
    const unsigned grid = (unsigned)(p.total_tiles * (TIES_TILE / (MECHA_BLOCK * E)));          // flat: one block per 256 x E elements
    if (p.out_dt == MECHA_BF16) MECHA_LAUNCH(MK_TIES_MERGE, (ties_merge_full_kernel<N, DT, MECHA_BF16, E, DARE>), dim3(grid), dim3(MECHA_BLOCK), 0, s, p, x, out, flags);
    else if (p.out_dt == MECHA_F16) MECHA_LAUNCH(MK_TIES_MERGE, (ties_merge_full_kernel<N, DT, MECHA_F16, E, DARE>), dim3(grid), dim3(MECHA_BLOCK), 0, s, p, x, out, flags);
    else MECHA_LAUNCH(MK_TIES_MERGE, (ties_merge_full_kernel<N, DT, MECHA_F32, E, DARE>), dim3(grid), dim3(MECHA_BLOCK), 0, s, p, x, out, flags);
}

template <bool DARE>
void launch_merge(const TiesParams &p, const MergeExtra &x_in, unsigned grid, void *out, uint32_t *flags, hipStream_t s) {
    const bool w32 = p.in_dt == MECHA_F32;
    MergeExtra x = x_in;
    // the common case goes to ties_merge_full_kernel; ties_merge_kernel then only sees what that kernel leaves (timed as
    // "ties_fixup" so that the per-launch average of "ties_merge" stays that of the full-tile kernel)
    const bool full = p.base != nullptr && !w32 && p.vec_ok && !x.vote_sgn && (DARE || !x.has_rescale) && p.n_deltas <= 8 && out != nullptr &&
                      mecha_aligned16(out);
    if (full) {
        x.skip_full_tiles = 1;
        x.full_any_thr = DARE ? 1 : 0;
#define MECHA_FULL_CASE(N_) case N_: if (p.in_dt == MECHA_BF16) launch_merge_full<N_, MECHA_BF16, DARE>(p, x, grid, out, flags, s); \
                                      else launch_merge_full<N_, MECHA_F16, DARE>(p, x, grid, out, flags, s); break;
        switch (p.n_deltas) {
            MECHA_FULL_CASE(1) MECHA_FULL_CASE(2) MECHA_FULL_CASE(3) MECHA_FULL_CASE(4)
            MECHA_FULL_CASE(5) MECHA_FULL_CASE(6) MECHA_FULL_CASE(7)
            default: if (p.in_dt == MECHA_BF16) launch_merge_full<8, MECHA_BF16, DARE>(p, x, grid, out, flags, s);
                     else launch_merge_full<8, MECHA_F16, DARE>(p, x, grid, out, flags, s);
        }
#undef MECHA_FULL_CASE
    }
    const int id = full ? MK_TIES_FIXUP : MK_TIES_MERGE;
#define MECHA_MERGE_LAUNCH(CAP_, W32_) MECHA_LAUNCH(id, (ties_merge_kernel<CAP_, DARE, W32_>), dim3(grid), dim3(MECHA_BLOCK), 0, s, p, x, out, flags)
    if (p.n_deltas <= 2) { if (w32) MECHA_MERGE_LAUNCH(2, true); else MECHA_MERGE_LAUNCH(2, false); }
    else if (p.n_deltas <= 4) { if (w32) MECHA_MERGE_LAUNCH(4, true); else MECHA_MERGE_LAUNCH(4, false); }
    else { if (w32) MECHA_MERGE_LAUNCH(8, true); else MECHA_MERGE_LAUNCH(8, false); }
#undef MECHA_MERGE_LAUNCH
}

int run_ties(const TiesCall &c) {
    if (c.segs_dev == nullptr && c.numel == 0 && c.n_deltas >= 1) return MECHA_OK;   // ties.py:106-107: empty tensor, nothing to do
    MECHA_REQUIRE(c.out != nullptr, "out is NULL");
    MECHA_REQUIRE(c.inputs != nullptr && c.n_deltas >= 1 && c.n_deltas <= TIES_MAX_DELTAS - (c.base ? 1 : 0),
                  "TIES needs 1..%d deltas (got %d)", TIES_MAX_DELTAS - (c.base ? 1 : 0), c.n_deltas);
    MECHA_REQUIRE(mecha_valid_dtype(c.in_dt) && mecha_valid_dtype(c.out_dt), "unknown dtype code (in %d, out %d)", c.in_dt, c.out_dt);
    // k > 1 is legal: kth = max(int((1 - k) * numel), 1) clamps the rank to 1 (ties.py:129); k < 0 asks for a rank beyond numel
    MECHA_REQUIRE(c.k >= 0.0, "k must be >= 0 (got %g): the rank would exceed numel and torch.kthvalue would raise", c.k);
    const bool batched = c.segs_dev != nullptr;
    long long total_tiles = 0;
    int64_t max_numel = 0, sum_numel = 0;
    if (batched) {
        MECHA_REQUIRE(c.segs_host != nullptr && c.n_segs >= 1, "batched TIES needs host and device segment tables");
        int64_t prev_end = 0;
        for (int s = 0; s < c.n_segs; s++) {
            MECHA_REQUIRE(c.segs_host[s].numel > 0 && (c.segs_host[s].offset % 8) == 0 && c.segs_host[s].offset >= prev_end,
                          "segment %d: offsets must be multiples of 8, sorted and non-overlapping, numel > 0", s);
            prev_end = c.segs_host[s].offset + c.segs_host[s].numel;
            total_tiles += (c.segs_host[s].numel + TIES_TILE - 1) / TIES_TILE;
            if (c.segs_host[s].numel > max_numel) max_numel = c.segs_host[s].numel;
            sum_numel += c.segs_host[s].numel;
        }
        MECHA_REQUIRE(sum_numel <= c.total_elems, "segments hold %lld elements, more than total_elems = %lld", (long long)sum_numel, (long long)c.total_elems);
    } else {
        MECHA_REQUIRE(c.numel >= 0, "numel must be >= 0");
        total_tiles = (c.numel + TIES_TILE - 1) / TIES_TILE;
        max_numel = sum_numel = c.numel;
    }
    const int n_segs = batched ? c.n_segs : 1;
    MECHA_REQUIRE(!(c.base != nullptr && mecha_isclose(c.alpha, 0.0)),
                  "alpha == 0: add_difference returns `a` untouched (linear.py:72-73); call the cast entry point on base instead");
    const WorkspaceLayout w = layout(c.n_deltas, n_segs, batched ? c.total_elems : c.numel);
    size_t need = w.total;
    int64_t n_rows = 0;
    size_t row_starts_bytes = 0;
    if (c.model_stock) {
        MECHA_REQUIRE(c.n_deltas >= 2 && c.n_deltas <= STOCK_MAX, "the model-stock factor needs 2..%d deltas (got %d): the reference's "
                      "torch.stack of an empty list of cosines raises for one delta", STOCK_MAX, c.n_deltas);
        if (batched) {
            MECHA_REQUIRE(c.row_lens_dev != nullptr && c.row_lens_host != nullptr, "batched model-stock factor needs the keys' last dimensions (host and device)");
            for (int s = 0; s < c.n_segs; s++) {
                MECHA_REQUIRE(c.row_lens_host[s] >= 1 && c.segs_host[s].numel % c.row_lens_host[s] == 0, "segment %d: row_len (%lld) must divide numel (%lld)",
                              s, (long long)c.row_lens_host[s], (long long)c.segs_host[s].numel);
                n_rows += c.segs_host[s].numel / c.row_lens_host[s];
            }
            row_starts_bytes = align_up(sizeof(int64_t) * ((size_t)c.n_segs + 1), 256);
        } else {
            MECHA_REQUIRE(c.row_len >= 1 && c.numel % c.row_len == 0, "row_len (%lld) must divide numel (%lld)", (long long)c.row_len, (long long)c.numel);
            n_rows = c.numel / c.row_len;
        }
        need += row_starts_bytes + stock_bytes(c.n_deltas, n_rows);
    }
    MECHA_REQUIRE(c.workspace != nullptr && c.workspace_bytes >= need && mecha_aligned16(c.workspace),
                  "TIES workspace too small or unaligned: need %zu bytes, got %zu", need, c.workspace_bytes);
    if (c.keep_bits != nullptr) MECHA_REQUIRE(c.mask_stream || (!batched && c.base == nullptr), "DARE masks in row form are supported on the per-key delta form only");
    MECHA_REQUIRE(total_tiles * (TIES_TILE / 4) < (1ll << 32), "%lld tiles are beyond one launch of the flat merge kernel", total_tiles);

    bool vec = mecha_aligned16(c.out) && (c.base == nullptr || mecha_aligned16(c.base));
    for (int m = 0; m < c.n_deltas; m++) {
        MECHA_REQUIRE(c.inputs[m] != nullptr, "input %d is NULL", m);
        vec = vec && mecha_aligned16(c.inputs[m]);
    }
    hipStream_t s = static_cast<hipStream_t>(c.stream);
    char *ws = static_cast<char *>(c.workspace);
    const bool dare = c.keep_bits != nullptr;
    // MECHA_TIES_SELECT=radix forces pipeline B; =fail makes every bracket miss (exercises the per-pair fallback)
    const char *env = getenv("MECHA_TIES_SELECT");
    const bool force_radix = env != nullptr && strcmp(env, "radix") == 0;
    const int force_fail = (env != nullptr && strcmp(env, "fail") == 0) ? 1 : 0;
    const bool bracket = !dare && !force_radix && !c.model_stock;     // the stock kernels want final thresholds before they run

    TiesParams p{};
    p.base = c.base;
    for (int m = 0; m < c.n_deltas; m++) p.in[m] = c.inputs[m];
    p.keep_bits = c.keep_bits;
    p.mask_words = batched ? 0 : (c.numel + 31) / 32;
    p.mask_stream = c.mask_stream;
    p.n_deltas = c.n_deltas;
    p.in_dt = c.in_dt;
    p.out_dt = c.out_dt;
    p.vec_ok = vec ? 1 : 0;
    p.segs = batched ? c.segs_dev : nullptr;
    p.seg0 = mecha_segment{0, batched ? 0 : c.numel};
    p.n_segs = n_segs;
    p.tile_start = batched ? reinterpret_cast<const long long *>(ws + w.tile_start_off) : nullptr;
    p.tile_rec = batched ? reinterpret_cast<const TileRec *>(ws + w.tile_rec_off) : nullptr;
    p.total_tiles = total_tiles;
    p.state = reinterpret_cast<SelectState *>(ws + w.state_off);
    p.hist = reinterpret_cast<uint32_t *>(ws + w.hist_off);
    p.bracket = bracket ? 1 : 0;
    p.counts = reinterpret_cast<SelectCounts *>(ws + w.counts_off);
    p.fill = reinterpret_cast<uint32_t *>(ws + w.fill_off);
    p.globals = reinterpret_cast<TiesGlobals *>(ws + w.globals_off);
    p.amb_elem = reinterpret_cast<long long *>(ws + w.amb_elem_off);
    p.amb_bits = reinterpret_cast<uint32_t *>(ws + w.amb_bits_off);
    p.amb_sd = reinterpret_cast<uint32_t *>(ws + w.amb_sd_off);
    p.hit_elem = reinterpret_cast<long long *>(ws + w.hit_elem_off);
    p.hit_seg = reinterpret_cast<uint32_t *>(ws + w.hit_seg_off);
    p.amb_capacity = w.amb_capacity;

    const size_t n_sd = (size_t)n_segs * c.n_deltas;
    MECHA_REQUIRE(n_sd < ((size_t)1 << 31), "too many (key, delta) pairs");
    // state, fill, globals are contiguous up to the histograms: one memset clears them all
    MECHA_CHECK_HIP(hipMemsetAsync(ws + w.state_off, 0, w.amb_elem_off - w.state_off, s));
    if (batched) {
        MECHA_REQUIRE((size_t)total_tiles <= (size_t)(c.total_elems > 0 ? c.total_elems : 0) / TIES_TILE + (size_t)n_segs,
                      "segments hold more tiles than the workspace was sized for");
        MECHA_LAUNCH(MK_TIES_INDEX, ties_tile_index_kernel, dim3(1), dim3(1024), 0, s, c.segs_dev, n_segs,
                     reinterpret_cast<long long *>(ws + w.tile_start_off));
        MECHA_LAUNCH(MK_TIES_INDEX, ties_tile_records_kernel, dim3((unsigned)n_segs), dim3(MECHA_BLOCK), 0, s, c.segs_dev,
                     reinterpret_cast<const long long *>(ws + w.tile_start_off), reinterpret_cast<TileRec *>(ws + w.tile_rec_off));
    }
    unsigned grid = (unsigned)(total_tiles < MECHA_MAX_GRID ? total_tiles : MECHA_MAX_GRID);
    const double one_minus_k = 1.0 - c.k;
    MergeExtra x{};
    x.alpha = c.alpha;
    x.vote_sgn = c.vote_sgn;
    x.has_rescale = dare && c.rescale;
    x.rescale_div = 1.0f - c.probability;                 // 1.0 - probability as fp32 tensor arithmetic (ties.py:49)
    x.rescale_rcp = 1.0f / x.rescale_div;
    {
        uint32_t rb; memcpy(&rb, &x.rescale_div, 4); rb &= 0x7fffffffu;
        x.fast_rescale = (rb >= 0x21800000u && rb <= 0x5d800000u && (rb & 0x7fffffu) != 0x7fffffu) ? 1 : 0;
    }

    if (c.model_stock) {
        StockParams sp{};
        sp.row_len = c.row_len; sp.n_rows = n_rows; sp.cos_eps = c.cos_eps;
        char *stock = ws + w.total;
        if (batched) {
            sp.row_lens = c.row_lens_dev;
            sp.row_starts = reinterpret_cast<const int64_t *>(stock);
            MECHA_LAUNCH(MK_TIES_INDEX, ties_stock_rows_kernel, dim3(1), dim3(1024), 0, s, c.segs_dev, c.row_lens_dev, n_segs, reinterpret_cast<int64_t *>(stock));
            stock += row_starts_bytes;
        }
        sp.sums = reinterpret_cast<double *>(stock);
        sp.t = reinterpret_cast<float *>(stock + align_up(sizeof(double) * (size_t)n_rows * (2 * c.n_deltas - 1), 256));
        MECHA_CHECK_HIP(hipMemsetAsync(sp.sums, 0, sizeof(double) * (size_t)n_rows * (2 * c.n_deltas - 1), s));
        const unsigned sgrid = (unsigned)(total_tiles < 4 * MECHA_MAX_GRID ? total_tiles : 4 * MECHA_MAX_GRID);
        const int64_t tblocks = (n_rows + MECHA_BLOCK - 1) / MECHA_BLOCK;
        const unsigned tgrid = (unsigned)(tblocks < 65536 ? tblocks : 65536);
        if (dare) {
            launch_radix<true>(p, grid, one_minus_k, c.thresholds_out, s, max_numel);
            MECHA_LAUNCH(MK_TIES_MERGE, (ties_stock_reduce_kernel<true>), dim3(sgrid), dim3(MECHA_BLOCK), 0, s, p, sp, c.vote_sgn, c.flags);
            MECHA_LAUNCH(MK_TIES_MERGE, ties_stock_t_kernel, dim3(tgrid), dim3(MECHA_BLOCK), 0, s, sp, c.n_deltas, (const int64_t *)nullptr);
            MECHA_LAUNCH(MK_TIES_MERGE, (ties_stock_apply_kernel<true>), dim3(sgrid), dim3(MECHA_BLOCK), 0, s, p, sp, x, c.out, c.flags);
        } else {
            launch_radix<false>(p, grid, one_minus_k, c.thresholds_out, s);
            MECHA_LAUNCH(MK_TIES_MERGE, (ties_stock_reduce_kernel<false>), dim3(sgrid), dim3(MECHA_BLOCK), 0, s, p, sp, c.vote_sgn, c.flags);
            MECHA_LAUNCH(MK_TIES_MERGE, ties_stock_t_kernel, dim3(tgrid), dim3(MECHA_BLOCK), 0, s, sp, c.n_deltas, (const int64_t *)nullptr);
            MECHA_LAUNCH(MK_TIES_MERGE, (ties_stock_apply_kernel<false>), dim3(sgrid), dim3(MECHA_BLOCK), 0, s, p, sp, x, c.out, c.flags);
        }
    } else if (dare && !force_radix && max_numel > TIES_SAMPLE) {
        // DARE: pairs whose kth value is one of the dropped zeros are settled from the masks alone; the others take the bracket select
        // on the masked deltas (sample + ONE count pass instead of three radix passes), ambiguous values listed and fixed up as in
        // the TIES form; pairs whose bracket missed take the radix passes (which skip every other key).
        const int64_t slices = ((max_numel + 31) / 32 + 1 + PRESOLVE_SLICE_WORDS - 1) / PRESOLVE_SLICE_WORDS;
        MECHA_LAUNCH(MK_TIES_SAMPLE, ties_dare_presolve_kernel, dim3((unsigned)n_sd, (unsigned)slices), dim3(MECHA_BLOCK), 0, s, p);
        MECHA_LAUNCH(MK_TIES_SAMPLE, ties_dare_presolve_decide_kernel, dim3((unsigned)((n_sd + MECHA_BLOCK - 1) / MECHA_BLOCK)), dim3(MECHA_BLOCK), 0, s, p,
                     (int64_t)n_sd, one_minus_k, c.thresholds_out);
        p.presolved = 1;
        p.bracket = 1;
        MECHA_LAUNCH(MK_TIES_SAMPLE, ties_sample_kernel, dim3((unsigned)n_sd), dim3(MECHA_BLOCK), 0, s, p, one_minus_k, c.thresholds_out);
        launch_count(p, grid, s);
        MECHA_LAUNCH(MK_TIES_RESOLVE, ties_resolve_kernel, dim3((unsigned)n_sd), dim3(MECHA_BLOCK), 0, s, p, c.thresholds_out, force_fail, 0);
        p.presolved = 0;                                                 // from here on the radix kernels go by MODE_FAILED alone
        launch_radix<true>(p, grid, one_minus_k, c.thresholds_out, s, max_numel);
        MECHA_LAUNCH(MK_TIES_RESOLVE, ties_amb_offsets_kernel, dim3(1), dim3(1024), 0, s, p, (int64_t)n_sd);
        launch_merge<true>(p, x, grid, c.out, c.flags, s);
        MECHA_LAUNCH(MK_TIES_FIXUP, ties_amb_collect_kernel, dim3(64), dim3(MECHA_BLOCK), 0, s, p);
        MECHA_LAUNCH(MK_TIES_FIXUP, ties_amb_select_kernel, dim3((unsigned)(n_sd < 512 ? n_sd : 512)), dim3(MECHA_BLOCK), 0, s, p, c.thresholds_out, (int64_t)n_sd);
        MECHA_LAUNCH(MK_TIES_FIXUP, ties_fixup_kernel, dim3(256), dim3(MECHA_BLOCK), 0, s, p, x, c.out, c.flags);
    } else if (!bracket) {
        if (dare) { launch_radix<true>(p, grid, one_minus_k, c.thresholds_out, s, max_numel); launch_merge<true>(p, x, grid, c.out, c.flags, s); }
        else { launch_radix<false>(p, grid, one_minus_k, c.thresholds_out, s); launch_merge<false>(p, x, grid, c.out, c.flags, s); }
    } else {
        const bool need_count = max_numel > TIES_SAMPLE;
        MECHA_LAUNCH(MK_TIES_SAMPLE, ties_sample_kernel, dim3((unsigned)n_sd), dim3(MECHA_BLOCK), 0, s, p, one_minus_k, c.thresholds_out);
        if (need_count) {
            launch_count(p, grid, s);
            MECHA_LAUNCH(MK_TIES_RESOLVE, ties_resolve_kernel, dim3((unsigned)n_sd), dim3(MECHA_BLOCK), 0, s, p, c.thresholds_out, force_fail);
            launch_radix<false>(p, grid, one_minus_k, c.thresholds_out, s);     // serves MODE_FAILED pairs only; exits at once if there are none
            MECHA_LAUNCH(MK_TIES_RESOLVE, ties_amb_offsets_kernel, dim3(1), dim3(1024), 0, s, p, (int64_t)n_sd);
        }
        launch_merge<false>(p, x, grid, c.out, c.flags, s);
        if (need_count) {
            MECHA_LAUNCH(MK_TIES_FIXUP, ties_amb_collect_kernel, dim3(64), dim3(MECHA_BLOCK), 0, s, p);
            MECHA_LAUNCH(MK_TIES_FIXUP, ties_amb_select_kernel, dim3((unsigned)(n_sd < 512 ? n_sd : 512)), dim3(MECHA_BLOCK), 0, s, p, c.thresholds_out, (int64_t)n_sd);
            MECHA_LAUNCH(MK_TIES_FIXUP, ties_fixup_kernel, dim3(256), dim3(MECHA_BLOCK), 0, s, p, x, c.out, c.flags);
        }
    }
    MECHA_CHECK_HIP(hipGetLastError());
    return MECHA_OK;
}

// ------------------------------------------------------------------------------------------------ allocation-free entry point
// mecha_add_difference_ties_noalloc runs the same kernels on a workspace that lives in the library's own device image:
// callers that cannot call hipMalloc (HIP graph capture, serving loops, harnesses that forbid it) need no workspace.
// Every GPU the library runs on holds its own copy of the array until the process ends.  In HOST code `scratch` is only a
// stand-in for the symbol, not a device address: the per-device address comes from hipGetSymbolAddress, once per device,
// and only that pointer is handed to kernels and memsets.
constexpr size_t NOALLOC_SCRATCH_BYTES = (size_t)8 << 20;
constexpr int NOALLOC_MAX_MODELS = 8;
constexpr int64_t NOALLOC_MAX_NUMEL = (int64_t)1 << 26;
constexpr int NOALLOC_MAX_DEVICES = 64;

}  // namespace

#include <mutex>

// File scope, outside the anonymous namespace: a `static __device__` variable that host code names is given an external
// device symbol by the compiler, which is how hipGetSymbolAddress finds it; inside an anonymous namespace the device symbol
// stays local and the runtime aborts ("Cannot create GlobalVar Obj", tools/scratch_probe/).
static __device__ __align__(256) unsigned char scratch[NOALLOC_SCRATCH_BYTES];

namespace {

std::mutex g_scratch_mutex;
void *g_scratch_ptr[NOALLOC_MAX_DEVICES];          // device address of `scratch` on each device, 0 = not looked up yet

// The current device's copy of `scratch`; looked up on the first call on that device (with it current).
int scratch_for_current_device(void **out) {
    int dev = -1;
    MECHA_CHECK_HIP(hipGetDevice(&dev));
    MECHA_REQUIRE(dev >= 0 && dev < NOALLOC_MAX_DEVICES, "device index %d outside 0..%d", dev, NOALLOC_MAX_DEVICES - 1);
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (g_scratch_ptr[dev] == nullptr) {
        void *ptr = nullptr;
        MECHA_CHECK_HIP(hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)));
        MECHA_REQUIRE(ptr != nullptr, "hipGetSymbolAddress returned a null address for the scratch array");
        g_scratch_ptr[dev] = ptr;
    }
    *out = g_scratch_ptr[dev];
    return MECHA_OK;
}

}  // namespace

extern "C" {

size_t mecha_ties_workspace_bytes(int n_models, int64_t numel) {
    if (n_models < 1) n_models = 1;
    return layout(n_models, 1, numel).total;
}

size_t mecha_batched_ties_workspace_bytes(int n_models, int n_segments, int64_t total_elems) {
    if (n_models < 1) n_models = 1;
    if (n_segments < 1) n_segments = 1;
    return layout(n_models, n_segments, total_elems).total;
}

int mecha_ties_select_stats(const void *workspace, int n_models, int n_segments, int64_t total_elems,
                            unsigned long long *failed_pairs, unsigned long long *ambiguous_elems, void *stream) {
    MECHA_REQUIRE(workspace != nullptr && failed_pairs != nullptr && ambiguous_elems != nullptr, "NULL argument");
    const WorkspaceLayout w = layout(n_models < 1 ? 1 : n_models, n_segments < 1 ? 1 : n_segments, total_elems);
    TiesGlobals g{};
    hipStream_t s = static_cast<hipStream_t>(stream);
    MECHA_CHECK_HIP(hipMemcpyAsync(&g, static_cast<const char *>(workspace) + w.globals_off, sizeof(g), hipMemcpyDeviceToHost, s));
    MECHA_CHECK_HIP(hipStreamSynchronize(s));
    *failed_pairs = g.n_failed;
    *ambiguous_elems = g.n_amb;
    return MECHA_OK;
}

int mecha_ties_sum(const void *const *deltas, int n_models, void *out, int64_t numel, int in_dtype, int out_dtype, double k,
                   int vote_sgn, float *thresholds_out, void *workspace, size_t workspace_bytes, uint32_t *flags, void *stream) {
    TiesCall c{};
    c.inputs = deltas; c.n_deltas = n_models; c.out = out; c.numel = numel; c.in_dt = in_dtype; c.out_dt = out_dtype;
    c.k = k; c.vote_sgn = vote_sgn; c.thresholds_out = thresholds_out; c.workspace = workspace;
    c.workspace_bytes = workspace_bytes; c.flags = flags; c.stream = stream;
    return run_ties(c);
}

int mecha_add_difference_ties(const void *base, const void *const *models, int n_models, void *out, int64_t numel,
                              int in_dtype, int out_dtype, double k, float alpha, int vote_sgn, float *thresholds_out,
                              void *workspace, size_t workspace_bytes, uint32_t *flags, void *stream) {
    MECHA_REQUIRE(base != nullptr, "base is NULL");
    TiesCall c{};
    c.base = base; c.inputs = models; c.n_deltas = n_models; c.out = out; c.numel = numel; c.in_dt = in_dtype;
    c.out_dt = out_dtype; c.k = k; c.alpha = alpha; c.vote_sgn = vote_sgn; c.thresholds_out = thresholds_out;
    c.workspace = workspace; c.workspace_bytes = workspace_bytes; c.flags = flags; c.stream = stream;
    return run_ties(c);
}

int mecha_add_difference_ties_noalloc(const void *base, const void *const *models, int n_models, void *out, int64_t numel,
                                      int in_dtype, int out_dtype, double k, float alpha, int vote_sgn, float *thresholds_out,
                                      uint32_t *flags, void *stream) {
    MECHA_REQUIRE(base != nullptr, "base is NULL");
    MECHA_REQUIRE(n_models >= 1 && n_models <= NOALLOC_MAX_MODELS, "the allocation-free entry point takes 1..%d models (got %d): "
                  "use mecha_add_difference_ties with a caller workspace", NOALLOC_MAX_MODELS, n_models);
    MECHA_REQUIRE(numel >= 0 && numel <= NOALLOC_MAX_NUMEL, "the allocation-free entry point takes keys of up to %lld elements (got %lld): "
                  "use mecha_add_difference_ties with a caller workspace", (long long)NOALLOC_MAX_NUMEL, (long long)numel);
    const size_t need = layout(n_models, 1, numel).total;
    MECHA_REQUIRE(need <= NOALLOC_SCRATCH_BYTES, "the key needs %zu bytes of workspace, the built-in scratch holds %zu", need, NOALLOC_SCRATCH_BYTES);
    void *ptr = nullptr;
    const int st = scratch_for_current_device(&ptr);
    if (st != MECHA_OK) return st;
    TiesCall c{};
    c.base = base; c.inputs = models; c.n_deltas = n_models; c.out = out; c.numel = numel; c.in_dt = in_dtype;
    c.out_dt = out_dtype; c.k = k; c.alpha = alpha; c.vote_sgn = vote_sgn; c.thresholds_out = thresholds_out;
    c.workspace = ptr; c.workspace_bytes = NOALLOC_SCRATCH_BYTES; c.flags = flags; c.stream = stream;
    return run_ties(c);       // clears the part of the workspace that must start at zero (state, counters, histograms) itself
}

int mecha_ties_sum_with_dropout(const void *const *deltas, int n_models, const uint32_t *keep_bits, void *out, int64_t numel,
                                int in_dtype, int out_dtype, double k, int vote_sgn, float probability, int rescale,
                                float *thresholds_out, void *workspace, size_t workspace_bytes, uint32_t *flags, void *stream) {
    MECHA_REQUIRE(keep_bits != nullptr, "keep_bits is NULL (draw them with mecha_dare_draw_masks_host)");
    TiesCall c{};
    c.inputs = deltas; c.n_deltas = n_models; c.keep_bits = keep_bits; c.out = out; c.numel = numel; c.in_dt = in_dtype;
    c.out_dt = out_dtype; c.k = k; c.vote_sgn = vote_sgn; c.probability = probability; c.rescale = rescale;
    c.thresholds_out = thresholds_out; c.workspace = workspace; c.workspace_bytes = workspace_bytes; c.flags = flags;
    c.stream = stream;
    return run_ties(c);
}

size_t mecha_ties_extended_workspace_bytes(int n_models, int64_t numel, int64_t row_len) {
    if (n_models < 1) n_models = 1;
    if (row_len < 1) row_len = 1;
    return layout(n_models, 1, numel).total + stock_bytes(n_models, (numel + row_len - 1) / row_len);
}

int mecha_ties_sum_extended(const void *const *deltas, int n_models, const uint32_t *keep_bits, void *out, int64_t numel, int64_t row_len,
                            int in_dtype, int out_dtype, double k, int vote_sgn, int model_stock, float cos_eps, float probability,
                            int rescale, float *thresholds_out, void *workspace, size_t workspace_bytes, uint32_t *flags, void *stream) {
    TiesCall c{};
    c.inputs = deltas; c.n_deltas = n_models; c.keep_bits = keep_bits; c.out = out; c.numel = numel; c.in_dt = in_dtype;
    c.out_dt = out_dtype; c.k = k; c.vote_sgn = vote_sgn; c.probability = probability; c.rescale = rescale;
    c.model_stock = model_stock; c.cos_eps = cos_eps; c.row_len = row_len;
    c.thresholds_out = thresholds_out; c.workspace = workspace; c.workspace_bytes = workspace_bytes; c.flags = flags;
    c.stream = stream;
    return run_ties(c);
}

int mecha_batched_add_difference_ties(const void *base_arena, const void *const *model_arenas, int n_models, void *out_arena,
                                      const mecha_segment *segments, const mecha_segment *segments_host, int n_segments,
                                      int64_t total_elems, int in_dtype, int out_dtype, double k, float alpha, int vote_sgn,
                                      float *thresholds_out, void *workspace, size_t workspace_bytes, uint32_t *seg_flags,
                                      void *stream) {
    MECHA_REQUIRE(base_arena != nullptr, "base arena is NULL");
    MECHA_REQUIRE(segments != nullptr, "device segment table is NULL");
    TiesCall c{};
    c.base = base_arena; c.inputs = model_arenas; c.n_deltas = n_models; c.out = out_arena; c.segs_dev = segments;
    c.segs_host = segments_host; c.n_segs = n_segments; c.total_elems = total_elems; c.in_dt = in_dtype; c.out_dt = out_dtype;
    c.k = k; c.alpha = alpha; c.vote_sgn = vote_sgn; c.thresholds_out = thresholds_out; c.workspace = workspace;
    c.workspace_bytes = workspace_bytes; c.flags = seg_flags; c.stream = stream;
    return run_ties(c);
}

size_t mecha_batched_dare_workspace_bytes(int n_models, int n_segments, int64_t total_elems, int64_t total_rows) {
    if (n_models < 1) n_models = 1;
    if (n_segments < 1) n_segments = 1;
    size_t need = layout(n_models, n_segments, total_elems).total;
    if (total_rows > 0) need += align_up(sizeof(int64_t) * ((size_t)n_segments + 1), 256) + stock_bytes(n_models, total_rows);
    return need;
}

int mecha_batched_add_difference_dare(const void *base_arena, const void *const *model_arenas, int n_models, void *out_arena,
                                      const mecha_segment *segments, const mecha_segment *segments_host, int n_segments,
                                      int64_t total_elems, int in_dtype, int out_dtype, double k, float alpha, int vote_sgn,
                                      float probability, int rescale, const uint32_t *mask_stream, int64_t mask_stream_bits,
                                      int model_stock, float cos_eps, const int64_t *row_lens, const int64_t *row_lens_host,
                                      float *thresholds_out, void *workspace, size_t workspace_bytes, uint32_t *seg_flags, void *stream) {
    MECHA_REQUIRE(base_arena != nullptr, "base arena is NULL");
    MECHA_REQUIRE(segments != nullptr && segments_host != nullptr && n_segments >= 1, "batched DARE needs host and device segment tables");
    MECHA_REQUIRE(mask_stream != nullptr, "mask_stream is NULL (draw it with mecha_dare_draw_stream_host)");
    for (int s = 0; s < n_segments; s++)
        MECHA_REQUIRE((int64_t)n_models * segments_host[s].numel <= mask_stream_bits, "segment %d needs %lld mask bits, the stream holds %lld",
                      s, (long long)n_models * (long long)segments_host[s].numel, (long long)mask_stream_bits);
    TiesCall c{};
    c.base = base_arena; c.inputs = model_arenas; c.n_deltas = n_models; c.out = out_arena; c.segs_dev = segments;
    c.segs_host = segments_host; c.n_segs = n_segments; c.total_elems = total_elems; c.in_dt = in_dtype; c.out_dt = out_dtype;
    c.k = k; c.alpha = alpha; c.vote_sgn = vote_sgn; c.probability = probability; c.rescale = rescale;
    c.keep_bits = mask_stream; c.mask_stream = 1;
    c.model_stock = model_stock; c.cos_eps = cos_eps; c.row_lens_dev = row_lens; c.row_lens_host = row_lens_host;
    c.thresholds_out = thresholds_out; c.workspace = workspace;
    c.workspace_bytes = workspace_bytes; c.flags = seg_flags; c.stream = stream;
    return run_ties(c);
}

size_t mecha_model_stock_workspace_bytes(int n_models, int64_t numel, int64_t row_len) {
    if (n_models < 2) n_models = 2;
    if (row_len < 1) row_len = 1;
    return stock_bytes(n_models, (numel + row_len - 1) / row_len) + 256;
}

int mecha_model_stock(const void *const *deltas, int n_models, void *out, int64_t numel, int64_t row_len, int in_dtype, int out_dtype,
                      float cos_eps, void *workspace, size_t workspace_bytes, uint32_t *flags, void *stream) {
    MECHA_REQUIRE(deltas != nullptr && out != nullptr, "NULL argument");
    MECHA_REQUIRE(n_models >= 2 && n_models <= STOCK_MAX, "model_stock takes 2..%d deltas (got %d): with one, the reference's torch.stack "
                  "of an empty list of cosines raises", STOCK_MAX, n_models);
    MECHA_REQUIRE(mecha_valid_dtype(in_dtype) && mecha_valid_dtype(out_dtype), "unknown dtype code");
    MECHA_REQUIRE(numel >= 0 && row_len >= 1 && numel % row_len == 0, "row_len (%lld) must divide numel (%lld)", (long long)row_len, (long long)numel);
    if (numel == 0) return MECHA_OK;
    const int64_t n_rows = numel / row_len;
    MECHA_REQUIRE(workspace != nullptr && mecha_aligned16(workspace) && workspace_bytes >= stock_bytes(n_models, n_rows),
                  "model_stock workspace too small or unaligned: need %zu bytes, got %zu", stock_bytes(n_models, n_rows), workspace_bytes);
    TiesParams p{};
    bool vec = mecha_aligned16(out);
    for (int m = 0; m < n_models; m++) {
        MECHA_REQUIRE(deltas[m] != nullptr, "input %d is NULL", m);
        p.in[m] = deltas[m];
        vec = vec && mecha_aligned16(deltas[m]);
    }
    p.n_deltas = n_models; p.in_dt = in_dtype; p.out_dt = out_dtype; p.raw = 1; p.vec_ok = vec ? 1 : 0;
    p.seg0 = mecha_segment{0, numel}; p.n_segs = 1;
    p.total_tiles = (numel + TIES_TILE - 1) / TIES_TILE;
    char *ws = static_cast<char *>(workspace);
    StockParams sp{};
    sp.row_len = row_len; sp.n_rows = n_rows; sp.cos_eps = cos_eps;
    sp.sums = reinterpret_cast<double *>(ws);
    sp.t = reinterpret_cast<float *>(ws + align_up(sizeof(double) * (size_t)n_rows * (2 * n_models - 1), 256));
    hipStream_t s = static_cast<hipStream_t>(stream);
    MECHA_CHECK_HIP(hipMemsetAsync(sp.sums, 0, sizeof(double) * (size_t)n_rows * (2 * n_models - 1), s));
    const unsigned sgrid = (unsigned)(p.total_tiles < 4 * MECHA_MAX_GRID ? p.total_tiles : 4 * MECHA_MAX_GRID);
    const int64_t tblocks = (n_rows + MECHA_BLOCK - 1) / MECHA_BLOCK;
    const unsigned tgrid = (unsigned)(tblocks < 65536 ? tblocks : 65536);
    MergeExtra x{};
    MECHA_LAUNCH(MK_TIES_MERGE, (ties_stock_reduce_kernel<false>), dim3(sgrid), dim3(MECHA_BLOCK), 0, s, p, sp, 0, flags);
    MECHA_LAUNCH(MK_TIES_MERGE, ties_stock_t_kernel, dim3(tgrid), dim3(MECHA_BLOCK), 0, s, sp, n_models, (const int64_t *)nullptr);
    MECHA_LAUNCH(MK_TIES_MERGE, (ties_stock_apply_kernel<false>), dim3(sgrid), dim3(MECHA_BLOCK), 0, s, p, sp, x, out, flags);
    MECHA_CHECK_HIP(hipGetLastError());
    return MECHA_OK;
}

int mecha_dropout(const void *const *deltas, int n_models, const uint32_t *keep_bits, void *out, int64_t numel, int in_dtype,
                  int out_dtype, float rescalar, uint32_t *flags, void *stream) {
    MECHA_REQUIRE(deltas != nullptr && out != nullptr && keep_bits != nullptr, "NULL argument");
    MECHA_REQUIRE(n_models >= 1 && n_models <= MECHA_MAX_MODELS, "dropout takes 1..%d deltas (got %d)", MECHA_MAX_MODELS, n_models);
    MECHA_REQUIRE(mecha_valid_dtype(in_dtype) && mecha_valid_dtype(out_dtype), "unknown dtype code");
    MECHA_REQUIRE(numel >= 0, "numel must be >= 0");
    MECHA_REQUIRE(rescalar == rescalar && rescalar != 0.0f, "rescalar must be a non-zero number");
    if (numel == 0) return MECHA_OK;
    DropoutParams p{};
    bool vec = mecha_aligned16(out);
    for (int m = 0; m < n_models; m++) {
        MECHA_REQUIRE(deltas[m] != nullptr, "input %d is NULL", m);
        p.in[m] = deltas[m];
        vec = vec && mecha_aligned16(deltas[m]);
    }
    p.keep_bits = keep_bits; p.mask_words = (numel + 31) / 32; p.n = n_models; p.rescalar = rescalar;
    const int64_t blocks = ((numel + 7) / 8 + MECHA_BLOCK - 1) / MECHA_BLOCK;
    MECHA_REQUIRE(blocks * MECHA_BLOCK < (1ll << 32), "numel %lld is beyond one launch (2^24 - 1 blocks of 2048 elements)", (long long)numel);
    hipStream_t s = static_cast<hipStream_t>(stream);
    MECHA_DISPATCH_DTYPES(in_dtype, out_dtype, {
        MECHA_LAUNCH(MK_STREAMING, (dropout_kernel<IN_DT, OUT_DT>), dim3((unsigned)blocks), dim3(MECHA_BLOCK), 0, s, p, out, numel, vec, flags);
    });
    MECHA_CHECK_HIP(hipGetLastError());
    return MECHA_OK;
}

}  // extern "C"
