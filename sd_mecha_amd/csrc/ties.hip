// ties.hip -- TIES / DARE: exact top-k magnitude threshold by radix select, sign election, disjoint mean.
//
// Reference: extensions/builtin/merge_methods/ties.py:98-132 (ties_sum, ties_sum_deltas, filter_top_k),
// ties.py:10-50 (ties_sum_with_dropout), linear.py:101 (subtract), linear.py:76 (add_difference).
//
// Pipeline (one key, or every key of an arena in the batched form -- same kernels):
//   hist<1> -> scan<1> -> hist<2> -> scan<2> -> hist<3> -> scan<3> -> merge
// The threshold of delta m is the kth-smallest |delta_m| (kth = max(int((1-k)*numel), 1)), found exactly
// by a most-significant-digit radix select on the fp32 bit pattern of |delta| (31 value bits split
// 11 / 11 / 9).  hist<p> counts, per (key, delta), the elements whose already-selected prefix matches, by
// their next digit: LDS atomics into a per-block histogram, flushed to the global per-(key, delta)
// histogram with one atomic per non-empty bin.  scan<p> (one block per (key, delta)) walks the 2048 bins,
// picks the digit that holds rank kth, and zeroes the histogram for the next pass.
// merge recomputes the deltas, applies |delta| >= threshold (ties kept), elects the sign, averages the
// agreeing deltas, and fuses add_difference + output cast + finite checks.
//
// No fp32 delta is ever written to memory.  Roofline: HBM.  Algorithmic bytes per element for the fused
// graph = in_size * (n_models + 1) + out_size (10 B for base + 3 models in bf16); the three
// hist passes re-read the inputs (implementation traffic, reported separately in DESIGN.md).
//
// Storage dtypes are runtime (wave-uniform switches around the 16-byte loads and stores); what is
// compiled in is the register capacity CAP (deltas held per lane: 2, 4 or 8) and the DARE mask path.
#include <type_traits>
#include "common.h"

namespace {

constexpr int TIES_TILE = 4096;            // elements per tile: 256 lanes x 8 elements x 2 vectors
constexpr int TIES_BINS = 2048;
constexpr int TIES_GROUP = 8;              // most deltas handled per hist launch (LDS: 8 x 2048 x 4 B = 64 KiB)
constexpr int TIES_MAX_DELTAS = MECHA_MAX_MODELS;

struct SelectState {                       // per (key, delta)
    uint32_t prefix;                       // digits selected so far, in place
    uint32_t threshold_bits;               // final: fp32 bit pattern of the threshold
    unsigned long long kth_rem;            // rank still to find inside the matching prefix (1-indexed)
};

struct TiesParams {
    const void *base;                      // nullptr: inputs are deltas
    const void *in[TIES_MAX_DELTAS];       // models (base mode) or deltas
    const uint32_t *keep_bits;             // DARE masks or nullptr
    int64_t mask_words;                    // words per delta row of keep_bits
    int n_deltas;
    int in_dt, out_dt;
    int vec_ok;                            // every pointer 16-byte aligned
    const mecha_segment *segs;             // batched: device segment table; else nullptr
    mecha_segment seg0;                    // per key: the one segment
    int n_segs;
    const long long *tile_start;           // batched: device prefix of tiles per segment [n_segs + 1]
    long long total_tiles;
    SelectState *state;                    // [n_segs * n_deltas]
    uint32_t *hist;                        // [n_segs * n_deltas * TIES_BINS]
};

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
        if (dt == MECHA_BF16) load8<MECHA_BF16>(ptr, e, v);
        else if (dt == MECHA_F16) load8<MECHA_F16>(ptr, e, v);
        else load8<MECHA_F32>(ptr, e, v);
    } else {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            float x = 0.0f;
            if (j < valid) x = (dt == MECHA_BF16) ? load1<MECHA_BF16>(ptr, e + j) : (dt == MECHA_F16) ? load1<MECHA_F16>(ptr, e + j) : load1<MECHA_F32>(ptr, e + j);
            v[j] = x;
        }
    }
}
__device__ __forceinline__ void store_rt(int dt, bool vec, void *ptr, int64_t e, int valid, const float (&v)[8]) {
    if (vec) {
        if (dt == MECHA_BF16) store8<MECHA_BF16>(ptr, e, v);
        else if (dt == MECHA_F16) store8<MECHA_F16>(ptr, e, v);
        else store8<MECHA_F32>(ptr, e, v);
    } else {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            if (j < valid) {
                if (dt == MECHA_BF16) store1<MECHA_BF16>(ptr, e + j, v[j]);
                else if (dt == MECHA_F16) store1<MECHA_F16>(ptr, e + j, v[j]);
                else store1<MECHA_F32>(ptr, e + j, v[j]);
            }
        }
    }
}

// DARE: delta * mask (ties.py:33); mask bit i of row m.  `local` = element index inside the key.
__device__ __forceinline__ void apply_keep_bits(const TiesParams &p, int m, int64_t local, float (&v)[8]) {
    const uint32_t word = p.keep_bits[(int64_t)m * p.mask_words + (local >> 5)];
    const uint32_t bits8 = (word >> (local & 31)) & 0xffu;       // local % 8 == 0, so the 8 bits sit in one word
#pragma unroll
    for (int j = 0; j < 8; j++) v[j] = v[j] * (((bits8 >> j) & 1u) ? 1.0f : 0.0f);
}

// ------------------------------------------------------------------------------------------------ hist
template <int PASS> struct PassCfg;
template <> struct PassCfg<1> { static constexpr int shift = 20, bins = 2048; static constexpr uint32_t prefix_mask = 0u; };
template <> struct PassCfg<2> { static constexpr int shift = 9, bins = 2048; static constexpr uint32_t prefix_mask = 0x7ff00000u; };
template <> struct PassCfg<3> { static constexpr int shift = 0, bins = 512; static constexpr uint32_t prefix_mask = 0x7ffffe00u; };

template <int PASS, int CAP, bool DARE>
__global__ __launch_bounds__(MECHA_BLOCK) void ties_hist_kernel(TiesParams p, int d0, int nd) {
    using C = PassCfg<PASS>;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_hist[];     // nd * TIES_BINS words
    for (int i = threadIdx.x; i < nd * TIES_BINS; i += MECHA_BLOCK) lds_hist[i] = 0;
    __syncthreads();

    const long long chunk = (p.total_tiles + gridDim.x - 1) / gridDim.x;
    const long long t0 = (long long)blockIdx.x * chunk;
    const long long t1 = (t0 + chunk < p.total_tiles) ? t0 + chunk : p.total_tiles;
    TileCursor cur;
    int flushed_seg = -1;

    auto flush = [&](int s) {
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
    for (long long t = t0; t < t1; t++) {
        const int prev = cur.s;
        cursor_seek(p, t, cur);
        if (cur.s != prev) {
            if (prev >= 0) flush(prev);
            flushed_seg = prev;
            if constexpr (PASS > 1) {
#pragma unroll
                for (int d = 0; d < CAP; d++) want[d] = (d < nd) ? p.state[(int64_t)cur.s * p.n_deltas + d0 + d].prefix : 0u;
            }
        }
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
                if (d < nd) load_rt(p.in_dt, vec, p.in[d0 + d], e, valid, v[d]);
#pragma unroll
            for (int d = 0; d < CAP; d++) {
                if (d < nd) {
                    if (p.base != nullptr) {
#pragma unroll
                        for (int j = 0; j < 8; j++) v[d][j] = v[d][j] - b[j];              // subtract (linear.py:101)
                    }
                    if constexpr (DARE) apply_keep_bits(p, d0 + d, local, v[d]);
#pragma unroll
                    for (int j = 0; j < 8; j++) {
                        const uint32_t bits = __float_as_uint(v[d][j]) & 0x7fffffffu;        // |delta|
                        bool hit = j < valid;
                        if constexpr (PASS > 1) hit = hit && ((bits & C::prefix_mask) == want[d]);
                        if (hit) atomicAdd(&lds_hist[d * TIES_BINS + ((bits >> C::shift) & (C::bins - 1))], 1u);
                    }
                }
            }
        }
    }
    (void)flushed_seg;
    if (cur.s >= 0) flush(cur.s);
}

// ------------------------------------------------------------------------------------------------ scan
// One block per (key, delta): find the bin holding rank kth_rem, update the state, zero the histogram.
template <int PASS>
__global__ __launch_bounds__(MECHA_BLOCK) void ties_scan_kernel(TiesParams p, double one_minus_k, float *thresholds_out) {
    using C = PassCfg<PASS>;
    __shared__ unsigned long long wave_tot[MECHA_BLOCK / 64];
    __shared__ int sel_bin;
    __shared__ unsigned long long sel_below;
    const int64_t sd = blockIdx.x;                              // s * n_deltas + d
    const int s = (int)(sd / p.n_deltas);
    uint32_t *h = p.hist + sd * TIES_BINS;
    SelectState st = p.state[sd];
    if constexpr (PASS == 1) {
        const mecha_segment seg = (p.segs == nullptr) ? p.seg0 : p.segs[s];
        long long kth = (long long)(one_minus_k * (double)seg.numel);   // max(int((1 - k) * numel), 1)  (ties.py:129)
        if (kth < 1) kth = 1;
        st.prefix = 0; st.threshold_bits = 0; st.kth_rem = (unsigned long long)kth;
    }
    constexpr int PER = TIES_BINS / MECHA_BLOCK;                // 8 bins per thread, contiguous
    uint32_t c[PER];
    unsigned long long local = 0;
#pragma unroll
    for (int j = 0; j < PER; j++) { c[j] = h[threadIdx.x * PER + j]; local += c[j]; }
    // exclusive prefix over threads: wave scan, then wave totals
    unsigned long long incl = local;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        unsigned long long n = __shfl_up(incl, off, 64);
        if (lane >= off) incl += n;
    }
    if (lane == 63) wave_tot[wave] = incl;
    if (threadIdx.x == 0) sel_bin = -1;
    __syncthreads();
    unsigned long long before = incl - local;
    for (int w = 0; w < wave; w++) before += wave_tot[w];
    // exactly one thread's bin range contains rank kth_rem
    if (st.kth_rem > before && st.kth_rem <= before + local) {
        unsigned long long cum = before;
#pragma unroll
        for (int j = 0; j < PER; j++) {
            if (st.kth_rem > cum && st.kth_rem <= cum + c[j]) { sel_bin = threadIdx.x * PER + j; sel_below = cum; }
            cum += c[j];
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < PER; j++) h[threadIdx.x * PER + j] = 0;   // ready for the next pass / the next call
    if (threadIdx.x == 0) {
        int bin = sel_bin;
        if (bin < 0) { bin = 0; sel_below = 0; }                 // unreachable for kth <= numel; keeps the state defined
        st.prefix |= ((uint32_t)bin) << C::shift;
        st.kth_rem -= sel_below;
        if constexpr (PASS == 3) {
            st.threshold_bits = st.prefix;
            if (thresholds_out != nullptr) thresholds_out[sd] = __uint_as_float(st.prefix);
        }
        p.state[sd] = st;
    }
}

// ------------------------------------------------------------------------------------------------ merge
struct MergeExtra {
    float alpha;            // add_difference weight (base mode)
    float rescale_div;      // DARE: divide by (1 - probability), or 1
    int vote_sgn;
    int has_rescale;
};

__device__ __forceinline__ float sign_f(float x) { return (float)((x > 0.0f) - (x < 0.0f)); }   // torch.sign: sign(+-0) = 0

template <int CAP, bool DARE>
__global__ __launch_bounds__(MECHA_BLOCK) void ties_merge_kernel(TiesParams p, MergeExtra x, void *out, uint32_t *flags) {
    const long long chunk = (p.total_tiles + gridDim.x - 1) / gridDim.x;
    const long long t0 = (long long)blockIdx.x * chunk;
    const long long t1 = (t0 + chunk < p.total_tiles) ? t0 + chunk : p.total_tiles;
    const bool in_regs = p.n_deltas <= CAP;        // one sweep over the inputs; otherwise the second sweep re-reads them
    TileCursor cur;
    float thr[CAP];
    for (long long t = t0; t < t1; t++) {
        const int prev = cur.s;
        cursor_seek(p, t, cur);
        const SelectState *st = p.state + (int64_t)cur.s * p.n_deltas;
        if (cur.s != prev && in_regs) {
#pragma unroll
            for (int d = 0; d < CAP; d++) thr[d] = (d < p.n_deltas) ? __uint_as_float(st[d].threshold_bits) : 0.0f;
        }
        const mecha_segment seg = cur.seg;
#pragma unroll 1
        for (int u = 0; u < 2; u++) {
            const int64_t local = (t - cur.first) * TIES_TILE + (int64_t)u * (TIES_TILE / 2) + (int64_t)threadIdx.x * 8;
            if (local >= seg.numel) continue;
            const int valid = (seg.numel - local >= 8) ? 8 : (int)(seg.numel - local);
            const bool vec = p.vec_ok && valid == 8;
            const int64_t e = seg.offset + local;
            float b[8];
            bool bad_in = false;
            if (p.base != nullptr) {
                load_rt(p.in_dt, vec, p.base, e, valid, b);
#pragma unroll
                for (int j = 0; j < 8; j++) bad_in |= nonfinite_f32(b[j]);
            }
            // filtered delta f_m = delta_m * (|delta_m| >= thr_m)   (ties.py:128-132)
            auto filter = [&](int m, float t_m, float (&f)[8]) {
#pragma unroll
                for (int j = 0; j < 8; j++) bad_in |= nonfinite_f32(f[j]);
                if (p.base != nullptr) {
#pragma unroll
                    for (int j = 0; j < 8; j++) f[j] = f[j] - b[j];
                }
                if constexpr (DARE) apply_keep_bits(p, m, local, f);
#pragma unroll
                for (int j = 0; j < 8; j++) f[j] = f[j] * ((__builtin_fabsf(f[j]) >= t_m) ? 1.0f : 0.0f);
            };
            float vote[8], acc[8], cnt[8];
#pragma unroll
            for (int j = 0; j < 8; j++) { vote[j] = 0.0f; acc[j] = 0.0f; cnt[j] = 0.0f; }
            float f[CAP][8];
            // sweep 1: the vote  (ties.py:120: sum of signs, or of deltas if vote_sgn), index order from +0
            for (int g = 0; g < p.n_deltas; g += CAP) {
#pragma unroll
                for (int d = 0; d < CAP; d++)
                    if (g + d < p.n_deltas) load_rt(p.in_dt, vec, p.in[g + d], e, valid, f[d]);
#pragma unroll
                for (int d = 0; d < CAP; d++)
                    if (g + d < p.n_deltas) {
                        filter(g + d, in_regs ? thr[d] : __uint_as_float(st[g + d].threshold_bits), f[d]);
#pragma unroll
                        for (int j = 0; j < 8; j++) vote[j] += x.vote_sgn ? f[d][j] : sign_f(f[d][j]);
                    }
            }
            float fs[8];
#pragma unroll
            for (int j = 0; j < 8; j++) fs[j] = sign_f(vote[j]);
            // sweep 2: disjoint mean over the deltas whose sign agrees (ties.py:122-124, :110)
            for (int g = 0; g < p.n_deltas; g += CAP) {
                if (!in_regs) {
#pragma unroll
                    for (int d = 0; d < CAP; d++)
                        if (g + d < p.n_deltas) {
                            load_rt(p.in_dt, vec, p.in[g + d], e, valid, f[d]);
                            filter(g + d, __uint_as_float(st[g + d].threshold_bits), f[d]);
                        }
                }
#pragma unroll
                for (int d = 0; d < CAP; d++)
                    if (g + d < p.n_deltas) {
#pragma unroll
                        for (int j = 0; j < 8; j++) {
                            const float agree = (sign_f(f[d][j]) == fs[j]) ? 1.0f : 0.0f;
                            acc[j] += f[d][j] * agree;
                            cnt[j] += agree;
                        }
                    }
            }
            float r[8];
            bool bad_out = false;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                float q = acc[j] / cnt[j];
                if (nonfinite_f32(q)) q = 0.0f;                                     // nan_to_num(nan=0, posinf=0, neginf=0)
                if (x.has_rescale) q = q / x.rescale_div;                           // ties.py:50
                if (p.base != nullptr) q = __builtin_fmaf(q, x.alpha, b[j]);        // base.addcmul(ties, alpha)  (linear.py:76)
                r[j] = q;
                bad_out |= (j < valid) && nonfinite_f32(q);
            }
            store_rt(p.out_dt, vec, out, e, valid, r);
            if ((bad_in | bad_out) && flags != nullptr)
                atomicOr(flags + (p.tile_start == nullptr ? 0 : cur.s),
                         (bad_in ? MECHA_FLAG_INPUT_NONFINITE : 0u) | (bad_out ? MECHA_FLAG_OUTPUT_NONFINITE : 0u));
        }
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

// ------------------------------------------------------------------------------------------------ host
size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct WorkspaceLayout {
    size_t tile_start_off, state_off, hist_off, total;
};
WorkspaceLayout layout(int n_deltas, int n_segs) {
    WorkspaceLayout w;
    size_t off = 0;
    w.tile_start_off = off; off = align_up(off + sizeof(long long) * ((size_t)n_segs + 1), 256);
    w.state_off = off;      off = align_up(off + sizeof(SelectState) * (size_t)n_segs * n_deltas, 256);
    w.hist_off = off;       off = align_up(off + sizeof(uint32_t) * (size_t)n_segs * n_deltas * TIES_BINS, 256);
    w.total = off;
    return w;
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
    int in_dt, out_dt;
    double k;
    float alpha;
    int vote_sgn;
    float probability;
    int rescale;
    float *thresholds_out;
    void *workspace;
    size_t workspace_bytes;
    uint32_t *flags;
    void *stream;
};

template <int PASS, bool DARE>
void launch_hist(const TiesParams &p, unsigned grid, hipStream_t s) {
    constexpr int id = PASS == 1 ? MK_TIES_HIST1 : (PASS == 2 ? MK_TIES_HIST2 : MK_TIES_HIST3);
    for (int d0 = 0; d0 < p.n_deltas; d0 += TIES_GROUP) {
        const int nd = (p.n_deltas - d0 < TIES_GROUP) ? p.n_deltas - d0 : TIES_GROUP;
        const size_t lds = (size_t)nd * TIES_BINS * sizeof(uint32_t);
        if (nd <= 2) MECHA_LAUNCH(id, (ties_hist_kernel<PASS, 2, DARE>), dim3(grid), dim3(MECHA_BLOCK), lds, s, p, d0, nd);
        else if (nd <= 4) MECHA_LAUNCH(id, (ties_hist_kernel<PASS, 4, DARE>), dim3(grid), dim3(MECHA_BLOCK), lds, s, p, d0, nd);
        else MECHA_LAUNCH(id, (ties_hist_kernel<PASS, 8, DARE>), dim3(grid), dim3(MECHA_BLOCK), lds, s, p, d0, nd);
    }
}

template <bool DARE>
void launch_all(const TiesParams &p, const MergeExtra &x, unsigned grid, double one_minus_k, float *thresholds_out, void *out,
                uint32_t *flags, hipStream_t s) {
    const unsigned scan_grid = (unsigned)((int64_t)p.n_segs * p.n_deltas);
    launch_hist<1, DARE>(p, grid, s);
    MECHA_LAUNCH(MK_TIES_SCAN, (ties_scan_kernel<1>), dim3(scan_grid), dim3(MECHA_BLOCK), 0, s, p, one_minus_k, thresholds_out);
    launch_hist<2, DARE>(p, grid, s);
    MECHA_LAUNCH(MK_TIES_SCAN, (ties_scan_kernel<2>), dim3(scan_grid), dim3(MECHA_BLOCK), 0, s, p, one_minus_k, thresholds_out);
    launch_hist<3, DARE>(p, grid, s);
    MECHA_LAUNCH(MK_TIES_SCAN, (ties_scan_kernel<3>), dim3(scan_grid), dim3(MECHA_BLOCK), 0, s, p, one_minus_k, thresholds_out);
    if (p.n_deltas <= 2) MECHA_LAUNCH(MK_TIES_MERGE, (ties_merge_kernel<2, DARE>), dim3(grid), dim3(MECHA_BLOCK), 0, s, p, x, out, flags);
    else if (p.n_deltas <= 4) MECHA_LAUNCH(MK_TIES_MERGE, (ties_merge_kernel<4, DARE>), dim3(grid), dim3(MECHA_BLOCK), 0, s, p, x, out, flags);
    else MECHA_LAUNCH(MK_TIES_MERGE, (ties_merge_kernel<8, DARE>), dim3(grid), dim3(MECHA_BLOCK), 0, s, p, x, out, flags);
}

int run_ties(const TiesCall &c) {
    if (c.segs_dev == nullptr && c.numel == 0 && c.n_deltas >= 1) return MECHA_OK;   // ties.py:106-107: empty tensor, nothing to do
    MECHA_REQUIRE(c.out != nullptr, "out is NULL");
    MECHA_REQUIRE(c.inputs != nullptr && c.n_deltas >= 1 && c.n_deltas <= TIES_MAX_DELTAS - (c.base ? 1 : 0),
                  "TIES needs 1..%d deltas (got %d)", TIES_MAX_DELTAS - (c.base ? 1 : 0), c.n_deltas);
    MECHA_REQUIRE(mecha_valid_dtype(c.in_dt) && mecha_valid_dtype(c.out_dt), "unknown dtype code (in %d, out %d)", c.in_dt, c.out_dt);
    MECHA_REQUIRE(c.k >= 0.0 && c.k <= 1.0, "k must be in [0, 1] (got %g): torch.kthvalue would raise", c.k);
    const bool batched = c.segs_dev != nullptr;
    long long total_tiles = 0;
    if (batched) {
        MECHA_REQUIRE(c.segs_host != nullptr && c.n_segs >= 1, "batched TIES needs host and device segment tables");
        int64_t prev_end = 0;
        for (int s = 0; s < c.n_segs; s++) {
            MECHA_REQUIRE(c.segs_host[s].numel > 0 && (c.segs_host[s].offset % 8) == 0 && c.segs_host[s].offset >= prev_end,
                          "segment %d: offsets must be multiples of 8, sorted and non-overlapping, numel > 0", s);
            prev_end = c.segs_host[s].offset + c.segs_host[s].numel;
            total_tiles += (c.segs_host[s].numel + TIES_TILE - 1) / TIES_TILE;
        }
    } else {
        MECHA_REQUIRE(c.numel >= 0, "numel must be >= 0");
        total_tiles = (c.numel + TIES_TILE - 1) / TIES_TILE;
    }
    const int n_segs = batched ? c.n_segs : 1;
    MECHA_REQUIRE(!(c.base != nullptr && mecha_isclose(c.alpha, 0.0)),
                  "alpha == 0: add_difference returns `a` untouched (linear.py:72-73); call the cast entry point on base instead");
    const WorkspaceLayout w = layout(c.n_deltas, n_segs);
    MECHA_REQUIRE(c.workspace != nullptr && c.workspace_bytes >= w.total && mecha_aligned16(c.workspace),
                  "TIES workspace too small or unaligned: need %zu bytes, got %zu", w.total, c.workspace_bytes);
    if (c.keep_bits != nullptr) MECHA_REQUIRE(!batched && c.base == nullptr, "DARE masks are supported on the per-key delta form only");

    bool vec = mecha_aligned16(c.out) && (c.base == nullptr || mecha_aligned16(c.base));
    for (int m = 0; m < c.n_deltas; m++) {
        MECHA_REQUIRE(c.inputs[m] != nullptr, "input %d is NULL", m);
        vec = vec && mecha_aligned16(c.inputs[m]);
    }
    hipStream_t s = static_cast<hipStream_t>(c.stream);
    char *ws = static_cast<char *>(c.workspace);
    TiesParams p{};
    p.base = c.base;
    for (int m = 0; m < c.n_deltas; m++) p.in[m] = c.inputs[m];
    p.keep_bits = c.keep_bits;
    p.mask_words = batched ? 0 : (c.numel + 31) / 32;
    p.n_deltas = c.n_deltas;
    p.in_dt = c.in_dt;
    p.out_dt = c.out_dt;
    p.vec_ok = vec ? 1 : 0;
    p.segs = batched ? c.segs_dev : nullptr;
    p.seg0 = mecha_segment{0, batched ? 0 : c.numel};
    p.n_segs = n_segs;
    p.tile_start = batched ? reinterpret_cast<const long long *>(ws + w.tile_start_off) : nullptr;
    p.total_tiles = total_tiles;
    p.state = reinterpret_cast<SelectState *>(ws + w.state_off);
    p.hist = reinterpret_cast<uint32_t *>(ws + w.hist_off);

    MECHA_CHECK_HIP(hipMemsetAsync(p.hist, 0, sizeof(uint32_t) * (size_t)n_segs * c.n_deltas * TIES_BINS, s));
    if (batched)
        MECHA_LAUNCH(MK_TIES_INDEX, ties_tile_index_kernel, dim3(1), dim3(1024), 0, s, c.segs_dev, n_segs,
                     reinterpret_cast<long long *>(ws + w.tile_start_off));
    unsigned grid = (unsigned)(total_tiles < MECHA_MAX_GRID ? total_tiles : MECHA_MAX_GRID);
    const double one_minus_k = 1.0 - c.k;
    const bool dare = c.keep_bits != nullptr;
    MergeExtra x{};
    x.alpha = c.alpha;
    x.vote_sgn = c.vote_sgn;
    x.has_rescale = dare && c.rescale;
    x.rescale_div = 1.0f - c.probability;                 // 1.0 - probability as fp32 tensor arithmetic (ties.py:49)
    if (dare) launch_all<true>(p, x, grid, one_minus_k, c.thresholds_out, c.out, c.flags, s);
    else launch_all<false>(p, x, grid, one_minus_k, c.thresholds_out, c.out, c.flags, s);
    MECHA_CHECK_HIP(hipGetLastError());
    return MECHA_OK;
}

}  // namespace

extern "C" {

size_t mecha_ties_workspace_bytes(int n_models, int64_t numel) {
    (void)numel;
    if (n_models < 1) n_models = 1;
    return layout(n_models, 1).total;
}

size_t mecha_batched_ties_workspace_bytes(int n_models, int n_segments, int64_t total_elems) {
    (void)total_elems;
    if (n_models < 1) n_models = 1;
    if (n_segments < 1) n_segments = 1;
    return layout(n_models, n_segments).total;
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

int mecha_batched_add_difference_ties(const void *base_arena, const void *const *model_arenas, int n_models, void *out_arena,
                                      const mecha_segment *segments, const mecha_segment *segments_host, int n_segments,
                                      int64_t total_elems, int in_dtype, int out_dtype, double k, float alpha, int vote_sgn,
                                      float *thresholds_out, void *workspace, size_t workspace_bytes, uint32_t *seg_flags,
                                      void *stream) {
    (void)total_elems;
    MECHA_REQUIRE(base_arena != nullptr, "base arena is NULL");
    MECHA_REQUIRE(segments != nullptr, "device segment table is NULL");
    TiesCall c{};
    c.base = base_arena; c.inputs = model_arenas; c.n_deltas = n_models; c.out = out_arena; c.segs_dev = segments;
    c.segs_host = segments_host; c.n_segs = n_segments; c.in_dt = in_dtype; c.out_dt = out_dtype; c.k = k; c.alpha = alpha;
    c.vote_sgn = vote_sgn; c.thresholds_out = thresholds_out; c.workspace = workspace; c.workspace_bytes = workspace_bytes;
    c.flags = seg_flags; c.stream = stream;
    return run_ties(c);
}

}  // extern "C"
