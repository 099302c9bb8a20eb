// abi.cpp -- version, per-thread error string and the host-only entry points of libmecha_hip.so.
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include <mutex>
#include <vector>

#include "common.h"

static thread_local char g_last_error[512] = "";

// ---------------------------------------------------------------- per-kernel timers (off by default)
bool g_mecha_profiling = false;
namespace {
struct ProfRecord { hipEvent_t start, stop; int id; };
std::mutex g_prof_mutex;
std::vector<ProfRecord> g_prof_open;                 // begin recorded, end pending (one per launching thread at a time)
std::vector<ProfRecord> g_prof_done;
double g_prof_ms[MK_COUNT];
long long g_prof_launches[MK_COUNT];
const char *const g_prof_names[MK_COUNT] = {
    "streaming_ew", "ties_tile_index", "ties_hist1", "ties_hist2", "ties_hist3", "ties_scan", "ties_merge",
    "reduce_partial", "reduce_finish", "reduce_apply", "reduce_fixup", "lora_mfma", "rotate_gemm_mfma", "rotate_gram_norms",
    "ties_sample", "ties_count", "ties_resolve", "ties_fixup"};
thread_local ProfRecord t_current;
}  // namespace

void mecha_prof_begin(hipStream_t s, int id) {
    ProfRecord r{};
    r.id = id;
    if (hipEventCreate(&r.start) != hipSuccess || hipEventCreate(&r.stop) != hipSuccess) return;
    (void)hipEventRecord(r.start, s);
    t_current = r;
}

void mecha_prof_end(hipStream_t s, int id) {
    ProfRecord r = t_current;
    if (r.id != id || r.start == nullptr) return;
    (void)hipEventRecord(r.stop, s);
    std::lock_guard<std::mutex> lock(g_prof_mutex);
    g_prof_done.push_back(r);
    t_current = ProfRecord{};
}

void mecha_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_last_error, sizeof(g_last_error), fmt, ap);
    va_end(ap);
}

extern "C" {

int mecha_abi_version(void) { return MECHA_ABI_VERSION; }

int mecha_profile_enable(int on) {
    std::lock_guard<std::mutex> lock(g_prof_mutex);
    g_mecha_profiling = on != 0;
    return MECHA_OK;
}

int mecha_profile_kernel_count(void) { return MK_COUNT; }

const char *mecha_profile_kernel_name(int id) { return (id >= 0 && id < MK_COUNT) ? g_prof_names[id] : ""; }

// Waits for every recorded launch, folds its elapsed time into the per-kernel totals, returns them.
int mecha_profile_read(int id, double *total_ms, long long *launches, int reset) {
    if (id < 0 || id >= MK_COUNT) {
        mecha_set_error("mecha_profile_read: kernel id %d out of range", id);
        return MECHA_ERR_ARG;
    }
    std::lock_guard<std::mutex> lock(g_prof_mutex);
    for (auto &r : g_prof_done) {
        float ms = 0.0f;
        if (hipEventSynchronize(r.stop) == hipSuccess && hipEventElapsedTime(&ms, r.start, r.stop) == hipSuccess) {
            g_prof_ms[r.id] += ms;
            g_prof_launches[r.id] += 1;
        }
        (void)hipEventDestroy(r.start);
        (void)hipEventDestroy(r.stop);
    }
    g_prof_done.clear();
    if (total_ms) *total_ms = g_prof_ms[id];
    if (launches) *launches = g_prof_launches[id];
    if (reset) { g_prof_ms[id] = 0.0; g_prof_launches[id] = 0; }
    return MECHA_OK;
}

const char *mecha_last_error(void) { return g_last_error; }

// torch's CPU generator (at::mt19937, seeded with the low 32 bits of the seed) and its tensor-p
// Bernoulli kernel: one 32-bit draw per element, u = (x & (2^24 - 1)) * 2^-24, keep = u < p,
// serial in element order, deltas in argument order (reference ties.py:29-33, :63-64).
// mecha_dare_draw_stream_host writes the keep bits of the first n_bits draws, bit (i & 31) of word (i >> 5) = draw i.
// The generator is sequential, but nothing else is: the state is advanced 624 words at a time by three loops without a
// loop-carried dependence shorter than 227 words, and tempering, compare and bit packing work on whole blocks, so the
// compiler vectorises all of it (one thread: ~1 ns per draw before, a few tenths now).
}  // extern "C"  (helpers follow; the exports resume below)

namespace {

struct Mt19937 {
    uint32_t mt[624 + 1];
    explicit Mt19937(uint32_t seed) {
        mt[0] = seed;
        for (int i = 1; i < 624; i++) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
    }
    static inline uint32_t mix(uint32_t hi, uint32_t lo, uint32_t far) {
        const uint32_t y = (hi & 0x80000000u) | (lo & 0x7fffffffu);
        return far ^ (y >> 1) ^ ((0u - (y & 1u)) & 0x9908b0dfu);
    }
    void twist() {
        for (int j = 0; j < 227; j++) mt[j] = mix(mt[j], mt[j + 1], mt[j + 397]);
        for (int j = 227; j < 623; j++) mt[j] = mix(mt[j], mt[j + 1], mt[j - 227]);
        mt[623] = mix(mt[623], mt[0], mt[396]);
    }
    // keep bits of the next 624 draws: keep[i] = 1 iff u_i < p
    void block(float p24, uint8_t *keep) {
        twist();
        for (int i = 0; i < 624; i++) {
            uint32_t y = mt[i];
            y ^= y >> 11;
            y ^= (y << 7) & 0x9d2c5680u;
            y ^= (y << 15) & 0xefc60000u;
            y ^= y >> 18;
            keep[i] = (float)(y & 0xffffffu) < p24 ? 1 : 0;       // (y & mask) * 2^-24 < p, both sides scaled by 2^24 (exact)
        }
    }
};

void draw_stream(uint64_t seed, float probability, int64_t n_bits, uint32_t *out) {
    float p = 1.0f - probability;                    // ones_like(delta) - probability, then clamp(0, 1)
    if (p < 0.0f) p = 0.0f;
    if (p > 1.0f) p = 1.0f;
    const float p24 = p * 16777216.0f;
    Mt19937 gen((uint32_t)seed);
    const int64_t words = (n_bits + 31) / 32;
    uint8_t keep[1248];                              // two generator blocks = 39 whole words
    for (int64_t w0 = 0; w0 < words; w0 += 39) {
        gen.block(p24, keep);
        gen.block(p24, keep + 624);
        const int64_t nw = words - w0 < 39 ? words - w0 : 39;
        for (int64_t w = 0; w < nw; w++) {
            uint32_t word = 0;
            for (int b = 0; b < 32; b++) word |= (uint32_t)keep[w * 32 + b] << b;
            out[w0 + w] = word;
        }
    }
    if (n_bits & 31) out[words - 1] &= (1u << (n_bits & 31)) - 1u;      // bits past the end stay clear
}

}  // namespace

extern "C" {

int mecha_dare_draw_stream_host(uint64_t seed, float probability, int64_t n_bits, uint32_t *bits_out) {
    if (bits_out == nullptr || n_bits < 0) {
        mecha_set_error("mecha_dare_draw_stream_host: bad arguments");
        return MECHA_ERR_ARG;
    }
    draw_stream(seed, probability, n_bits, bits_out);
    bits_out[(n_bits + 31) / 32] = 0u;               // the pad word the kernels may read behind the last bits
    return MECHA_OK;
}

// The same draws cut into one row per delta (rows start at word boundaries): row m = draws [m numel, (m + 1) numel).
int mecha_dare_draw_masks_host(uint64_t seed, float probability, int n_models, int64_t numel, uint32_t *bits_out) {
    if (bits_out == nullptr || n_models < 0 || numel < 0) {
        mecha_set_error("mecha_dare_draw_masks_host: bad arguments");
        return MECHA_ERR_ARG;
    }
    const int64_t words = (numel + 31) / 32;
    if (n_models == 0 || numel == 0) return MECHA_OK;
    if ((numel & 31) == 0 || n_models == 1) {        // rows are word aligned inside the stream too
        draw_stream(seed, probability, (int64_t)n_models * numel, bits_out);
        return MECHA_OK;
    }
    const int64_t total = (int64_t)n_models * numel;
    std::vector<uint32_t> stream((size_t)((total + 31) / 32) + 1, 0u);
    draw_stream(seed, probability, total, stream.data());
    for (int m = 0; m < n_models; m++) {
        uint32_t *row = bits_out + (int64_t)m * words;
        const int64_t s0 = (int64_t)m * numel;
        const uint32_t sh = (uint32_t)(s0 & 31);
        const uint32_t *src = stream.data() + (s0 >> 5);
        for (int64_t w = 0; w < words; w++) row[w] = sh ? ((src[w] >> sh) | (src[w + 1] << (32u - sh))) : src[w];
        if (numel & 31) row[words - 1] &= (1u << (numel & 31)) - 1u;
    }
    return MECHA_OK;
}

}  // extern "C"
