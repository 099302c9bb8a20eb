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
    "reduce_partial", "reduce_finish", "reduce_apply", "reduce_fixup", "lora_mfma", "rotate_gemm_f32mfma", "rotate_gram_norms",
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
int mecha_dare_draw_masks_host(uint64_t seed, float probability, int n_models, int64_t numel, uint32_t *bits_out) {
    if (bits_out == nullptr || n_models < 0 || numel < 0) {
        mecha_set_error("mecha_dare_draw_masks_host: bad arguments");
        return MECHA_ERR_ARG;
    }
    uint32_t mt[624];
    mt[0] = (uint32_t)seed;
    for (int i = 1; i < 624; i++) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
    int idx = 624;
    float p = 1.0f - probability;                    // ones_like(delta) - probability, then clamp(0, 1)
    if (p < 0.0f) p = 0.0f;
    if (p > 1.0f) p = 1.0f;
    const int64_t words = (numel + 31) / 32;
    for (int m = 0; m < n_models; m++) {
        uint32_t *row = bits_out + (int64_t)m * words;
        for (int64_t w = 0; w < words; w++) row[w] = 0;
        for (int64_t i = 0; i < numel; i++) {
            if (idx >= 624) {
                for (int j = 0; j < 624; j++) {
                    uint32_t y = (mt[j] & 0x80000000u) | (mt[(j + 1) % 624] & 0x7fffffffu);
                    mt[j] = mt[(j + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
                }
                idx = 0;
            }
            uint32_t y = mt[idx++];
            y ^= y >> 11;
            y ^= (y << 7) & 0x9d2c5680u;
            y ^= (y << 15) & 0xefc60000u;
            y ^= y >> 18;
            const float u = (float)(y & 0xffffffu) * (1.0f / 16777216.0f);
            if (u < p) row[i >> 5] |= 1u << (i & 31);
        }
    }
    return MECHA_OK;
}

}  // extern "C"
