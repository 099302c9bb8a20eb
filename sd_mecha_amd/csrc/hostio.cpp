// hostio.cpp -- host side of the key scheduler's file traffic: many-piece pread / pwrite on a small thread team.
//
// Replaces, for the batched path, the reference's per-key `readinto` window + clone (streaming.py:151-178, one lock per
// file) and its per-thread write buffer flushed under one lock (streaming.py:277-308): the scheduler hands over the whole
// scatter list of a wave (file offset -> address inside a pinned staging arena, or the reverse) and the pieces are moved
// by up to `threads` workers straight between the page cache and pinned memory.  No torch, no HIP: plain POSIX.
#include <errno.h>
#include <immintrin.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <thread>
#include <vector>

#include "common.h"

namespace {

constexpr size_t PIECE = (size_t)4 << 20;          // work unit: 4 MiB, so that one 100 MB tensor spreads over the team

struct Piece { int64_t entry; size_t skip, len; };

std::vector<Piece> split(const size_t *nbytes, int64_t n) {
    std::vector<Piece> pieces;
    for (int64_t i = 0; i < n; i++)
        for (size_t done = 0; done < nbytes[i]; done += PIECE)
            pieces.push_back(Piece{i, done, nbytes[i] - done < PIECE ? nbytes[i] - done : PIECE});
    return pieces;
}

template <typename F>
int run_team(const std::vector<Piece> &pieces, int threads, F &&one) {
    if (threads < 1) threads = 1;
    if ((size_t)threads > pieces.size()) threads = (int)(pieces.size() ? pieces.size() : 1);
    std::atomic<size_t> next{0};
    std::atomic<int> err{0};
    auto worker = [&]() {
        for (;;) {
            const size_t i = next.fetch_add(1, std::memory_order_relaxed);
            if (i >= pieces.size() || err.load(std::memory_order_relaxed) != 0) return;
            const int e = one(pieces[i]);
            if (e != 0) err.store(e, std::memory_order_relaxed);
        }
    };
    std::vector<std::thread> team;
    for (int t = 1; t < threads; t++) team.emplace_back(worker);
    worker();
    for (auto &t : team) t.join();
    return err.load();
}

// memcpy whose stores bypass the caches: the output pages are written once and not read again by this process, and an
// ordinary store to a line that is not cached first READS the line (6.9 GB of SDXL output: 6.9 GB of extra memory traffic
// beside the staging team's preads).  glibc's memcpy switches to such stores only above ~3/4 of the shared cache per copy;
// the team's pieces are 4 MiB.  Measured end to end on the GPU box (16 vCPUs): SDXL x4 TIES 0.75-0.82 s against 0.76-0.78 with
// memcpy, the LoRA job 0.83-0.85 against 0.86-0.88: the first touch of every 4-KiB page of the mapping, not the copy, is what
// the writer spends its time on, and the runs with these stores spread wider (up to 0.88 s).  Off unless MECHA_HOST_NT=1.
__attribute__((target("avx2"))) void copy_nt_avx2(char *dst, const char *src, size_t n) {
    const size_t head = (32 - (reinterpret_cast<uintptr_t>(dst) & 31)) & 31;
    if (head >= n) { memcpy(dst, src, n); return; }
    memcpy(dst, src, head);
    dst += head; src += head; n -= head;
    size_t i = 0;
    for (; i + 128 <= n; i += 128) {
        const __m256i a = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(src + i));
        const __m256i b = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(src + i + 32));
        const __m256i c = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(src + i + 64));
        const __m256i d = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(src + i + 96));
        _mm256_stream_si256(reinterpret_cast<__m256i *>(dst + i), a);
        _mm256_stream_si256(reinterpret_cast<__m256i *>(dst + i + 32), b);
        _mm256_stream_si256(reinterpret_cast<__m256i *>(dst + i + 64), c);
        _mm256_stream_si256(reinterpret_cast<__m256i *>(dst + i + 96), d);
    }
    _mm_sfence();
    memcpy(dst + i, src + i, n - i);
}
void copy_out(char *dst, const char *src, size_t n) {
    static const bool nt = []() {
        const char *e = getenv("MECHA_HOST_NT");
        return e != nullptr && e[0] == '1' && __builtin_cpu_supports("avx2");
    }();
    if (nt && n >= 4096) copy_nt_avx2(dst, src, n);
    else memcpy(dst, src, n);
}

}  // namespace

extern "C" {

int mecha_host_pread_many(const int *fds, void *const *dst, const size_t *nbytes, const int64_t *offsets, int64_t n, int threads) {
    MECHA_REQUIRE(n >= 0 && (n == 0 || (fds != nullptr && dst != nullptr && nbytes != nullptr && offsets != nullptr)), "NULL argument");
    const std::vector<Piece> pieces = split(nbytes, n);
    const int e = run_team(pieces, threads, [&](const Piece &p) -> int {
        char *out = static_cast<char *>(dst[p.entry]) + p.skip;
        if (fds[p.entry] < 0) { memset(out, 0, p.len); return 0; }           // a gap of the arena: keep it finite
        size_t done = 0;
        while (done < p.len) {
            const ssize_t got = pread(fds[p.entry], out + done, p.len - done, (off_t)(offsets[p.entry] + (int64_t)(p.skip + done)));
            if (got < 0) { if (errno == EINTR) continue; return errno; }
            if (got == 0) return EIO;                                           // the file is shorter than its header says
            done += (size_t)got;
        }
        return 0;
    });
    MECHA_REQUIRE(e == 0, "pread failed: %s", strerror(e));
    return MECHA_OK;
}

int mecha_host_pwrite_many(int fd, const void *const *src, const size_t *nbytes, const int64_t *offsets, int64_t n, int threads) {
    MECHA_REQUIRE(fd >= 0 && n >= 0 && (n == 0 || (src != nullptr && nbytes != nullptr && offsets != nullptr)), "bad argument");
    const std::vector<Piece> pieces = split(nbytes, n);
    const int e = run_team(pieces, threads, [&](const Piece &p) -> int {
        const char *in = static_cast<const char *>(src[p.entry]) + p.skip;
        size_t done = 0;
        while (done < p.len) {
            const ssize_t put = pwrite(fd, in + done, p.len - done, (off_t)(offsets[p.entry] + (int64_t)(p.skip + done)));
            if (put < 0) { if (errno == EINTR) continue; return errno; }
            done += (size_t)put;
        }
        return 0;
    });
    MECHA_REQUIRE(e == 0, "pwrite failed: %s", strerror(e));
    return MECHA_OK;
}

// The same scatter list through a shared mapping of the output file.  Buffered pwrite()s to ONE file serialise on its
// inode lock whatever the number of threads; stores into a MAP_SHARED mapping fault pages in independently, so the team
// scales.  The file must already be at least as long as the furthest piece (the writer ftruncates it at creation).
int mecha_host_mmap_write_many(int fd, const void *const *src, const size_t *nbytes, const int64_t *offsets, int64_t n, int threads) {
    MECHA_REQUIRE(fd >= 0 && n >= 0 && (n == 0 || (src != nullptr && nbytes != nullptr && offsets != nullptr)), "bad argument");
    int64_t lo = INT64_MAX, hi = 0;
    for (int64_t i = 0; i < n; i++) {
        if (nbytes[i] == 0) continue;
        MECHA_REQUIRE(offsets[i] >= 0, "negative file offset");
        if (offsets[i] < lo) lo = offsets[i];
        if (offsets[i] + (int64_t)nbytes[i] > hi) hi = offsets[i] + (int64_t)nbytes[i];
    }
    if (hi == 0) return MECHA_OK;
    struct stat st;
    MECHA_REQUIRE(fstat(fd, &st) == 0, "fstat failed: %s", strerror(errno));
    MECHA_REQUIRE((int64_t)st.st_size >= hi, "output file holds %lld bytes, the scatter list reaches %lld: ftruncate it first",
                  (long long)st.st_size, (long long)hi);
    const int64_t page = sysconf(_SC_PAGESIZE);
    const int64_t map_lo = lo / page * page;
    void *base = mmap(nullptr, (size_t)(hi - map_lo), PROT_READ | PROT_WRITE, MAP_SHARED, fd, (off_t)map_lo);
    MECHA_REQUIRE(base != MAP_FAILED, "mmap of the output file failed: %s", strerror(errno));
    const std::vector<Piece> pieces = split(nbytes, n);
    const int e = run_team(pieces, threads, [&](const Piece &p) -> int {
        copy_out(static_cast<char *>(base) + (offsets[p.entry] - map_lo) + p.skip, static_cast<const char *>(src[p.entry]) + p.skip, p.len);
        return 0;
    });
    MECHA_REQUIRE(munmap(base, (size_t)(hi - map_lo)) == 0, "munmap failed: %s", strerror(errno));
    MECHA_REQUIRE(e == 0, "store into the output mapping failed: %s", strerror(e));
    return MECHA_OK;
}

// n host-to-device copies enqueued on one stream (the "inputs already in pinned RAM" mode of the scheduler: every (key, model)
// pair goes from its place in a pinned checkpoint image to its arena offset; a python loop over 10 000 copies costs more than
// the copies).  `src` must be pinned for the copies to be asynchronous.
int mecha_host_h2d_many(void *const *dst_dev, const void *const *src_host, const size_t *nbytes, int64_t n, void *stream) {
    MECHA_REQUIRE(n >= 0 && (n == 0 || (dst_dev != nullptr && src_host != nullptr && nbytes != nullptr)), "NULL argument");
    hipStream_t s = static_cast<hipStream_t>(stream);
    for (int64_t i = 0; i < n; i++) {
        if (nbytes[i] == 0) continue;
        MECHA_CHECK_HIP(hipMemcpyAsync(dst_dev[i], src_host[i], nbytes[i], hipMemcpyHostToDevice, s));
    }
    return MECHA_OK;
}

// The header of a .safetensors file (reference streaming.py:141-149 reads it with json.loads: 10-17 ms of interpreter time per
// LoRA file with ~2 400 entries, and a LoRA job opens 17 files) as flat tables.  `json`: the header text, `len` bytes.  An entry is
// "name": {"dtype": "F16", "shape": [..], "data_offsets": [begin, end]} in any member order; "__metadata__" (string -> string) is
// returned as a span for the caller's own JSON reader.  Entries come out ordered by data offset (stable), as the reference orders
// them.  Per entry i: its name, NUL-terminated, appended to `names` (room for `len` bytes; *names_bytes = bytes used: one decode and
// one split on the caller's side instead of a string operation per entry), dtype_code[i] = index into
// F64 I64 F32 I32 F16 BF16 I16 I8 U8 BOOL, ndim[i] <= 8, shape[8 i ..], offsets[2 i], [2 i + 1].
// Returns the number of entries, or -1 for anything it does not read (another dtype tag, more than 8 dimensions, more than
// max_entries, a name with a JSON escape or a NUL, malformed text): the caller then falls back to its JSON reader.
namespace {
struct HeaderScan {
    const char *p, *end;
    void ws() { while (p < end && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) p++; }
    bool eat(char c) { ws(); if (p < end && *p == c) { p++; return true; } return false; }
    // a string: p at the opening quote; span = the text between the quotes
    bool str(const char *&b, int64_t &n, bool &escaped) {
        ws();
        if (p >= end || *p != '"') return false;
        b = ++p; escaped = false;
        while (p < end && *p != '"') { if (*p == '\\') { escaped = true; p++; } p++; }
        if (p >= end) return false;
        n = p - b; p++;
        return true;
    }
    bool integer(int64_t &v) {
        ws();
        if (p >= end || *p < '0' || *p > '9') return false;
        v = 0;
        while (p < end && *p >= '0' && *p <= '9') { v = v * 10 + (*p - '0'); p++; }
        return true;
    }
    // any JSON value, skipped
    bool skip() {
        ws();
        if (p >= end) return false;
        if (*p == '"') { const char *b; int64_t n; bool e; return str(b, n, e); }
        if (*p == '{' || *p == '[') {
            const char close = *p == '{' ? '}' : ']';
            p++;
            if (eat(close)) return true;
            for (;;) {
                if (close == '}') { const char *b; int64_t n; bool e; if (!str(b, n, e) || !eat(':')) return false; }
                if (!skip()) return false;
                if (eat(',')) continue;
                return eat(close);
            }
        }
        while (p < end && *p != ',' && *p != '}' && *p != ']' && *p != ' ' && *p != '\n') p++;      // number / true / false / null
        return true;
    }
};
bool span_is(const char *b, int64_t n, const char *lit) { return (int64_t)strlen(lit) == n && memcmp(b, lit, (size_t)n) == 0; }
}  // namespace

int64_t mecha_host_parse_safetensors_header(const char *json, int64_t len, int64_t max_entries, char *names, int64_t *names_bytes,
                                            int32_t *dtype_code, int32_t *ndim, int64_t *shape, int64_t *offsets, int64_t *metadata_span) {
    if (json == nullptr || len <= 0 || names == nullptr || names_bytes == nullptr || dtype_code == nullptr || ndim == nullptr ||
        shape == nullptr || offsets == nullptr || metadata_span == nullptr) return -1;
    struct Entry { const char *name; int64_t name_len; int32_t code, nd; int64_t shape[8]; int64_t off[2]; };
    std::vector<Entry> found;
    static const char *const tags[] = {"F64", "I64", "F32", "I32", "F16", "BF16", "I16", "I8", "U8", "BOOL"};
    HeaderScan s{json, json + len};
    metadata_span[0] = 0; metadata_span[1] = 0;
    if (!s.eat('{')) return -1;
    int64_t n = 0;
    if (s.eat('}')) return 0;
    for (;;) {
        const char *kb; int64_t kn; bool kesc;
        if (!s.str(kb, kn, kesc) || !s.eat(':')) return -1;
        if (!kesc && span_is(kb, kn, "__metadata__")) {
            s.ws();
            const char *v0 = s.p;
            if (!s.skip()) return -1;
            metadata_span[0] = v0 - json; metadata_span[1] = s.p - v0;
        } else {
            if (n >= max_entries || kesc || memchr(kb, 0, (size_t)kn) != nullptr || !s.eat('{')) return -1;
            int code = -1, nd = -1;
            bool have_off = false;
            Entry cur{};
            if (!s.eat('}')) {
                for (;;) {
                    const char *mb; int64_t mn; bool mesc;
                    if (!s.str(mb, mn, mesc) || !s.eat(':')) return -1;
                    if (span_is(mb, mn, "dtype")) {
                        const char *tb; int64_t tn; bool tesc;
                        if (!s.str(tb, tn, tesc)) return -1;
                        for (int t = 0; t < 10; t++) if (span_is(tb, tn, tags[t])) code = t;
                        if (code < 0) return -1;
                    } else if (span_is(mb, mn, "shape")) {
                        if (!s.eat('[')) return -1;
                        nd = 0;
                        if (!s.eat(']')) {
                            for (;;) {
                                int64_t v;
                                if (nd >= 8 || !s.integer(v)) return -1;
                                cur.shape[nd++] = v;
                                if (s.eat(',')) continue;
                                if (!s.eat(']')) return -1;
                                break;
                            }
                        }
                    } else if (span_is(mb, mn, "data_offsets")) {
                        if (!s.eat('[') || !s.integer(cur.off[0]) || !s.eat(',') || !s.integer(cur.off[1]) || !s.eat(']')) return -1;
                        have_off = true;
                    } else if (!s.skip()) return -1;
                    if (s.eat(',')) continue;
                    if (!s.eat('}')) return -1;
                    break;
                }
            }
            if (code < 0 || nd < 0 || !have_off) return -1;
            cur.name = kb; cur.name_len = kn; cur.code = code; cur.nd = nd;
            found.push_back(cur);
            n++;
        }
        if (s.eat(',')) continue;
        if (!s.eat('}')) return -1;
        break;
    }
    s.ws();
    if (s.p != s.end) return -1;
    std::stable_sort(found.begin(), found.end(), [](const Entry &a, const Entry &b) { return a.off[0] < b.off[0]; });
    int64_t used = 0;
    for (int64_t i = 0; i < n; i++) {
        const Entry &e = found[(size_t)i];
        if (used + e.name_len + 1 > len) return -1;
        memcpy(names + used, e.name, (size_t)e.name_len);
        used += e.name_len;
        names[used++] = 0;
        dtype_code[i] = e.code; ndim[i] = e.nd;
        for (int d = 0; d < 8; d++) shape[8 * i + d] = d < e.nd ? e.shape[d] : 0;
        offsets[2 * i] = e.off[0]; offsets[2 * i + 1] = e.off[1];
    }
    *names_bytes = used;
    return n;
}

}  // extern "C"
