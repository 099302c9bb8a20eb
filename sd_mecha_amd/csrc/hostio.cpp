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

}  // extern "C"
