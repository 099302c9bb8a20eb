// write_bench.cpp -- how fast can one new file on tmpfs / the page cache be filled from user memory?  (tuning harness for
// sd_mecha_amd/csrc/hostio.cpp, not part of the library.)  Build: g++ -O2 -pthread -o write_bench write_bench.cpp
#include <fcntl.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv) {
    const char *path = argc > 1 ? argv[1] : "/dev/shm/write_bench.bin";
    const size_t total = (size_t)(argc > 2 ? atof(argv[2]) : 4.0) * (1ull << 30);
    const size_t piece = 4ull << 20;
    char *src = (char *)mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    memset(src, 1, total);
    {
        FILE *f = fopen("/sys/kernel/mm/transparent_hugepage/shmem_enabled", "r");
        char line[256] = "?";
        if (f) { if (!fgets(line, sizeof line, f)) line[0] = 0; fclose(f); }
        printf("transparent_hugepage/shmem_enabled: %s", line);
    }
    // modes 5-7 (round 2): 5 = fallocate + mapping, every worker batch-faults its piece with MADV_POPULATE_WRITE before the copy;
    // 6 = no fallocate, the workers' MADV_POPULATE_WRITE allocates the pages too (in parallel); 7 = as 2 with MADV_HUGEPAGE
    for (int mode = 0; mode < 8; mode++) {
        for (int threads : {1, 4, 16}) {
            unlink(path);
            int fd = open(path, O_RDWR | O_CREAT | O_TRUNC, 0644);
            double t0 = now();
            if (ftruncate(fd, total) != 0) return 1;
            if (mode == 2 || mode == 4 || mode == 5 || mode == 7) { if (posix_fallocate(fd, 0, total) != 0) return 1; }
            double t_alloc = now() - t0;
            char *map = nullptr;
            if (mode == 1 || mode == 2 || mode >= 5) map = (char *)mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
            if (mode == 7) madvise(map, total, 14 /* MADV_HUGEPAGE */);
            if (mode == 3) { map = (char *)mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_POPULATE, fd, 0); }
            std::atomic<size_t> next{0};
            auto worker = [&]() {
                for (;;) {
                    size_t off = next.fetch_add(piece);
                    if (off >= total) return;
                    size_t len = total - off < piece ? total - off : piece;
                    if (map && (mode == 5 || mode == 6)) madvise(map + off, len, 23 /* MADV_POPULATE_WRITE */);
                    if (map) memcpy(map + off, src + off, len);
                    else { size_t done = 0; while (done < len) done += pwrite(fd, src + off + done, len - done, off + done); }
                }
            };
            std::vector<std::thread> team;
            for (int t = 1; t < threads; t++) team.emplace_back(worker);
            worker();
            for (auto &t : team) t.join();
            if (map) munmap(map, total);
            close(fd);
            double dt = now() - t0;
            const char *names[] = {"pwrite", "mmap", "fallocate+mmap", "mmap populate", "fallocate+pwrite", "falloc+madv populate", "madv populate only", "fallocate+mmap huge"};
            printf("%-18s threads=%2d  %.3f s  %.2f GB/s (prealloc %.3f s)\n", names[mode], threads, dt, total / dt / 1e9, t_alloc);
            fflush(stdout);
        }
    }
    unlink(path);
    return 0;
}
