// How fast can one process put bytes into ONE new file on this box?  (the BAM writer's question: 23.6 kB per read go out)
// build + run: g++ -O2 -pthread tools/write_bench.cpp -o /tmp/write_bench && /tmp/write_bench <dir> [GB]
// Variants: pwrite from T threads into one descriptor (what plo_bam_writer does), the same after fallocate, memcpy into a MAP_SHARED mapping
// of the file from T threads (with / without fallocate), T separate files (what the page cache can take when no inode lock is shared),
// O_DIRECT pwrite.  Every variant writes a new file; the time includes close() / munmap(), not the unlink.
#define _GNU_SOURCE 1
#include <fcntl.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <string>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

template <class F>
static void par(int T, size_t n, F f) {
    std::atomic<size_t> next{0};
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t)
        th.emplace_back([&]() {
            for (;;) {
                size_t i = next.fetch_add(1);
                if (i >= n) break;
                f(i);
            }
        });
    for (auto &t : th) t.join();
}

int main(int argc, char **argv) {
    const std::string dir = argc > 1 ? argv[1] : "/tmp";
    const size_t total = (size_t)(argc > 2 ? atof(argv[2]) * (1 << 30) : 4.0 * (1 << 30));
    const size_t chunk = 16u << 20, nsrc = 256u << 20, nchunks = total / chunk;
    uint8_t *src = nullptr;
    if (posix_memalign((void **)&src, 1 << 21, nsrc)) return 1;
    for (size_t i = 0; i < nsrc; ++i) src[i] = (uint8_t)(i * 2654435761u >> 13);
    const std::string path = dir + "/write_bench.dat";
    auto report = [&](const char *what, int T, double s) { printf("%-58s T=%2d  %.3f s  %.2f GB/s\n", what, T, s, total / s / 1e9); fflush(stdout); };
    for (int T : {8, 16}) {
        for (int fa = 0; fa < 2; ++fa) {
            unlink(path.c_str());
            int fd = open(path.c_str(), O_CREAT | O_TRUNC | O_WRONLY, 0644);
            double t0 = now();
            if (fa && posix_fallocate(fd, 0, (off_t)total)) perror("fallocate");
            par(T, nchunks, [&](size_t i) {
                size_t off = i * chunk, left = chunk;
                const uint8_t *p = src + (off % nsrc);
                while (left) {
                    ssize_t k = pwrite(fd, p, left, (off_t)off);
                    if (k <= 0) { perror("pwrite"); exit(1); }
                    p += k, off += (size_t)k, left -= (size_t)k;
                }
            });
            close(fd);
            report(fa ? "pwrite, one descriptor, after fallocate" : "pwrite, one descriptor", T, now() - t0);
        }
        for (int fa = 0; fa < 2; ++fa) {
            unlink(path.c_str());
            int fd = open(path.c_str(), O_CREAT | O_TRUNC | O_RDWR, 0644);
            double t0 = now();
            if (fa ? posix_fallocate(fd, 0, (off_t)total) : ftruncate(fd, (off_t)total)) perror("size");
            uint8_t *m = (uint8_t *)mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
            if (m == MAP_FAILED) { perror("mmap"); return 1; }
            par(T, nchunks, [&](size_t i) { memcpy(m + i * chunk, src + (i * chunk % nsrc), chunk); });
            munmap(m, total);
            close(fd);
            report(fa ? "memcpy into a MAP_SHARED mapping, after fallocate" : "memcpy into a MAP_SHARED mapping (ftruncate)", T, now() - t0);
        }
        {
            double t0 = now();
            std::vector<int> fds(T);
            for (int t = 0; t < T; ++t) {
                std::string p = path + "." + std::to_string(t);
                unlink(p.c_str());
                fds[t] = open(p.c_str(), O_CREAT | O_TRUNC | O_WRONLY, 0644);
            }
            t0 = now();
            std::vector<std::thread> th;
            for (int t = 0; t < T; ++t)
                th.emplace_back([&, t]() {
                    size_t off = 0;
                    for (size_t i = (size_t)t; i < nchunks; i += (size_t)T) {
                        if (pwrite(fds[t], src + (i * chunk % nsrc), chunk, (off_t)off) != (ssize_t)chunk) { perror("pwrite"); exit(1); }
                        off += chunk;
                    }
                });
            for (auto &x : th) x.join();
            for (int t = 0; t < T; ++t) close(fds[t]);
            report("pwrite, one file per thread", T, now() - t0);
            for (int t = 0; t < T; ++t) unlink((path + "." + std::to_string(t)).c_str());
        }
        {
            unlink(path.c_str());
            int fd = open(path.c_str(), O_CREAT | O_TRUNC | O_WRONLY | O_DIRECT, 0644);
            if (fd < 0) {
                printf("O_DIRECT: open failed (%s)\n", strerror(errno));
            } else {
                double t0 = now();
                std::atomic<int> bad{0};
                par(T, nchunks, [&](size_t i) {
                    if (pwrite(fd, src + (i * chunk % nsrc), chunk, (off_t)(i * chunk)) != (ssize_t)chunk) bad = 1;
                });
                close(fd);
                if (bad) printf("O_DIRECT: pwrite failed\n");
                else report("pwrite, O_DIRECT, one descriptor", T, now() - t0);
            }
        }
    }
    unlink(path.c_str());
    return 0;
}
