// How does this box hand a freshly written file to 16 threads: page faults on a private mapping against pread into private
// buffers.  Writes a new file per round (the page cache fills and fragments as rounds go by), then reads it both ways.
// usage: file_read_ubench <dir> <GiB> <rounds> <threads>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv) {
    if (argc < 5) return 2;
    const std::string dir = argv[1];
    const size_t n = (size_t)(atof(argv[2]) * (1ull << 30));
    const int rounds = atoi(argv[3]), T = atoi(argv[4]);
    std::vector<char> blk(8 << 20);
    for (size_t i = 0; i < blk.size(); i++) blk[i] = (char)(i * 2654435761u >> 13);
    for (int r = 0; r < rounds; r++) {
        const std::string path = dir + "/ubench_" + std::to_string(r) + ".bin";
        int fd = open(path.c_str(), O_CREAT | O_TRUNC | O_WRONLY, 0644);
        double t0 = now();
        for (size_t o = 0; o < n; o += blk.size()) {
            blk[0] = (char)o;
            if (write(fd, blk.data(), std::min(blk.size(), n - o)) < 0) return 1;
        }
        close(fd);
        const double tw = now() - t0;
        fd = open(path.c_str(), O_RDONLY);
        std::vector<unsigned long> sums(T);
        auto run = [&](bool mapped) {
            const double t1 = now();
            void *m = mapped ? mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0) : nullptr;
            std::vector<std::thread> th;
            for (int t = 0; t < T; t++)
                th.emplace_back([&, t]() {
                    size_t lo = n * t / T, hi = n * (t + 1) / T;
                    unsigned long s = 0;
                    if (mapped) {
                        const char *p = static_cast<const char *>(m);
                        for (size_t i = lo; i < hi; i += 64) s += (unsigned char)p[i];
                    } else {
                        std::vector<char> buf(4 << 20);
                        while (lo < hi) {
                            const ssize_t k = pread(fd, buf.data(), std::min(buf.size(), hi - lo), (off_t)lo);
                            if (k <= 0) break;
                            for (ssize_t i = 0; i < k; i += 64) s += (unsigned char)buf[i];
                            lo += (size_t)k;
                        }
                    }
                    sums[t] = s;
                });
            for (auto &x : th) x.join();
            if (m) munmap(m, n);
            return now() - t1;
        };
        const double tm = run(true), tp = run(false), tm2 = run(true);
        printf("round %d: write %.2f s; %d threads: mapping %.3f s = %.1f GB/s, pread %.3f s = %.1f GB/s, mapping again %.3f s = %.1f GB/s\n", r, tw, T, tm,
               n / tm / 1e9, tp, n / tp / 1e9, tm2, n / tm2 / 1e9);
        fflush(stdout);
        close(fd);
    }
    for (int r = 0; r < rounds; r++) unlink((dir + "/ubench_" + std::to_string(r) + ".bin").c_str());
    return 0;
}
