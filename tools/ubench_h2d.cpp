// How fast can a file in the page cache reach HBM?  (one-shot scorer: 14 GB of .issl per process)
//   hipcc -O2 -o tools/_build/ubench_h2d tools/ubench_h2d.cpp -lpthread ; tools/_build/ubench_h2d /dev/shm/x.bin 8
// Strategies: (a) hipMemcpy straight from the private file mapping (what the upload does); (b) the same after
// MADV_POPULATE_READ on threads; (c) pread by T threads into a ring of pinned chunks + hipMemcpyAsync per chunk;
// (d) hipHostRegister of the mapping + one hipMemcpyAsync.
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <mutex>
#include <sys/mman.h>
#include <sys/stat.h>
#include <thread>
#include <unistd.h>
#include <vector>
static double now() { using namespace std::chrono; return duration<double>(steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); std::exit(1); } } while (0)
int main(int argc, char **argv)
{
    const char *path = argc > 1 ? argv[1] : "/dev/shm/h2d.bin";
    const size_t gb = argc > 2 ? std::atoi(argv[2]) : 8;
    const size_t bytes = gb << 30;
    int fd = ::open(path, O_RDWR | O_CREAT, 0600);
    struct stat st; ::fstat(fd, &st);
    if (static_cast<size_t>(st.st_size) != bytes) {
        ::ftruncate(fd, bytes);
        std::vector<char> blk(64 << 20);
        for (size_t i = 0; i < blk.size(); i += 8) { uint64_t v = i * 0x9E3779B97F4A7C15ull; std::memcpy(&blk[i], &v, 8); }
        for (size_t off = 0; off < bytes; off += blk.size()) if (::pwrite(fd, blk.data(), blk.size(), off) < 0) return 1;
    }
    CK(hipSetDevice(0)); CK(hipFree(nullptr));
    void *dev = nullptr; CK(hipMalloc(&dev, bytes));
    auto report = [&](const char *what, double t) { std::printf("%-72s %7.1f ms  %6.1f GB/s\n", what, t * 1e3, bytes / t / 1e9); std::fflush(stdout); };
    for (int rep = 0; rep < 2; ++rep) {
        { // (a)
            void *m = ::mmap(nullptr, bytes, PROT_READ, MAP_PRIVATE, fd, 0);
            double t = now(); CK(hipMemcpy(dev, m, bytes, hipMemcpyHostToDevice)); report("(a) hipMemcpy from a fresh private file mapping", now() - t);
            t = now(); CK(hipMemcpy(dev, m, bytes, hipMemcpyHostToDevice)); report("(a') again, same mapping (pages mapped)", now() - t);
            ::munmap(m, bytes);
        }
        for (int threads : {4, 16}) { // (b)
            void *m = ::mmap(nullptr, bytes, PROT_READ, MAP_PRIVATE, fd, 0);
            double t = now();
            std::vector<std::thread> pool;
            for (int i = 0; i < threads; ++i) pool.emplace_back([&, i] { const size_t lo = bytes / threads * i, len = bytes / threads; ::madvise(static_cast<char *>(m) + lo, len, 22 /*MADV_POPULATE_READ*/); });
            for (auto &th : pool) th.join();
            const double tp = now() - t;
            CK(hipMemcpy(dev, m, bytes, hipMemcpyHostToDevice));
            char what[128]; std::snprintf(what, sizeof what, "(b) MADV_POPULATE_READ on %d threads (%.0f ms) + hipMemcpy", threads, tp * 1e3);
            report(what, now() - t);
            ::munmap(m, bytes);
        }
        for (int threads : {4, 8, 16}) for (size_t chunk_mb : {16, 64}) { // (c)
            const size_t chunk = chunk_mb << 20, n_chunks = bytes / chunk, ring = 2 * threads;
            std::vector<void *> pin(ring); std::vector<hipEvent_t> ev(ring);
            for (size_t i = 0; i < ring; ++i) { CK(hipHostMalloc(&pin[i], chunk, hipHostMallocDefault)); CK(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming)); }
            hipStream_t s; CK(hipStreamCreate(&s));
            double t = now();
            std::atomic<size_t> next{0};
            std::mutex mu; // stream calls serialised
            std::vector<std::thread> pool;
            for (int w = 0; w < threads; ++w) pool.emplace_back([&, w] {
                CK(hipSetDevice(0));
                size_t mine = 0; // this thread owns ring slots w and w + threads, alternating
                while (true) {
                    const size_t c = next.fetch_add(1);
                    if (c >= n_chunks) break;
                    const size_t slot = w + (mine++ & 1) * threads;
                    CK(hipEventSynchronize(ev[slot])); // the slot's previous copy is done
                    size_t got = 0;
                    while (got < chunk) { ssize_t k = ::pread(fd, static_cast<char *>(pin[slot]) + got, chunk - got, c * chunk + got); if (k <= 0) std::exit(2); got += k; }
                    std::lock_guard<std::mutex> lock(mu);
                    CK(hipMemcpyAsync(static_cast<char *>(dev) + c * chunk, pin[slot], chunk, hipMemcpyHostToDevice, s));
                    CK(hipEventRecord(ev[slot], s));
                }
            });
            for (auto &th : pool) th.join();
            CK(hipStreamSynchronize(s));
            char what[128]; std::snprintf(what, sizeof what, "(c) pread by %d threads into pinned chunks of %zu MiB + async copies", threads, chunk_mb);
            report(what, now() - t);
            for (size_t i = 0; i < ring; ++i) { CK(hipHostFree(pin[i])); CK(hipEventDestroy(ev[i])); }
            CK(hipStreamDestroy(s));
        }
        { // (d)
            void *m = ::mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE, fd, 0);
            double t = now();
            hipError_t e = hipHostRegister(m, bytes, hipHostRegisterDefault);
            if (e == hipSuccess) {
                const double tr = now() - t;
                CK(hipMemcpy(dev, m, bytes, hipMemcpyHostToDevice));
                char what[128]; std::snprintf(what, sizeof what, "(d) hipHostRegister of the mapping (%.0f ms) + hipMemcpy", tr * 1e3);
                report(what, now() - t);
                CK(hipHostUnregister(m));
            } else { std::printf("(d) hipHostRegister failed: %s\n", hipGetErrorString(e)); (void)hipGetLastError(); }
            ::munmap(m, bytes);
        }
    }
    ::close(fd); ::unlink(path);
    return 0;
}
