// h2d_probe.cpp -- how fast can a file that sits in the page cache reach HBM, first touch, one process per variant?
// build: hipcc --offload-arch=gfx950 -O2 -o tools/h2d_probe tools/h2d_probe.cpp -lpthread ; run: tools/h2d_probe <file> <variant> [threads] [piece MiB]
//   0  hipMemcpy from the mmap'ed file (what the CLI's load path does per chunk)
//   1  N threads pread() into page-locked pieces (ring of 2 per thread), hipMemcpyAsync per piece on a stream per thread
//   2  as 1, the threads memcpy() from the mmap instead of pread()
//   3  mmap with MAP_POPULATE, then as 0
//   4  N threads, each ONE pageable hipMemcpyAsync of its share of the mmap'ed file on its own stream (round 4)
//   5  hipHostRegister of the whole mapping (MAP_SHARED), then one hipMemcpyAsync from it (round 4)
//   6  N threads, each registering its next piece of the mapping, copying it asynchronously, unregistering the piece before (round 4)
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main(int argc, char **argv) {
    if (argc < 3) return 2;
    const char *path = argv[1];
    const int variant = atoi(argv[2]);
    const int nthr = argc > 3 ? atoi(argv[3]) : 8;
    const size_t piece = (size_t)(argc > 4 ? atoi(argv[4]) : 8192) << 10;     // KiB
    const double t00 = now();
    int fd = open(path, O_RDONLY);
    struct stat sb;
    fstat(fd, &sb);
    const size_t n = (size_t)sb.st_size;
    CK(hipSetDevice(0));
    CK(hipFree(0));
    const double t0 = now();
    void *dev = nullptr;
    CK(hipMalloc(&dev, n + 4096));
    const double t1 = now();
    double t_setup = 0;
    if (variant == 4) {
        const uint8_t *m = (const uint8_t *)mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
        std::vector<hipStream_t> st((size_t)nthr);
        for (auto &s : st) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        t_setup = now() - t1;
        std::vector<std::thread> th;
        const size_t share = ((n + (size_t)nthr - 1) / (size_t)nthr + 4095) & ~(size_t)4095;
        for (int k = 0; k < nthr; ++k) th.emplace_back([&, k] {
            (void)hipSetDevice(0);
            const size_t off = (size_t)k * share;
            if (off >= n) return;
            const size_t len = std::min(share, n - off);
            (void)hipMemcpyAsync((uint8_t *)dev + off, m + off, len, hipMemcpyHostToDevice, st[(size_t)k]);
            (void)hipStreamSynchronize(st[(size_t)k]);
        });
        for (auto &t : th) t.join();
    } else if (variant == 5) {
        void *m = mmap(nullptr, n, PROT_READ, MAP_SHARED, fd, 0);
        const double ta = now();
        hipError_t e = hipHostRegister(m, n, hipHostRegisterDefault);
        if (e != hipSuccess) { (void)hipGetLastError(); e = hipHostRegister(m, n, 0x08 /* hipHostRegisterReadOnly */); }
        if (e != hipSuccess) { fprintf(stderr, "hipHostRegister: %s\n", hipGetErrorString(e)); return 1; }
        const double tb = now();
        t_setup = 0;
        CK(hipMemcpyAsync(dev, m, n, hipMemcpyHostToDevice, 0));
        CK(hipDeviceSynchronize());
        fprintf(stderr, "  hipHostRegister %.3f s, copy %.3f s\n", tb - ta, now() - tb);
    } else if (variant == 6) {
        uint8_t *m = (uint8_t *)mmap(nullptr, n, PROT_READ, MAP_SHARED, fd, 0);
        std::vector<hipStream_t> st((size_t)nthr);
        for (auto &s : st) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        t_setup = now() - t1;
        std::vector<std::thread> th;
        for (int k = 0; k < nthr; ++k) th.emplace_back([&, k] {
            (void)hipSetDevice(0);
            uint8_t *prev = nullptr;
            for (size_t off = (size_t)k * piece; off < n; off += (size_t)nthr * piece) {
                const size_t len = std::min(piece, n - off);
                hipError_t e = hipHostRegister(m + off, len, hipHostRegisterDefault);
                if (e != hipSuccess) { (void)hipGetLastError(); e = hipHostRegister(m + off, len, 0x08); }
                if (e != hipSuccess) { fprintf(stderr, "hipHostRegister piece: %s\n", hipGetErrorString(e)); return; }
                (void)hipMemcpyAsync((uint8_t *)dev + off, m + off, len, hipMemcpyHostToDevice, st[(size_t)k]);
                if (prev) { (void)hipStreamSynchronize(st[(size_t)k]); (void)hipHostUnregister(prev); (void)hipHostUnregister(m + off); prev = nullptr; }
                else prev = m + off;
            }
            (void)hipStreamSynchronize(st[(size_t)k]);
        });
        for (auto &t : th) t.join();
    } else if (variant == 0 || variant == 3) {
        void *m = mmap(nullptr, n, PROT_READ, MAP_PRIVATE | (variant == 3 ? MAP_POPULATE : 0), fd, 0);
        t_setup = now() - t1;
        CK(hipMemcpy(dev, m, n, hipMemcpyHostToDevice));
    } else {
        const uint8_t *m = (const uint8_t *)mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
        std::vector<void *> pin((size_t)2 * nthr);
        void *ring = nullptr;
        const double ta = now();
        CK(hipHostMalloc(&ring, piece * 2 * (size_t)nthr, getenv("H2D_NONCOHERENT") ? hipHostMallocNonCoherent : hipHostMallocDefault));     // ONE allocation, carved
        for (size_t k = 0; k < pin.size(); ++k) pin[k] = (uint8_t *)ring + k * piece;
        const double tb = now();
        std::vector<hipStream_t> st((size_t)nthr);
        for (auto &s : st) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        t_setup = now() - t1;
        fprintf(stderr, "  hipHostMalloc %.3f s for %zu MiB, %d streams %.3f s\n", tb - ta, (piece * 2 * (size_t)nthr) >> 20, nthr, now() - tb);
        std::vector<std::thread> th;
        for (int k = 0; k < nthr; ++k) th.emplace_back([&, k] {
            (void)hipSetDevice(0);
            hipEvent_t ev[2];
            (void)hipEventCreateWithFlags(&ev[0], hipEventDisableTiming); (void)hipEventCreateWithFlags(&ev[1], hipEventDisableTiming);
            bool used[2] = {false, false};
            size_t j = 0;
            for (size_t off = (size_t)k * piece; off < n; off += (size_t)nthr * piece, ++j) {
                const int b = (int)(j & 1);
                if (used[b]) (void)hipEventSynchronize(ev[b]);
                const size_t len = std::min(piece, n - off);
                uint8_t *dstp = (uint8_t *)pin[(size_t)2 * k + b];
                if (variant == 1) { size_t got = 0; while (got < len) { ssize_t r = pread(fd, dstp + got, len - got, (off_t)(off + got)); if (r <= 0) break; got += (size_t)r; } }
                else memcpy(dstp, m + off, len);
                (void)hipMemcpyAsync((uint8_t *)dev + off, dstp, len, hipMemcpyHostToDevice, st[(size_t)k]);
                (void)hipEventRecord(ev[b], st[(size_t)k]);
                used[b] = true;
            }
            (void)hipStreamSynchronize(st[(size_t)k]);
        });
        for (auto &t : th) t.join();
    }
    CK(hipDeviceSynchronize());
    const double t2 = now();
    printf("variant %d threads %d piece %zu KiB: %.2f GB file, runtime init %.3f s, hipMalloc %.3f s, setup %.3f s, copy %.3f s = %.1f GB/s (alloc + setup + copy %.3f s)\n",
           variant, nthr, piece >> 10, n / 1e9, t0 - t00, t1 - t0, t_setup, t2 - t1 - t_setup, n / (t2 - t1 - t_setup) / 1e9, t2 - t0);
    return 0;
}
