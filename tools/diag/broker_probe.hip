// broker_probe.hip -- two facts the per-GPU broker (csrc/sk_rt.h) stands on, measured on the box:
//   (1) a shared-memory segment mapped at the SAME virtual address in a client and in the server, page-locked in the server with
//       hipHostRegister, is read and written by the server's kernels through that very address (no pointer translation in kernel arguments);
//   (2) N caller threads of ONE process, a stream each, run "jobs" (20 short launches + one wait) side by side, where N caller PROCESSES
//       beyond the device's eight compute slots are time-sliced.
//   hipcc --offload-arch=gfx950 -O2 tools/diag/broker_probe.hip -o tools/diag/_broker_probe -lpthread
//   _broker_probe map | procs N [jobs] | threads N [jobs]        ($GPU_MAX_HW_QUEUES as exported)
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#include <sys/mman.h>
#include <sys/wait.h>
#include <unistd.h>

#define CK(x)                                                                                      \
    do {                                                                                           \
        hipError_t e_ = (x);                                                                       \
        if (e_ != hipSuccess) {                                                                    \
            std::fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__);       \
            std::exit(2);                                                                          \
        }                                                                                          \
    } while (0)

__global__ void touch_kernel(const uint32_t* in, uint32_t* out, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[i] * 3u + 1u;
}

// ~`ticks` of the 100 MHz constant clock (a job's kernels last a few microseconds)
__global__ void short_kernel(uint32_t* sink, int ticks)
{
    const uint64_t t0 = wall_clock64();
    uint32_t acc = threadIdx.x;
    while (wall_clock64() - t0 < (uint64_t)ticks) acc = acc * 1664525u + 1013904223u;
    if (acc == 0x12345u) sink[0] = acc;
}

static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static int probe_map()
{
    const size_t bytes = 1 << 20;
    void* const want = reinterpret_cast<void*>(0x600000000000ull);
    const int fd = memfd_create("sk_probe", 0);
    if (fd < 0 || ftruncate(fd, bytes)) return std::perror("memfd"), 1;
    int pfd[2];
    if (pipe(pfd)) return 1;
    const pid_t child = fork(); // the "client": no HIP in this process
    if (child == 0) {
        void* p = mmap(want, bytes, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_FIXED_NOREPLACE, fd, 0);
        if (p != want) _exit(3);
        uint32_t* w = static_cast<uint32_t*>(p);
        for (int i = 0; i < 1024; ++i) w[i] = 1000u + i;
        char c = 1;
        if (write(pfd[1], &c, 1) != 1) _exit(4);
        sleep(3);
        _exit(0);
    }
    char c;
    if (read(pfd[0], &c, 1) != 1) return 1;
    CK(hipSetDevice(0));
    void* p = mmap(want, bytes, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_FIXED_NOREPLACE, fd, 0);
    std::printf("map: server mapping at %p (wanted %p)\n", p, want);
    if (p != want) return 1;
    const double t0 = now_s();
    CK(hipHostRegister(p, bytes, hipHostRegisterDefault));
    const double t1 = now_s();
    void* dp = nullptr;
    CK(hipHostGetDevicePointer(&dp, p, 0));
    std::printf("map: hipHostRegister %.3f ms; device pointer %p %s host pointer\n", (t1 - t0) * 1e3, dp, dp == p ? "==" : "!=");
    uint32_t* w = static_cast<uint32_t*>(p);
    hipLaunchKernelGGL(touch_kernel, dim3(4), dim3(256), 0, 0, w, w + 4096, 1024); // through the HOST address
    CK(hipDeviceSynchronize());
    int bad = 0;
    for (int i = 0; i < 1024; ++i) bad += (w[4096 + i] != (1000u + i) * 3u + 1u);
    std::printf("map: kernel through the shared address: %d of 1024 wrong\n", bad);
    CK(hipHostUnregister(p));
    int st;
    waitpid(child, &st, 0);
    return bad != 0 || dp != p;
}

struct Shared
{
    std::atomic<int> ready, go;
    double t_begin[64], t_end[64], wait_s[64];
};

static void worker(int id, int jobs, hipStream_t st, uint32_t* sink, Shared* sh, int n)
{
    for (int w = 0; w < 20; ++w) hipLaunchKernelGGL(short_kernel, dim3(8), dim3(64), 0, st, sink, 300);
    CK(hipStreamSynchronize(st));
    sh->ready.fetch_add(1);
    while (sh->go.load() == 0) usleep(100);
    sh->t_begin[id] = now_s();
    double waited = 0;
    for (int j = 0; j < jobs; ++j) {
        for (int k = 0; k < 20; ++k) hipLaunchKernelGGL(short_kernel, dim3(8), dim3(64), 0, st, sink, 300);
        const double w0 = now_s();
        CK(hipStreamSynchronize(st));
        waited += now_s() - w0;
        // the caller's host work between two jobs (a segment process uses the device a few per cent of its time)
        const double h0 = now_s();
        while (now_s() - h0 < 200e-6) {}
    }
    sh->t_end[id] = now_s();
    sh->wait_s[id] = waited;
}

static void report(const char* what, int n, int jobs, Shared* sh)
{
    double b = 1e300, e = 0, w = 0;
    for (int i = 0; i < n; ++i) {
        b = std::min(b, sh->t_begin[i]);
        e = std::max(e, sh->t_end[i]);
        w += sh->wait_s[i];
    }
    const char* q = std::getenv("GPU_MAX_HW_QUEUES");
    std::printf("%s n=%d queues=%s jobs=%d: wall %.3f s, %.1f us per job per caller (200 us of it host work), wait %.1f us per job, %.0f jobs/s in all\n", what, n,
                q ? q : "default", jobs, e - b, (e - b) / jobs * 1e6, w / (double(n) * jobs) * 1e6, double(n) * jobs / (e - b));
}

__global__ __launch_bounds__(256) void copy16_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, const uint64_t n16)
{
    for (uint64_t i = uint64_t(blockIdx.x) * 256 + threadIdx.x; i < n16; i += uint64_t(gridDim.x) * 256) dst[i] = src[i];
}

// (3) what a kernel / a copy call gets out of host memory of three kinds: hipHostMalloc, a memfd segment page-locked with hipHostRegister
//     (4 KB pages: what the broker's clients share), anonymous memory with MADV_HUGEPAGE page-locked the same way
static int probe_mem()
{
    CK(hipSetDevice(0));
    const size_t cap = 32u << 20;
    void* dev;
    CK(hipMalloc(&dev, cap));
    void* kinds[3];
    const char* names[3] = { "hipHostMalloc", "memfd + hipHostRegister", "anonymous huge pages + hipHostRegister" };
    CK(hipHostMalloc(&kinds[0], cap, hipHostMallocDefault));
    const int fd = memfd_create("sk_probe_mem", 0);
    if (fd < 0 || ftruncate(fd, cap)) return 1;
    kinds[1] = mmap(nullptr, cap, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    kinds[2] = mmap(nullptr, cap + (2u << 20), PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    kinds[2] = reinterpret_cast<void*>((reinterpret_cast<uintptr_t>(kinds[2]) + (2u << 20) - 1) & ~uintptr_t((2u << 20) - 1));
    (void)madvise(kinds[2], cap, MADV_HUGEPAGE);
    for (int k = 1; k < 3; ++k) {
        std::memset(kinds[k], 1, cap);
        CK(hipHostRegister(kinds[k], cap, hipHostRegisterDefault));
    }
    std::memset(kinds[0], 1, cap);
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    for (size_t bytes : { size_t(64) << 10, size_t(1) << 20, size_t(8) << 20, size_t(32) << 20 })
        for (int k = 0; k < 3; ++k)
            for (int dir = 0; dir < 2; ++dir)
                for (int how = 0; how < 2; ++how) {
                    const void* src = dir == 0 ? kinds[k] : dev;
                    void* dst = dir == 0 ? dev : kinds[k];
                    const int reps = 20;
                    double best = 1e300;
                    for (int rep = 0; rep < reps; ++rep) {
                        const double t0 = now_s();
                        if (how == 0) hipLaunchKernelGGL(copy16_kernel, dim3(unsigned(std::min<size_t>((bytes / 16 + 255) / 256, 1024))), dim3(256), 0, st, static_cast<const uint4*>(src),
                                                         static_cast<uint4*>(dst), uint64_t(bytes / 16));
                        else CK(hipMemcpyAsync(dst, src, bytes, dir == 0 ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost, st));
                        CK(hipStreamSynchronize(st));
                        best = std::min(best, now_s() - t0);
                    }
                    std::printf("mem: %8zu KB %-40s %s by %-14s best %.1f us = %.2f GB/s\n", bytes >> 10, names[k], dir == 0 ? "host->device" : "device->host",
                                how == 0 ? "kernel" : "hipMemcpyAsync", best * 1e6, double(bytes) / best / 1e9);
                }
    return 0;
}

// (4) what an allocation costs the server: hipMalloc / hipHostRegister of several sizes from one thread and from 16 at once
static int probe_alloc()
{
    CK(hipSetDevice(0));
    void* warm;
    CK(hipMalloc(&warm, 1 << 20));
    static std::mutex one_at_a_time;
    for (int threads : { 1, 16, -16 }) // (-16: sixteen threads, the calls one at a time under a mutex)
        for (size_t mb : { size_t(8), size_t(32), size_t(256), size_t(1024) }) {
            const bool serial = threads < 0;
            if (serial) threads = -threads;
            std::vector<std::thread> th;
            std::vector<double> dev_s(threads), pin_s(threads), free_s(threads);
            for (int t = 0; t < threads; ++t)
                th.emplace_back([&, t] {
                    CK(hipSetDevice(0));
                    const size_t bytes = mb << 20;
                    void* d[3];
                    double t0 = now_s();
                    for (int i = 0; i < 3; ++i) {
                        std::unique_lock<std::mutex> g(one_at_a_time, std::defer_lock);
                        if (serial) g.lock();
                        CK(hipMalloc(&d[i], bytes));
                    }
                    dev_s[t] = (now_s() - t0) / 3;
                    t0 = now_s();
                    for (int i = 0; i < 3; ++i) CK(hipFree(d[i]));
                    free_s[t] = (now_s() - t0) / 3;
                    if (mb <= 256) {
                        void* h = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
                        t0 = now_s();
                        {
                            std::unique_lock<std::mutex> g(one_at_a_time, std::defer_lock);
                            if (serial) g.lock();
                            CK(hipHostRegister(h, bytes, hipHostRegisterDefault));
                        }
                        pin_s[t] = now_s() - t0;
                        CK(hipHostUnregister(h));
                        munmap(h, bytes);
                    }
                });
            for (auto& x : th) x.join();
            double a = 0, b = 0, c = 0;
            for (int t = 0; t < threads; ++t) a += dev_s[t], b += pin_s[t], c += free_s[t];
            std::printf("alloc: %2d threads%s, %4zu MB: hipMalloc %.2f ms, hipFree %.2f ms, hipHostRegister (fresh shared pages) %.2f ms  (mean per call, waiting for the mutex included)\n", threads, serial ? " one call at a time" : "", mb,
                        a / threads * 1e3, c / threads * 1e3, b / threads * 1e3);
        }
    return 0;
}

int main(int argc, char** argv)
{
    if (argc < 2) return 1;
    if (!std::strcmp(argv[1], "map")) return probe_map();
    if (!std::strcmp(argv[1], "alloc")) return probe_alloc();
    if (!std::strcmp(argv[1], "mem")) return probe_mem();
    const int n = argc > 2 ? std::atoi(argv[2]) : 8;
    const int jobs = argc > 3 ? std::atoi(argv[3]) : 2000;
    if (n < 1 || n > 64) return 1;
    Shared* sh = static_cast<Shared*>(mmap(nullptr, sizeof(Shared), PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0));
    new (sh) Shared();
    if (!std::strcmp(argv[1], "procs")) {
        std::vector<pid_t> kids;
        for (int i = 0; i < n; ++i) {
            const pid_t c = fork();
            if (c == 0) {
                CK(hipSetDevice(0));
                (void)hipSetDeviceFlags(hipDeviceScheduleBlockingSync);
                hipStream_t st;
                CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
                uint32_t* sink;
                CK(hipMalloc(reinterpret_cast<void**>(&sink), 256));
                worker(i, jobs, st, sink, sh, n);
                _exit(0);
            }
            kids.push_back(c);
        }
        while (sh->ready.load() < n) usleep(1000);
        sh->go.store(1);
        for (pid_t c : kids) {
            int st;
            waitpid(c, &st, 0);
        }
        report("procs", n, jobs, sh);
    } else {
        CK(hipSetDevice(0));
        (void)hipSetDeviceFlags(hipDeviceScheduleBlockingSync);
        uint32_t* sink;
        CK(hipMalloc(reinterpret_cast<void**>(&sink), 256));
        std::vector<std::thread> th;
        std::vector<hipStream_t> sts(n);
        for (int i = 0; i < n; ++i) CK(hipStreamCreateWithFlags(&sts[i], hipStreamNonBlocking));
        for (int i = 0; i < n; ++i)
            th.emplace_back([=] {
                CK(hipSetDevice(0));
                worker(i, jobs, sts[i], sink, sh, n);
            });
        while (sh->ready.load() < n) usleep(1000);
        sh->go.store(1);
        for (auto& t : th) t.join();
        report("threads", n, jobs, sh);
    }
    return 0;
}
