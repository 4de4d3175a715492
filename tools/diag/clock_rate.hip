// what clock64() counts on this device: ticks of clock64() (s_memtime) per tick of wall_clock64() (a constant 100 MHz counter) over a
// busy loop, and the same against the host's clock around the launch -- the unit of $SK_F5_TIMING's "cycles per block"
//   hipcc --offload-arch=gfx950 -O2 -o tools/diag/_clock_rate tools/diag/clock_rate.hip && tools/diag/_clock_rate
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>

__global__ void spin(long long* out, const int iters)
{
    const long long c0 = clock64(), w0 = wall_clock64();
    unsigned x = threadIdx.x;
    for (int i = 0; i < iters; ++i) x = x * 1664525u + 1013904223u; // a dependent chain of integer multiply-adds
    const long long c1 = clock64(), w1 = wall_clock64();
    if (threadIdx.x == 0) {
        out[0] = c1 - c0;
        out[1] = w1 - w0;
        out[2] = x;
    }
}

int main()
{
    long long* d = nullptr;
    long long h[3] = { 0, 0, 0 };
    if (hipMalloc(reinterpret_cast<void**>(&d), sizeof(h)) != hipSuccess) return 1;
    int wall_khz = 0, sclk_khz = 0;
    (void)hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0);
    (void)hipDeviceGetAttribute(&sclk_khz, hipDeviceAttributeClockRate, 0);
    for (const int iters : { 1000000, 10000000 }) {
        hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, 0, d, iters);
        (void)hipDeviceSynchronize();
        const auto t0 = std::chrono::steady_clock::now();
        hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, 0, d, iters);
        (void)hipDeviceSynchronize();
        const double host_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        std::printf("iters %d: clock64 %lld ticks, wall_clock64 %lld ticks (rate attribute %d kHz), host %.6f s: clock64 = %.1f MHz by wall_clock64, "
                    "%.1f MHz by the host's clock; %.2f clock64 ticks per iteration (device clock attribute %d kHz)\n",
                    iters, h[0], h[1], wall_khz, host_s, double(h[0]) / double(h[1]) * wall_khz / 1e3, double(h[0]) / host_s / 1e6, double(h[0]) / iters,
                    sclk_khz);
    }
    return 0;
}
