// How many workgroups of a given LDS size and width does a CU hold at a time?  Every block of a launch spins for a fixed time on the constant
// 100 MHz counter and notes the CU it ran on (HW_ID: XCC, SE, CU) and when it started and ended; the most blocks alive at one instant on
// one CU is the answer, the launch's time over the spin time the check (blocks / (CUs x residents) rounds).  Written for F5, whose
// 76 736-byte, 8-wave blocks were meant to sit two to a CU while the counters said ~7.5 waves a CU: two DO sit there (and sixteen one-wave
// blocks of 9 592 bytes; 32 waves without LDS) -- the empty slots were waves that had finished their read while the block's heaviest
// read was still running (profiles/r06_f5_history.txt).
//   hipcc --offload-arch=gfx950 -O2 -o tools/diag/_lds_residency tools/diag/lds_residency.hip && tools/diag/_lds_residency
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <map>
#include <vector>

struct Note
{
    unsigned long long t0, t1;
    unsigned hw_id, xcc_id;
};

__global__ void hold(Note* out, const long long ticks, const int vgpr_waste)
{
    extern __shared__ unsigned char lds[];
    const unsigned long long t0 = wall_clock64();
    if (threadIdx.x == 0) lds[0] = 1;
    while (static_cast<long long>(wall_clock64() - t0) < ticks) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) {
        unsigned hw = 0, xcc = 0;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        Note n;
        n.t0 = t0;
        n.t1 = wall_clock64();
        n.hw_id = hw;
        n.xcc_id = xcc;
        out[blockIdx.x] = n;
    }
    (void)vgpr_waste;
}

int main()
{
    const int blocks = 16384;
    Note* d = nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&d), sizeof(Note) * blocks) != hipSuccess) return 1;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(hold), hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const long long ticks = 10000; // 100 us
    struct Case
    {
        int lds, threads;
    };
    const Case cases[] = { { 76736, 512 }, { 76800, 512 }, { 65536, 512 }, { 57552, 384 }, { 81920, 512 }, { 38368, 256 }, { 19184, 128 }, { 9592, 64 },
                           { 153472, 1024 }, { 40960, 512 }, { 32768, 512 }, { 1024, 512 },
                           // one-wave workgroups and small blocks: the 32-wave cap
                           { 0, 64 }, { 512, 64 }, { 1024, 64 }, { 4096, 64 }, { 5120, 64 }, { 0, 128 }, { 4096, 128 }, { 0, 256 }, { 4096, 256 } };
    for (const Case& c : cases) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(hold, dim3(blocks), dim3(c.threads), c.lds, 0, d, ticks, 0);
        (void)hipEventRecord(e1, 0);
        if (hipDeviceSynchronize() != hipSuccess) {
            std::printf("lds %d threads %d: launch failed (%s)\n", c.lds, c.threads, hipGetErrorString(hipGetLastError()));
            continue;
        }
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        std::vector<Note> h(blocks);
        (void)hipMemcpy(h.data(), d, sizeof(Note) * blocks, hipMemcpyDeviceToHost);
        // the most blocks alive at one instant on one CU (CU = xcc, se, sh, cu of HW_ID: bits 8-11 cu, 12 sh, 13-15 se)
        std::map<unsigned, std::vector<std::pair<unsigned long long, int>>> ev;
        for (const Note& n : h) {
            const unsigned cu = (n.xcc_id & 0xfu) << 16 | ((n.hw_id >> 8) & 0xffu);
            ev[cu].push_back({ n.t0, +1 });
            ev[cu].push_back({ n.t1, -1 });
        }
        int most = 0;
        double mean_most = 0;
        for (auto& kv : ev) {
            std::sort(kv.second.begin(), kv.second.end());
            int cur = 0, m = 0;
            for (const auto& e : kv.second) {
                cur += e.second;
                m = std::max(m, cur);
            }
            most = std::max(most, m);
            mean_most += m;
        }
        mean_most /= double(ev.size());
        std::printf("lds %6d B, %4d threads (%2d waves): %zu CUs seen, blocks alive at once on a CU: most %d, mean of the CUs' most %.2f -> waves %d; "
                    "launch %.3f ms = %.2f spins (expected %.2f at that residency)\n",
                    c.lds, c.threads, c.threads / 64, ev.size(), most, mean_most, most * c.threads / 64, ms, ms / (ticks / 100000.0),
                    double(blocks) / (double(ev.size()) * most));
    }
    return 0;
}
