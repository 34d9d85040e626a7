// Micro-benchmark (round 4): what does hipMalloc of tens of GB cost, and does it parallelise over host threads?
// (splash_grid_run's FIRST call spends 5.3 s allocating ~100 GB of block buffers; later calls reuse them.)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    (void)hipFree(nullptr);
    const size_t GB = (size_t)1 << 30;
    for (size_t sz : {1 * GB, 4 * GB, 12 * GB, 24 * GB}) {
        void* p = nullptr;
        double t0 = now();
        hipError_t e = hipMalloc(&p, sz);
        double t1 = now();
        (void)hipMemset(p, 0, 64);
        (void)hipDeviceSynchronize();
        double t2 = now();
        (void)hipFree(p);
        double t3 = now();
        std::printf("hipMalloc %3zu GB: %.3f s (%s), first touch %.3f s, hipFree %.3f s\n", sz / GB, t1 - t0, hipGetErrorString(e), t2 - t1, t3 - t2);
    }
    for (int nthr : {1, 2, 4, 8}) {
        std::vector<void*> ptr(8, nullptr);
        double t0 = now();
        std::vector<std::thread> th;
        for (int w = 0; w < nthr; w++)
            th.emplace_back([&, w] {
                (void)hipSetDevice(0);
                for (int i = w; i < 8; i += nthr) (void)hipMalloc(&ptr[i], 12 * GB);
            });
        for (auto& t : th) t.join();
        double t1 = now();
        for (void* p : ptr) (void)hipFree(p);
        std::printf("8 x 12 GB on %d host thread(s): %.3f s (free %.3f s)\n", nthr, t1 - t0, now() - t1);
    }
    return 0;
}
