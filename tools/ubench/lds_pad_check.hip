// Does a dynamic LDS request beyond half of a CU's 160 KB keep two workgroups off one CU?  (round 4)
// A kernel with 25 856 B of static LDS (like k_spin_phase) is launched with 0 / 56 KB of dynamic LDS, 512 workgroups of
// 256 threads that spin for ~2 ms and record their CU; the number of workgroups seen at once per CU is reported.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
__global__ void __launch_bounds__(256, 2) k(unsigned* where, unsigned long long* t)
{
    __shared__ double pad[3232];
    pad[threadIdx.x] = threadIdx.x;
    __syncthreads();
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < 200000) {} // 2 ms
    if (threadIdx.x == 0) {
        unsigned hwid, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        where[blockIdx.x] = ((hwid >> 8) & 0xFF) | ((xcc & 0xF) << 8);
        t[2 * blockIdx.x] = t0;
        t[2 * blockIdx.x + 1] = wall_clock64();
    }
    if (pad[threadIdx.x] < 0) where[0] = 0;
}
int main()
{
    const int W = 512;
    unsigned* d_w;
    unsigned long long* d_t;
    hipMalloc(&d_w, W * 4);
    hipMalloc(&d_t, W * 16);
    for (int pad : {0, 57344}) {
        hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, pad);
        int occ = -1;
        hipError_t e2 = hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k, 256, pad);
        hipLaunchKernelGGL(k, dim3(W), dim3(256), pad, 0, d_w, d_t);
        hipError_t e3 = hipDeviceSynchronize();
        std::vector<unsigned> w(W);
        std::vector<unsigned long long> t(2 * W);
        hipMemcpy(w.data(), d_w, W * 4, hipMemcpyDeviceToHost);
        hipMemcpy(t.data(), d_t, W * 16, hipMemcpyDeviceToHost);
        int worst = 0;
        for (int i = 0; i < W; i++) {
            int together = 0;
            for (int j = 0; j < W; j++)
                if (w[j] == w[i] && t[2 * j] < t[2 * i + 1] && t[2 * i] < t[2 * j + 1]) together++;
            if (together > worst) worst = together;
        }
        unsigned long long lo = ~0ull, hi = 0;
        for (int i = 0; i < W; i++) { if (t[2 * i] < lo) lo = t[2 * i]; if (t[2 * i + 1] > hi) hi = t[2 * i + 1]; }
        std::printf("dynamic LDS %6d B: set attribute %s, occupancy query %s -> %d per CU, launch %s; most workgroups resident on one CU at once: %d; all %d done after %.1f ms\n",
                    pad, hipGetErrorName(e), hipGetErrorName(e2), occ, hipGetErrorName(e3), worst, W, (hi - lo) * 1e-5); std::fflush(stdout);
        (void)hipGetLastError();
    }
    return 0;
}
