// Micro-benchmark (round 4): do lone wavefronts run at the same speed on every CU of an MI355X?
// The spin-up's tail showed the SAME job taking 600 ms on some CUs and up to 1 580 ms on others
// (tools/solo_tail.py).  This separates the suspects: W workgroups of one wavefront each run
//   mode 0  a dependent chain of v_fma_f64 only (registers);
//   mode 1  the same chain plus, every 1 024 instructions, one 16-byte load issued "a day ahead" from a
//           record whose address moves by STRIDE bytes per step (the read-back loop's pattern: the next
//           day's record of a job lies cap x 160 B = 63 MB further on -- a new page every day);
//   mode 2  as 1 with a small stride (4 KiB): same loads, no new page per step;
//   mode 3  640 FMAs per step in 40 pieces of 16, each preceded by a forward branch that is TAKEN over 32 instructions
//           that never run (what a job alone on its wavefront does with the rare branches of the day step).
// Each wavefront reports its elapsed time (100 MHz counter) and where it ran (XCC_ID, HW_ID).
//   hipcc --offload-arch=gfx950 -O2 cu_variance.hip -o cu_variance && ./cu_variance
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

__global__ void k(int mode, int steps, const double* rec, size_t stride_d, float* ms, unsigned* where)
{
    if (threadIdx.x != 0) return; // one lane, like a job that has the wavefront to itself
    const unsigned long long t0 = wall_clock64();
    double a = 1.0 + blockIdx.x * 1e-9, b = 0.999999, c = 1e-9;
    const double* p = rec + (size_t)blockIdx.x * 20;
    double2 w = mode ? *(const double2*)p : double2{0, 0};
    if (mode == 3) { // 40 forward branches TAKEN per step, each over a block of 32 instructions that is never executed
        for (int i = 0; i < steps; i++) {
#define SKIP_BLOCK "s_branch 1f\n" REP8("v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n") "1:\n" \
                   REP8("v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n")
            asm volatile(REP8(SKIP_BLOCK) REP8(SKIP_BLOCK) REP8(SKIP_BLOCK) REP8(SKIP_BLOCK) REP8(SKIP_BLOCK) : "+v"(a) : "v"(b), "v"(c));
        }
        steps = 0;
    }
    if (mode == 4) { // 640 FMAs per step with 10 LDS round trips whose address depends on the running value
        __shared__ double tab[512];
        for (int j = 0; j < 512; j++) tab[j] = 1e-12 * j;
        for (int i = 0; i < steps; i++) {
            for (int r = 0; r < 10; r++) {
                asm volatile(REP64("v_fma_f64 %0, %0, %1, %2\n") : "+v"(a) : "v"(b), "v"(c));
                const int idx = (int)(__double2hiint(a) & 255) * 2;
                const double2 t = *(const double2*)&tab[idx];
                a += t.x * 1e-300 + t.y * 1e-300;
            }
        }
        steps = 0;
    }
    for (int i = 0; i < steps; i++) {
        const double2 cur = w;
        if (mode) w = *(const double2*)(p + (size_t)((i + 1) % 365) * stride_d); // tomorrow's, untouched until tomorrow
        for (int r = 0; r < 16; r++) asm volatile(REP64("v_fma_f64 %0, %0, %1, %2\n") : "+v"(a) : "v"(b), "v"(c));
        a += cur.x * 1e-300;
    }
    unsigned hwid, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    ms[blockIdx.x] = (float)((double)(wall_clock64() - t0) * 1e-5) + (a == 123.0 ? 1.0f : 0.0f);
    where[blockIdx.x] = (hwid & 0xFFFF) | ((xcc & 0xF) << 16);
}

int main()
{
    const int W = 256, steps = 100000; // 1.6e8 dependent FMAs per wavefront: ~0.3 s at 4.47 cycles
    const size_t stride_big = (size_t)393216 * 20, stride_small = 512; // doubles
    double* rec;
    float* d_ms;
    unsigned* d_where;
    const size_t bytes = (366 * stride_big + 65536) * sizeof(double);
    if (hipMalloc(&rec, bytes) != hipSuccess) { std::printf("alloc failed\n"); return 1; }
    hipMemset(rec, 0, bytes);
    hipMalloc(&d_ms, W * sizeof(float));
    hipMalloc(&d_where, W * sizeof(unsigned));
    std::vector<float> ms(W);
    std::vector<unsigned> wh(W);
    for (int rep = 0; rep < 2; rep++)
        for (int mode = 0; mode < 5; mode++) {
            hipLaunchKernelGGL(k, dim3(W), dim3(64), 0, 0, mode, mode >= 3 ? steps * 2 : steps, rec, mode == 2 ? stride_small : stride_big, d_ms, d_where);
            hipDeviceSynchronize();
            hipMemcpy(ms.data(), d_ms, W * sizeof(float), hipMemcpyDeviceToHost);
            hipMemcpy(wh.data(), d_where, W * sizeof(unsigned), hipMemcpyDeviceToHost);
            std::vector<float> s = ms;
            std::sort(s.begin(), s.end());
            std::printf("mode %d (%s): %d lone wavefronts: min %.1f ms, median %.1f, p90 %.1f, max %.1f; per XCC max:", mode,
                        mode == 0 ? "FMA chain only" : (mode == 1 ? "+ one load per 1024 FMAs, 63 MB apart" : (mode == 2 ? "+ one load per 1024 FMAs, 4 KiB apart" : (mode == 3 ? "640 FMAs + 40 taken branches per step" : "640 FMAs + 10 dependent LDS reads per step"))),
                        W, s[0], s[W / 2], s[W * 9 / 10], s[W - 1]);
            float mx[8] = {0};
            for (int i = 0; i < W; i++) mx[(wh[i] >> 16) & 7] = std::max(mx[(wh[i] >> 16) & 7], ms[i]);
            for (int x = 0; x < 8; x++) std::printf(" %.0f", mx[x]);
            std::printf("\n");
            if (rep == 1 && s[W - 1] > 1.3f * s[0]) {
                std::printf("   slowest:");
                for (int i = 0; i < W; i++)
                    if (ms[i] > 1.3f * s[0]) std::printf(" [xcc %u se %u cu %u: %.0f]", wh[i] >> 16, (wh[i] >> 13) & 7, (wh[i] >> 8) & 15, ms[i]);
                std::printf("\n");
            }
        }
    return 0;
}
