// Micro-benchmark (round 4), part 2: WHICH instruction class runs at a different speed on different CUs?
// The spin-up's read-back day loop takes 600 ms on some CUs and up to 1 590 ms on others for the SAME job
// (tools/solo_tail.py SOLO_SAME); FMA chains, LDS round trips, taken branches and strided loads do not
// (cu_variance.hip).  One lane per wavefront, 256 workgroups, per class: min / median / max over the wavefronts.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

__global__ void k(int mode, int steps, float* ms, unsigned* where)
{
    if (threadIdx.x != 0) return;
    const unsigned long long t0 = wall_clock64();
    double a = 1.0 + blockIdx.x * 1e-9, b = 0.999999, c = 1e-9, d = 1.5;
    float f = 1.0f + blockIdx.x * 1e-6f, g = 0.999f;
    int iv = (int)blockIdx.x;
    for (int i = 0; i < steps; i++) {
        switch (mode) {
        case 0: asm volatile(REP64("v_fma_f64 %0, %0, %1, %2\n") : "+v"(a) : "v"(b), "v"(c)); break;
        case 1: asm volatile(REP64("v_rcp_f64 %0, %0\n") : "+v"(a)); break;
        case 2: asm volatile(REP64("v_rsq_f64 %0, %0\n") : "+v"(a)); break;
        case 3: asm volatile(REP64("v_sqrt_f64 %0, %0\n") : "+v"(a)); break;
        case 4: asm volatile(REP64("v_div_scale_f64 %0, vcc, %0, %1, %0\n v_div_fmas_f64 %0, %0, %1, %2\n v_div_fixup_f64 %0, %0, %1, %2\n")
                             : "+v"(a) : "v"(b), "v"(d) : "vcc"); break;
        case 5: asm volatile(REP64("v_frexp_mant_f64 %0, %0\n v_ldexp_f64 %0, %0, 1\n v_rndne_f64 %1, %0\n v_cvt_i32_f64 %2, %1\n v_cvt_f64_i32 %1, %2\n")
                             : "+v"(a), "+v"(d), "+v"(iv)); break;
        case 6: asm volatile(REP64("v_cmp_gt_f64 vcc, %0, %1\n v_cndmask_b32 %2, %2, %3, vcc\n") : "+v"(a), "+v"(b), "+v"(f), "+v"(g) : : "vcc"); break;
        case 7: asm volatile(REP64("v_readlane_b32 s20, %0, 0\n v_writelane_b32 %0, s20, 0\n") : "+v"(iv) : : "s20"); break;
        case 8: asm volatile(REP64("v_cmp_gt_f64 vcc, %0, %1\n s_and_saveexec_b64 s[20:21], vcc\n s_or_b64 exec, exec, s[20:21]\n") : "+v"(a) : "v"(b) : "vcc", "s20", "s21"); break;
        case 9: asm volatile(REP64("v_mov_b64 %0, %1\n v_mov_b64 %1, %0\n") : "+v"(a), "+v"(d)); break;
        case 10: asm volatile(REP64("v_mul_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n") : "+v"(f) : "v"(g)); break;
        case 11: asm volatile(REP64("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 1\n v_mul_f32 %0, %0, %1\n s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0\n") : "+v"(f) : "v"(g)); break;
        case 12: asm volatile(REP64("v_add_f64 %0, %0, %1\n") : "+v"(a) : "v"(c)); break;
        case 13: asm volatile(REP64("v_mul_f64 %0, %0, %1\n") : "+v"(a) : "v"(b)); break;
        case 14: asm volatile(REP64("v_max_f64 %0, %0, %1\n v_min_f64 %0, %0, %2\n") : "+v"(a) : "v"(b), "v"(d)); break;
        case 15: asm volatile(REP64("s_mov_b32 s20, 0x12345678\n s_mov_b32 s21, 0x3ff01234\n v_fma_f64 %0, %0, %1, s[20:21]\n") : "+v"(a) : "v"(b) : "s20", "s21"); break;
        case 16: asm volatile(REP64("s_nop 3\n v_fma_f64 %0, %0, %1, %2\n") : "+v"(a) : "v"(b), "v"(c)); break;
        case 17: asm volatile(REP64("v_cvt_f32_f64 %1, %0\n v_cvt_f64_f32 %0, %1\n") : "+v"(a), "+v"(f)); break;
        }
    }
    unsigned hwid, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    ms[blockIdx.x] = (float)((double)(wall_clock64() - t0) * 1e-5) + ((a + d + f + iv) == 123.0 ? 1.0f : 0.0f);
    where[blockIdx.x] = (hwid & 0xFFFF) | ((xcc & 0xF) << 16);
}

int main()
{
    const int W = 256;
    float* d_ms;
    unsigned* d_where;
    (void)hipMalloc(&d_ms, W * sizeof(float));
    (void)hipMalloc(&d_where, W * sizeof(unsigned));
    std::vector<float> ms(W);
    const char* names[] = {"v_fma_f64", "v_rcp_f64", "v_rsq_f64", "v_sqrt_f64", "div_scale+fmas+fixup", "frexp/ldexp/rndne/cvt", "v_cmp+v_cndmask",
                           "readlane+writelane", "cmp+saveexec+or exec", "v_mov_b64 x2", "v_mul_f32+v_add_f32", "s_setreg MODE x2 + v_mul_f32",
                           "v_add_f64", "v_mul_f64", "v_max_f64+v_min_f64", "s_mov x2 + v_fma(sgpr)", "s_nop 3 + v_fma_f64", "cvt f64<->f32"};
    for (int mode = 0; mode < 18; mode++) {
        hipLaunchKernelGGL(k, dim3(W), dim3(64), 0, 0, mode, 1000, d_ms, d_where);
        (void)hipDeviceSynchronize();
        hipLaunchKernelGGL(k, dim3(W), dim3(64), 0, 0, mode, 200000, d_ms, d_where);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(ms.data(), d_ms, W * sizeof(float), hipMemcpyDeviceToHost);
        std::sort(ms.begin(), ms.end());
        std::printf("%-30s min %7.1f  median %7.1f  p90 %7.1f  max %7.1f ms   max/min %.2f\n", names[mode], ms[0], ms[W / 2], ms[W * 9 / 10], ms[W - 1],
                    ms[W - 1] / ms[0]);
    }
    return 0;
}
