// Micro-benchmark: issue/latency of FP64 VALU instruction streams of ONE wavefront on gfx950.
// Answers what bounds a wave that is alone on its SIMD (the spin-up tail and small grids):
// time per instruction for dependent chains vs independent streams, quarter-rate ops, partial
// exec masks, and 1 vs 2 waves per SIMD.   hipcc --offload-arch=gfx950 -O2 valu_latency.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int MODE>
__global__ void k(double* out, int iters, int lanes)
{
    double a = 1.0 + threadIdx.x * 1e-9, b = 0.999999, c = 1e-9;
    double a1 = a + 1e-3, a2 = a + 2e-3, a3 = a + 3e-3;
    float fa = (float)a;
    if ((int)(threadIdx.x & 63) >= lanes) return;
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) asm volatile(REP64("v_fma_f64 %0, %0, %1, %2\n") : "+v"(a) : "v"(b), "v"(c));
        if (MODE == 1)
            asm volatile(REP8(REP8("v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3\n"))
                         : "+v"(a), "+v"(a1) : "v"(b), "v"(c));
        if (MODE == 2)
            asm volatile(REP8(REP8("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n "
                                   "v_fma_f64 %3, %3, %4, %5\n"))
                         : "+v"(a), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));
        if (MODE == 3) asm volatile(REP64("v_mul_f64 %0, %0, %1\n") : "+v"(a) : "v"(b));
        if (MODE == 4) asm volatile(REP64("v_add_f64 %0, %0, %1\n") : "+v"(a) : "v"(c));
        if (MODE == 5) asm volatile(REP64("v_rcp_f64 %0, %0\n") : "+v"(a));
        if (MODE == 6) asm volatile(REP64("v_fma_f32 %0, %0, %1, %2\n") : "+v"(fa) : "v"((float)b), "v"((float)c));
        if (MODE == 7) // Horner step as generated: constant from SGPR pair set by two s_mov
            asm volatile(REP64("s_mov_b32 s100, 0x12345678\n s_mov_b32 s101, 0x3ff01234\n v_fma_f64 %0, %0, %1, s[100:101]\n")
                         : "+v"(a) : "v"(b) : "s100", "s101");
        if (MODE == 8) asm volatile(REP64("v_fma_f64 %0, %0, %2, %3\n v_cndmask_b32 %1, %1, %1, vcc\n") : "+v"(a), "+v"(fa) : "v"(b), "v"(c));
        if (MODE == 9) asm volatile(REP64("v_rsq_f64 %0, %0\n") : "+v"(a));
        if (MODE == 10) asm volatile(REP64("v_fmac_f64 %0, %1, %2\n") : "+v"(a) : "v"(b), "v"(c));
        if (MODE == 11) asm volatile(REP64("v_fma_f64 %0, %0, %1, %2\n s_nop 0\n") : "+v"(a) : "v"(b), "v"(c));
        if (MODE == 12) asm volatile(REP64("v_ldexp_f64 %0, %0, 0\n") : "+v"(a));
        if (MODE == 13) asm volatile(REP64("v_cvt_f32_f64 %1, %0\n v_cvt_f64_f32 %0, %1\n") : "+v"(a), "+v"(fa));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + a1 + a2 + a3 + fa;
}

template <int MODE>
static void run(const char* name, int per_iter, int threads, int lanes, double* d)
{
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(threads), 0, 0, d, 100, lanes);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(threads), 0, 0, d, iters, lanes);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::printf("%-44s threads %4d lanes %2d : %7.3f ns per instruction per wave\n", name, threads, lanes,
                ms * 1e6 / ((double)iters * per_iter));
}

int main()
{
    double* d;
    hipMalloc(&d, 1 << 20);
    int clk = 0;
    hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
    std::printf("clock rate attribute: %d kHz\n", clk);
    run<0>("v_fma_f64 dependent", 64, 64, 64, d);
    run<1>("v_fma_f64 2 independent chains", 128, 64, 64, d);
    run<2>("v_fma_f64 4 independent chains", 256, 64, 64, d);
    run<3>("v_mul_f64 dependent", 64, 64, 64, d);
    run<4>("v_add_f64 dependent", 64, 64, 64, d);
    run<10>("v_fmac_f64 (VOP2) dependent", 64, 64, 64, d);
    run<5>("v_rcp_f64 dependent", 64, 64, 64, d);
    run<9>("v_rsq_f64 dependent", 64, 64, 64, d);
    run<12>("v_ldexp_f64 dependent", 64, 64, 64, d);
    run<13>("v_cvt f64->f32->f64 dependent", 128, 64, 64, d);
    run<6>("v_fma_f32 dependent", 64, 64, 64, d);
    run<7>("s_mov x2 + v_fma_f64 (count 3)", 192, 64, 64, d);
    run<8>("v_fma_f64 + v_cndmask (count 2)", 128, 64, 64, d);
    run<11>("v_fma_f64 + s_nop 0 (count 2)", 128, 64, 64, d);
    run<0>("v_fma_f64 dependent, 16 lanes", 64, 64, 16, d);
    run<0>("v_fma_f64 dependent, 32 lanes", 64, 64, 32, d);
    run<0>("v_fma_f64 dependent, 4 waves (1/SIMD)", 64, 256, 64, d);
    run<0>("v_fma_f64 dependent, 8 waves (2/SIMD)", 64, 512, 64, d);
    run<0>("v_fma_f64 dependent, 16 waves (4/SIMD)", 64, 1024, 64, d);
    run<2>("v_fma_f64 4 chains, 8 waves (2/SIMD)", 256, 512, 64, d);
    run<5>("v_rcp_f64 dependent, 8 waves (2/SIMD)", 64, 512, 64, d);
    return 0;
}
