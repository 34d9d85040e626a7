// Micro-benchmark (round 4): what a table lookup costs a wavefront that is ALONE on its SIMD and whose 64 lanes all
// do the same work (the last stage of the spin-up's tail: one job per wavefront, the other lanes repeat it).
// The logarithm reads (1/c, log c) of one of 128 intervals from LDS -- a round trip a lone wavefront cannot hide.
// Alternatives measured here, each as a dependent chain (the looked-up value feeds the next index):
//   mode 0  64 dependent v_fma_f64 per step (the unit)
//   mode 1  + one ds_read_b128 whose address depends on the running value          (what the kernel does)
//   mode 2  + the same 16 bytes fetched from VGPRs spread over the lanes: v_readfirstlane of the index, 8 v_readlane
//           (two entries per lane: 128 entries in 8 registers), 4 s_cselect               (uniform index only)
//   mode 3  + s_load_dwordx4 from a table in memory through the scalar cache               (uniform index only)
//   mode 4  + one TAKEN forward branch over 32 instructions
//   mode 5  + one NOT taken branch (s_cbranch_execz with lanes active) around 0 instructions
//   hipcc --offload-arch=gfx950 -O2 lone_wave_lookup.hip -o lone_wave_lookup && ./lone_wave_lookup
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int mode> // (compile-time: a run-time mode would put a different chain of branches into every variant's loop)
__global__ void __launch_bounds__(64) k(int steps, const double* gtab, float* ns_per_step, double* sink)
{
    __shared__ double tab[256];
    for (int j = threadIdx.x; j < 256; j += 64) tab[j] = gtab[j];
    __syncthreads();
    // two entries per lane: entry l in (e0x, e0y), entry l + 64 in (e1x, e1y)
    const double e0x = gtab[2 * threadIdx.x], e0y = gtab[2 * threadIdx.x + 1];
    const double e1x = gtab[2 * (threadIdx.x + 64)], e1y = gtab[2 * (threadIdx.x + 64) + 1];
    double a = 1.0 + blockIdx.x * 1e-9, b = 0.999999, c = 1e-9;
    const unsigned long long t0 = wall_clock64();
    for (int i = 0; i < steps; i++) {
        asm volatile(REP64("v_fma_f64 %0, %0, %1, %2\n") : "+v"(a) : "v"(b), "v"(c));
        const int idx = (__double2hiint(a) >> 4) & 127; // depends on the chain
        double x = 0, y = 0;
        asm volatile("" : "+v"(x), "+v"(y)); // (opaque zeros: the two closing FMAs stay in every variant)
        if (mode == 1) {
            const double2 t = *(const double2*)&tab[2 * idx];
            x = t.x;
            y = t.y;
        } else if (mode == 2) {
            const int u = __builtin_amdgcn_readfirstlane(idx);
            const int lane = u & 63;
            const int lo0 = __builtin_amdgcn_readlane(__double2loint(e0x), lane), hi0 = __builtin_amdgcn_readlane(__double2hiint(e0x), lane);
            const int lo1 = __builtin_amdgcn_readlane(__double2loint(e0y), lane), hi1 = __builtin_amdgcn_readlane(__double2hiint(e0y), lane);
            const int lo2 = __builtin_amdgcn_readlane(__double2loint(e1x), lane), hi2 = __builtin_amdgcn_readlane(__double2hiint(e1x), lane);
            const int lo3 = __builtin_amdgcn_readlane(__double2loint(e1y), lane), hi3 = __builtin_amdgcn_readlane(__double2hiint(e1y), lane);
            const bool up = u >= 64;
            x = __hiloint2double(up ? hi2 : hi0, up ? lo2 : lo0);
            y = __hiloint2double(up ? hi3 : hi1, up ? lo3 : lo1);
        } else if (mode == 3) {
            const int u = __builtin_amdgcn_readfirstlane(idx);
            const double* p = gtab + 2 * u; // uniform address: scalar loads
            x = p[0];
            y = p[1];
        } else if (mode == 4) {
            asm volatile("s_branch 1f\n" REP8("v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n") "1:\n" : "+v"(a) : "v"(b), "v"(c));
        } else if (mode == 5) {
            asm volatile("s_cbranch_execz 1f\n v_fma_f64 %0, %0, %1, %2\n 1:\n" : "+v"(a) : "v"(b), "v"(c));
        }
        a = fma(x, 1e-300, a);
        a = fma(y, 1e-300, a);
    }
    const unsigned long long t1 = wall_clock64();
    if (threadIdx.x == 0) ns_per_step[blockIdx.x] = (float)((double)(t1 - t0) * 10.0 / steps);
    if (a == 123.0) sink[0] = a + e0x + e0y + e1x + e1y;
}

int main()
{
    const int W = 64, steps = 200000;
    double* gtab;
    float* d_ns;
    double* sink;
    hipMalloc(&gtab, 256 * sizeof(double));
    hipMalloc(&d_ns, W * sizeof(float));
    hipMalloc(&sink, 8);
    std::vector<double> h(256);
    for (int j = 0; j < 256; j++) h[j] = 1e-12 * j;
    hipMemcpy(gtab, h.data(), 256 * sizeof(double), hipMemcpyHostToDevice);
    static const char* names[6] = {"64 dependent FMAs", "+ ds_read_b128, dependent address", "+ 8 v_readlane + 4 s_cselect (uniform index)",
                                   "+ scalar loads of 16 B (uniform index)", "+ one taken branch over 32 instructions", "+ one branch not taken"};
    float base = 0;
    for (int rep = 0; rep < 2; rep++)
        for (int mode = 0; mode < 6; mode++) {
            switch (mode) {
            case 0: hipLaunchKernelGGL(k<0>, dim3(W), dim3(64), 0, 0, steps, gtab, d_ns, sink); break;
            case 1: hipLaunchKernelGGL(k<1>, dim3(W), dim3(64), 0, 0, steps, gtab, d_ns, sink); break;
            case 2: hipLaunchKernelGGL(k<2>, dim3(W), dim3(64), 0, 0, steps, gtab, d_ns, sink); break;
            case 3: hipLaunchKernelGGL(k<3>, dim3(W), dim3(64), 0, 0, steps, gtab, d_ns, sink); break;
            case 4: hipLaunchKernelGGL(k<4>, dim3(W), dim3(64), 0, 0, steps, gtab, d_ns, sink); break;
            default: hipLaunchKernelGGL(k<5>, dim3(W), dim3(64), 0, 0, steps, gtab, d_ns, sink); break;
            }
            hipDeviceSynchronize();
            std::vector<float> ns(W);
            hipMemcpy(ns.data(), d_ns, W * sizeof(float), hipMemcpyDeviceToHost);
            std::sort(ns.begin(), ns.end());
            if (mode == 0) base = ns[W / 2];
            if (rep == 1)
                std::printf("mode %d  %-52s median %7.1f ns per step (min %.1f, max %.1f)  extra over mode 0: %6.1f ns = %5.1f FMAs\n", mode, names[mode],
                            ns[W / 2], ns[0], ns[W - 1], ns[W / 2] - base, (ns[W / 2] - base) / (base / 66.0)); std::fflush(stdout);
        }
    return 0;
}
