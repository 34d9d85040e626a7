// Drives the C++ host mirror (include/splash.hpp) the way an Rcpp-free C++ caller would.
//   test_cpp_mirror                  link/throw check: without a GPU the call must throw
//   test_cpp_mirror in.bin out.bin [host]   one site: spin_up on year y, then run_all over the same year
//                                           (third argument "host": SPLASH_BACKEND_HOST, no GPU needed)
// in.bin : doubles  n, y, lat, elv, slop, asp, soil_info[12], sw[n], ta[n], pn[n], snowfall[n]
// out.bin: doubles  spin_up {sm,snow,qin,tdrain,ro,snwage,pet}[n], run_all 11 series [n]
#include <cstdio>
#include <vector>
#include "../../include/splash.hpp"

static std::vector<double> take(const std::vector<double>& v, size_t& at, size_t n)
{
    std::vector<double> r(v.begin() + at, v.begin() + at + n);
    at += n;
    return r;
}

int main(int argc, char** argv)
{
    if (argc < 3) {
        splash::SPLASH sp(44.83, 1783.08);
        try {
            double th = sp.moist_surf(1.07, 10.0, -802.95, 150.0, 257.3, 23.8, 0.37);
            std::printf("moist_surf = %.6f\n", th);
            return (th > 0.0 && th < 1.0) ? 0 : 1;
        } catch (const std::runtime_error& e) {
            std::printf("threw: %s\n", e.what());
            return splash_hip_device_count() == 0 ? 0 : 1;
        }
    }
    std::FILE* f = std::fopen(argv[1], "rb");
    if (!f) return 2;
    std::fseek(f, 0, SEEK_END);
    std::vector<double> in((size_t)std::ftell(f) / sizeof(double));
    std::fseek(f, 0, SEEK_SET);
    if (std::fread(in.data(), sizeof(double), in.size(), f) != in.size()) return 2;
    std::fclose(f);
    const int n = (int)in[0], y = (int)in[1];
    size_t at = 6;
    std::vector<double> soil = take(in, at, 12), sw = take(in, at, n), ta = take(in, at, n),
                        pn = take(in, at, n), sf = take(in, at, n);
    const bool on_host = argc > 3 && std::string(argv[3]) == "host";
    splash::SPLASH sp(in[2], in[3], on_host ? SPLASH_BACKEND_HOST : SPLASH_BACKEND_HIP);
    splash::List s = sp.spin_up(n, y, sw, ta, pn, in[4], in[5], sf, soil);
    std::vector<int> doys(n), yrs(n, y);
    for (int i = 0; i < n; i++) doys[i] = i + 1;
    splash::List r = sp.run_all(doys, yrs, sw, ta, pn, s["sm"].back(), in[4], in[5], s["snow"].back(),
                                sf, soil, s["qin"].back(), s["tdrain"].back(), s["snwage"].back());
    f = std::fopen(argv[2], "wb");
    for (const char* k : {"sm", "snow", "qin", "tdrain", "ro", "snwage", "pet"})
        std::fwrite(s[k].data(), sizeof(double), n, f);
    for (const char* k : {"wn", "ro", "pet", "aet", "snow", "cond", "bflow", "netr", "qin_prev", "tdrain", "snwage"})
        std::fwrite(r[k].data(), sizeof(double), n, f);
    std::fclose(f);
    return 0;
}
