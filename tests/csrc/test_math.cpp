// Host check of rsplash_amd/csrc/splash_math.hpp against libm (compiled with g++ by
// tests/test_host_math.py).  Prints the worst error in ulp of each function; exit code 1 when a
// bound is exceeded.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>

#include "../../rsplash_amd/csrc/splash_math.hpp"

static double ulp_err(double a, double ref)
{
    if (std::isnan(a) && std::isnan(ref)) return 0;
    if (a == ref) return 0;
    if (std::isinf(a) || std::isinf(ref) || std::isnan(a) || std::isnan(ref)) return 1e300;
    double u = std::nextafter(std::fabs(ref), INFINITY) - std::fabs(ref);
    return std::fabs(a - ref) / u;
}

int main()
{
    std::mt19937_64 g(12345);
    std::uniform_real_distribution<double> U(0.0, 1.0);
    double wl = 0, we = 0, wp = 0, wa = 0, wrel_pow = 0;
    const int N = 4000000;
    for (int i = 0; i < N; i++) {
        double u = U(g);
        // log: wide range and the (0, 1.5) range the model uses, incl. values next to 1
        double x = (i & 1) ? std::exp(700 * (u - 0.5)) : (i & 2 ? 1.0 + (u - 0.5) * 1e-6 * (i % 7)
                                                                : u * 1.5);
        if (x > 0) wl = std::fmax(wl, ulp_err(splash::flog(x), std::log(x)));
        double y = (i & 1) ? 1400 * (u - 0.5) : 40 * (u - 0.5);
        we = std::fmax(we, ulp_err(splash::fexp(y), std::exp(y)));
        double th = U(g), ex = 0.3 + 12 * U(g); // Theta^(1/lambda), Theta^(3+1/lambda) ...
        double pr = std::pow(th, ex), pa = splash::fpow(th, ex);
        if (pr > 1e-300) {
            wp = std::fmax(wp, ulp_err(pa, pr));
            wrel_pow = std::fmax(wrel_pow, std::fabs(pa - pr) / pr);
        }
        double c = 2 * u - 1;
        if (i % 5 == 0) c = 1 - u * 1e-9;
        if (i % 5 == 1) c = -1 + u * 1e-9;
        wa = std::fmax(wa, ulp_err(splash::facos(c), std::acos(c)));
    }
    int bad = 0;
    // special values
    bad += !(splash::flog(0.0) == -INFINITY) + !std::isnan(splash::flog(-1.0)) +
           !(splash::flog(INFINITY) == INFINITY) + !std::isnan(splash::flog(NAN)) +
           !(splash::flog(1.0) == 0.0);
    bad += !(splash::fexp(-INFINITY) == 0.0) + !(splash::fexp(INFINITY) == INFINITY) +
           !std::isnan(splash::fexp(NAN)) + !(splash::fexp(0.0) == 1.0) +
           !(splash::fexp(-800.0) == 0.0) + !(splash::fexp(800.0) == INFINITY);
    bad += !(splash::fpow(0.0, 2.7) == 0.0) + !std::isnan(splash::fpow(-0.3, 2.7)) +
           !(splash::fpow(1.0, 123.4) == 1.0) + !(splash::fpow(-0.0, 0.37) == 0.0) +
           !(splash::fpow(0.0, -2.0) == INFINITY);
    bad += !(splash::facos(1.0) == 0.0) + !(std::fabs(splash::facos(-1.0) - M_PI) < 1e-15) +
           !(std::fabs(splash::facos(0.0) - M_PI / 2) < 1e-15);
    printf("flog %.3f ulp  fexp %.3f ulp  fpow %.3f ulp (rel %.3e)  facos %.3f ulp  special-case failures %d\n",
           wl, we, wp, wrel_pow, wa, bad);
    return (wl > 2.0 || we > 1.0 || wrel_pow > 1e-13 || wa > 2.5 || bad) ? 1 : 0;
}
