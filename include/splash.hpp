// splash.hpp -- C++ host mirror of the reference's compiled interface for the path:
// class SPLASH as the Rcpp module exposes it (src/SPLASH.cpp:12-21, src/SPLASH.h:67-108), with the
// same constructor and method argument lists, over the C ABI of librsplash_hip.so.
// Where the reference returns an Rcpp::List of named numeric vectors, these methods return
// splash::List (a map from the same names to std::vector<double>); errors throw std::runtime_error
// (Rcpp turns C++ exceptions into R errors the same way).  Header-only; link -lrsplash_hip.
#pragma once
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

#include "splash_hip.h"

namespace splash {

using List = std::map<std::string, std::vector<double>>;

inline void check(int rc, const char* what)
{
    if (rc != SPLASH_OK)
        throw std::runtime_error(std::string(what) + " failed (" + std::to_string(rc) +
                                 "): " + splash_hip_last_error());
}

class SPLASH {
  public:
    // src/SPLASH.cpp:90.  backend (ABI 6): SPLASH_BACKEND_HIP (default) or SPLASH_BACKEND_HOST -- where this object's
    // methods compute; the library keeps the choice per calling thread and every method sets it first.
    SPLASH(double lat, double elv, int backend = SPLASH_BACKEND_HIP) : lat_(lat), elv_(elv), backend_(backend) {}

    // src/SPLASH.cpp:1594 -> sm, snow, qin, tdrain, ro, snwage, pet (:1748)
    List spin_up(int n, int y, std::vector<double>& sw_in, std::vector<double>& tair,
                 std::vector<double>& pn, double slop, double asp, std::vector<double>& snowfall,
                 std::vector<double>& soil_info) const
    {
        static const char* names[7] = {"sm", "snow", "qin", "tdrain", "ro", "snwage", "pet"};
        enter();
        List out;
        double* p[7];
        for (int k = 0; k < 7; k++) p[k] = series(out, names[k], (size_t)n);
        int passes = 0;
        check(splash_mod_spin_up(lat_, elv_, n, y, sw_in.data(), tair.data(), pn.data(), slop, asp,
                                 snowfall.data(), soil_info.data(), p[0], p[1], p[2], p[3], p[4],
                                 p[5], p[6], &passes),
              "SPLASH::spin_up");
        return out;
    }

    // src/SPLASH.cpp:1833 -> wn, ro, pet, aet, snow, cond, bflow, netr, qin_prev, tdrain, snwage
    List run_all(std::vector<int>& doys, std::vector<int>& yrs, std::vector<double>& sw_in,
                 std::vector<double>& tair, std::vector<double>& pn, double wn_last, double slop,
                 double asp, double snow_last, std::vector<double>& snowfall,
                 std::vector<double>& soil_info, double qin_last, double td_last,
                 double nds_last) const
    {
        const int nt = (int)sw_in.size();
        enter();
        List out;
        double* p[11];
        for (int k = 0; k < 11; k++) p[k] = series(out, kSeries[k], (size_t)nt);
        check(splash_mod_run_all(lat_, elv_, nt, doys.data(), yrs.data(), sw_in.data(), tair.data(),
                                 pn.data(), wn_last, slop, asp, snow_last, snowfall.data(),
                                 soil_info.data(), qin_last, td_last, nds_last, p[0], p[1], p[2],
                                 p[3], p[4], p[5], p[6], p[7], p[8], p[9], p[10]),
              "SPLASH::run_all");
        return out;
    }

    // src/SPLASH.cpp:1751
    List run_one_year(int n, int y, std::vector<double>& sw_in, std::vector<double>& tair,
                      std::vector<double>& pn, std::vector<double>& wn_vec, double slop, double asp,
                      std::vector<double>& snow, std::vector<double>& snowfall,
                      std::vector<double>& soil_info, std::vector<double>& qin_vec,
                      std::vector<double>& td_vec, std::vector<double>& nds) const
    {
        enter();
        List out;
        double* p[11];
        for (int k = 0; k < 11; k++) p[k] = series(out, kSeries[k], (size_t)n);
        check(splash_mod_run_one_year(lat_, elv_, n, y, (int)wn_vec.size(), sw_in.data(), tair.data(),
                                      pn.data(), wn_vec.data(), slop, asp, snow.data(),
                                      snowfall.data(), soil_info.data(), qin_vec.data(),
                                      td_vec.data(), nds.data(), p[0], p[1], p[2], p[3], p[4], p[5],
                                      p[6], p[7], p[8], p[9], p[10]),
              "SPLASH::run_one_year");
        return out;
    }

    // src/SPLASH.cpp:1921
    double moist_surf(double depth, double z, double bub_p, double wn, double SAT, double RES,
                      double lambda) const
    {
        double v = 0;
        enter();
        check(splash_mod_moist_surf(depth, z, bub_p, wn, SAT, RES, lambda, &v), "SPLASH::moist_surf");
        return v;
    }

    // src/SPLASH.cpp:1963
    double inf_GA(double bub_press, double theta_i, double Ksat, double theta_s, double lambda,
                  double P, double tdur, double slop) const
    {
        double v = 0;
        enter();
        check(splash_mod_inf_GA(bub_press, theta_i, Ksat, theta_s, lambda, P, tdur, slop, &v),
              "SPLASH::inf_GA");
        return v;
    }

  private:
    void enter() const
    {
        if (splash_mod_backend(backend_) < 0) check(SPLASH_EINVAL, "SPLASH: backend");
    }
    static double* series(List& out, const char* name, size_t n)
    {
        std::vector<double>& v = out[name];
        v.assign(n, 0.0);
        return v.data();
    }
    double lat_, elv_;
    int backend_;
    static constexpr const char* kSeries[11] = {"wn",   "ro",    "pet",  "aet",      "snow",  "cond",
                                                "bflow", "netr", "qin_prev", "tdrain", "snwage"};
};

} // namespace splash
