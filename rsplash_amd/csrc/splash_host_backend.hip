// splash_host_backend.hip -- SPLASH_BACKEND_HOST: the product's own day step on a CPU core.
//
// BASELINE.json configs[0] is "splash.point on the bundled Bourne SNOTEL station, CPU reference path (plumbing, no
// GPU)", and SURVEY.md 8(b) lists a `backend CPU/HIP` option.  A single site is one cell: a chain of (2-1000 spin-up
// passes x 365 + nt) sequential day steps that a lone wavefront steps through at ~1.7-3 us each (every instruction
// is latency) while a CPU core does a day step in ~1 us -- the GPU buys nothing for ONE cell, it buys
// the other 16.7 M.  This translation unit compiles the SAME device functions (splash_device.hpp: soil_hydro_cell,
// cell_setup, day_prefetch, day_terms, visc_terms, day_update, rain_snow_split; splash_math.hpp) for the host and
// drives them the way the kernels do:
//     (unSWC, the epilogue of the README example, as well: splash_unswc.hpp)
//     K1 soil_hydro -> K2 snow threshold -> K4 both spin-ups (records of the year formed once, exact-cycle
//     detection as in k_spin_phase) -> K5 run_all + K6 sm_lim / monthly aggregation + K7 exact monthly totals.
// It is an EXPLICIT option (splash_opts.backend, splash_point_run_backend, splash_mod_backend), never a fallback:
// the default backend stays HIP, and without a HIP device every default call still fails with SPLASH_ENODEV
// (tests/test_abi.py::test_no_gpu_means_loud_failure).  The parity checker (the CPU restatement that tests/ compare with) is not involved: this is the product's
// arithmetic, a second executable of it that the device path is diffed against (host and device agree to rounding,
// not to the bit: IEEE divisions and libm stand where the device uses reciprocal sequences and ocml).
//
// Reference restated: R/splash.point.R:92-214 (driver), src/SPLASH.cpp:1594-1749 (spin_up), :1833-1916 (run_all).
#include "splash_host.hpp"

#include <cmath>
#include <limits>
#include <memory>
#include <new>

#include "splash_device.hpp"
#include "splash_unswc.hpp"

namespace splash {
namespace host {
namespace {

struct YearRecords { // the state-independent part of the spin-up year, formed once (k_spin_phase's cache)
    DayTerms T[365];
    ViscTerms V[365];
    double rain[365], snf[365];
    bool any_snf = false;
};

__attribute__((always_inline)) inline bool same_bits(double a, double b) { return d2u(a) == d2u(b); }

__attribute__((always_inline)) inline double fetch(const HostCellIn& in, int which, int t)
{
    const size_t k = (size_t)t * in.stride;
    if (in.sw) return which == 0 ? in.sw[k] : (which == 1 ? in.tc[k] : in.pn[k]);
    return (double)(which == 0 ? in.sw32[k] : (which == 1 ? in.tc32[k] : in.pn32[k]));
}

// one spin-up (src/SPLASH.cpp:1594-1749) over the year's records.  `first`: the PET of the first pass and the
// year's precipitation are summed for the aridity index (pet_vec is filled in pass 0 only, :1667).  `iterate`: run
// to the stopping rule (spin-up #1 is stopped after its first pass unless its pass count is asked for: its states
// are overwritten by spin-up #2, R/splash.point.R:148-152).  Exact-cycle detection as in k_spin_phase: the state
// after the closing day-1 step determines everything after it; a bit-for-bit repeat (the snow age set aside in a
// dry year once it has passed 64 days) means the remaining passes up to the 1000-pass cap only go round that cycle.
__attribute__((always_inline)) inline State spin_once(const CellConst& c, const YearRecords& Y, bool first, bool iterate,
                                                      double* ai, int* passes)
{
    State w{c.RES, 0.0, 0.0, 0.0, 0.0}; // wn_vec(n,RES), others 0, read at [n-1] (:1633-1648)
    int count = 1;
    State chk{};
    int chk_count = -1;
    int since = 0;
    const bool dry = !Y.any_snf;
    DayOut o;
    for (;;) {
        State cur = w;
        double start_sm = 0.0, pet_sum = 0.0, p_sum = 0.0;
        for (int i = 0; i < 365; i++) {
            day_update(c, Y.T[i], Y.V[i], Y.rain[i], Y.snf[i], cur, o);
            if (i == 0) start_sm = cur.wn;
            if (first && count == 1) { // sum(initial_AI$pet, na.rm = T) / sum(Pinit, na.rm = T), R/splash.point.R:150
                const double pday = Y.rain[i] + Y.snf[i];
                pet_sum += (o.pet == o.pet) ? o.pet : 0.0;
                p_sum += (pday == pday) ? pday : 0.0;
            }
        }
        if (first && count == 1 && ai) *ai = pet_sum / p_sum;
        State nxt = cur; // the closing day-1 step (:1674 / :1729) == day 1 of the next pass
        day_update(c, Y.T[0], Y.V[0], Y.rain[0], Y.snf[0], nxt, o);
        double diff_sm = nxt.wn - start_sm; // :1678-1688
        if (diff_sm < 0) diff_sm = start_sm - nxt.wn;
        if (!((diff_sm > 1.0) && (count < 1000) && iterate)) { // :1697
            if (passes) *passes = count;
            return cur; // initial$sm[365] etc., R/splash.point.R:164-172
        }
        count++;
        since++;
        const bool same4 = chk_count >= 0 && same_bits(nxt.wn, chk.wn) && same_bits(nxt.snow, chk.snow) &&
                           same_bits(nxt.qin, chk.qin) && same_bits(nxt.td, chk.td);
        const bool same_nd = same_bits(nxt.nd, chk.nd);
        if (same4 && (same_nd || (dry && chk.nd >= 64.0))) {
            const int left = (1000 - count) % (count - chk_count);
            if (!same_nd) cur.nd += 365.0 * (double)((1000 - count) - left); // one per skipped day step
            count = 1000 - left;
            chk_count = -1;
        } else if ((since & (since - 1)) == 0) { // passes since the start: 1, 2, 4, 8, ...
            chk = nxt;
            chk_count = count;
        }
        w = cur;
    }
}

// the whole of splash.point for one cell (R/splash.point.R:92-214)
__attribute__((always_inline)) inline int cell_impl(const HostCellIn& in, HostCellOut& out)
{
    const int nt = in.nt;
    double par[P_NPAR];
    soil_hydro_cell(in.lat, in.elev, in.slop, in.asp, in.resolution, in.soil[0], in.soil[1], in.soil[2], in.soil[3],
                    in.soil[4], in.soil[5], in.have_au ? in.au[0] : 0.0, in.have_au ? in.au[1] : 3.0,
                    in.have_au ? in.au[2] : 3.0, par);
    const double e1 = par[P_ELV] * 0.0004596581, e2 = std::fabs(par[P_LAT]) * 0.0110592101;
    // K2: Tt = max(tc[p_snow >= 0.5]) over the whole series, R/splash.point.R:120-122 (k_snow_threshold)
    double Tt;
    if (in.tt) {
        Tt = *in.tt;
    } else {
        double best = -std::numeric_limits<double>::infinity();
        bool gap = false;
        for (int t = 0; t < nt; t++) {
            const double v = fetch(in, 1, t);
            const double x = snow_logit(v, e1, e2);
            if (x <= kSnowLogitMax && v > best) best = v;
            gap |= (x != x);
        }
        Tt = gap ? std::numeric_limits<double>::quiet_NaN() : best;
    }
    out.tt = Tt;
    auto Tt_src = [Tt] { return Tt; };

    CellConst c;
    State st;
    out.passes1 = out.passes2 = 0;
    if (in.state0) { // continuing a run (ABI 5): the caller's state, cellout := the run's aridity index
        out.ai = *in.ai0;
        cell_setup(c, par, 1, 0, out.ai);
        st = State{in.state0[0], in.state0[1], in.state0[2], in.state0[3], in.state0[4]};
    } else {
        // K4: the year's records once -- doy 1..365 of the first year with the months of the records' real dates
        // (splash_calendar: spin_days) --, then both spin-ups on them
        std::unique_ptr<YearRecords> Y(new (std::nothrow) YearRecords); // 60 KB: not on a helper thread's stack
        if (!Y) return SPLASH_ENOMEM;
        cell_setup(c, par, 1, 0, par[P_CELLOUT]);
        for (int i = 0; i < 365; i++) {
            const DayRec& d = in.spin_days[i];
            const DayConst dc{d.dr, d.sd, d.cd, d.ttm_s, d.inv_trm14};
            const double swv = fetch(in, 0, i), tcv = fetch(in, 1, i), pnv = fetch(in, 2, i);
            rain_snow_split(dc, tcv, pnv, Tt_src, e1, e2, Y->rain[i], Y->snf[i]);
            Y->any_snf = Y->any_snf || (Y->snf[i] != 0.0); // (true for NaN as well)
            day_terms(c, dc, day_prefetch(c, tcv), swv, tcv, Y->T[i]);
            Y->V[i] = visc_terms(c, Y->T[i]);
        }
        double ai = std::numeric_limits<double>::quiet_NaN();
        (void)spin_once(c, *Y, true, in.want_passes != 0, &ai, &out.passes1);
        out.ai = ai;
        cell_setup(c, par, 1, 0, ai); // soil_info[12] <- AI: ncellout (R/splash.point.R:150; the day terms do not depend on it)
        st = spin_once(c, *Y, false, true, nullptr, &out.passes2);
    }
    par[P_CELLOUT] = out.ai;

    // K5 run_all (src/SPLASH.cpp:1833-1916) + K6 sm_lim and monthly aggregation (R/splash.point.R:197-214) + K7
    const double lim_den = par[P_WMAXR] - c.RES;
    double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    int na_days[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    double dg[4] = {0, 0, 0, 0};
    const bool in_basin = out.diag_rec != nullptr;
    for (int t = 0; t < nt; t++) {
        const DayRec& d = in.days[t];
        const DayConst dc{d.dr, d.sd, d.cd, d.ttm_s, d.inv_trm14};
        const double swv = fetch(in, 0, t), tcv = fetch(in, 1, t), pnv = fetch(in, 2, t);
        double rain, snf;
        rain_snow_split(dc, tcv, pnv, Tt_src, e1, e2, rain, snf);
        DayOut o;
        day_step(c, dc, day_prefetch(c, tcv), swv, tcv, rain, snf, st, o);
        double lim = (st.wn - c.RES) / lim_den;
        if (lim < 0) lim = 0.0;
        if (lim > 1) lim = 1.0;
        const double v[9] = {st.wn, o.ro, o.pet, o.aet, st.snow, o.cond, o.bflow, o.netr, lim};
        if (!in.monthly) {
            const size_t row = (size_t)t * out.ldo;
            for (int q = 0; q < 9; q++) {
                if (out.out[q]) out.out[q][row] = v[q];
                else if (out.out32[q]) out.out32[q][row] = (float)v[q];
            }
        } else { // ctapply(..., FUN = mean | sum, na.rm = T), R/splash.point.R:210-211: NA days are left out of their month
            for (int q = 0; q < 9; q++) {
                acc[q] += (v[q] == v[q]) ? v[q] : 0.0;
                na_days[q] += (v[q] != v[q]);
            }
        }
        if (in_basin) {
            dg[0] += rain + snf;
            dg[1] += o.aet;
            dg[2] += o.ro;
            dg[3] += o.bflow;
        }
        const bool month_end = (d.flags & 1) != 0;
        if (month_end || t == nt - 1) {
            const int m = d.month_id - in.days[0].month_id;
            if (in.monthly) {
                const size_t k = (size_t)m * out.ldo;
                for (int q = 0; q < 9; q++) {
                    // mean over the days that have a value (0 of 0: NaN, as mean(numeric(0))); sum of them
                    const double x = (q == 0 || q == 4 || q == 8) ? acc[q] * (1.0 / (double)(d.mdays - na_days[q])) : acc[q];
                    if (out.out[q]) out.out[q][k] = x;
                    else if (out.out32[q]) out.out32[q][k] = (float)x;
                    acc[q] = 0.0;
                    na_days[q] = 0;
                }
            }
            if (in_basin) { // exact fixed-point limbs, as diag_add() forms them (splash_kernels.hip)
                const double w6[6] = {dg[0], dg[1], dg[2], dg[3], month_end ? st.wn : 0.0, month_end ? st.snow : 0.0};
                long long* rec = out.diag_rec + (size_t)d.month_id * ((SPLASH_NDIAG + 1) * 2);
                long long skipped = 0;
                for (int q = 0; q < SPLASH_NDIAG; q++) {
                    const double x = w6[q];
                    const bool ok = std::fabs(x) < 1048576.0; // false for NaN/inf too
                    const double a = std::rint(x * 65536.0);
                    const double r = x - a * (1.0 / 65536.0);
                    if (ok) {
                        __atomic_fetch_add(rec + 2 * q, (long long)a, __ATOMIC_RELAXED);
                        __atomic_fetch_add(rec + 2 * q + 1, (long long)std::rint(r * 281474976710656.0), __ATOMIC_RELAXED);
                    } else {
                        skipped++;
                    }
                }
                if (month_end) __atomic_fetch_add(rec + 2 * SPLASH_NDIAG, 1ll, __ATOMIC_RELAXED);
                if (skipped) __atomic_fetch_add(rec + 2 * SPLASH_NDIAG + 1, skipped, __ATOMIC_RELAXED);
                dg[0] = dg[1] = dg[2] = dg[3] = 0.0;
            }
        }
    }
    out.state_end[0] = st.wn;
    out.state_end[1] = st.snow;
    out.state_end[2] = st.qin;
    out.state_end[3] = st.td;
    out.state_end[4] = st.nd;
    return SPLASH_OK;
}

// module mirrors: ONE spin-up / run_all of one cell with an explicit snowfall series, soil_info as given
// (src/SPLASH.h:85-103); written like k_spin_up_mod / k_run_all_mod
__attribute__((always_inline)) inline void mod_spin_impl(const double* par, int n, const DayRec* days, const double* sw,
                                                         const double* tc, const double* pn, const double* snowfall,
                                                         double* const series[7], int* passes)
{
    CellConst c;
    cell_setup(c, par, 1, 0, par[P_CELLOUT]);
    State w{c.RES, 0.0, 0.0, 0.0, 0.0};
    int count = 1;
    for (;;) {
        State cur = w;
        DayOut o;
        double start_sm = 0.0;
        for (int i = 0; i < n; i++) {
            const DayConst dc{days[i].dr, days[i].sd, days[i].cd, days[i].ttm_s, days[i].inv_trm14};
            day_step(c, dc, day_prefetch(c, tc[i]), sw[i], tc[i], pn[i], snowfall[i], cur, o);
            if (i == 0) start_sm = cur.wn;
            const double v[7] = {cur.wn, cur.snow, cur.qin, cur.td, o.ro, cur.nd, o.pet};
            for (int k = 0; k < 6; k++)
                if (series[k]) series[k][i] = v[k];
            if (count == 1 && series[6]) series[6][i] = o.pet; // pet from the first pass only (:1667)
        }
        State nxt = cur; // :1674 / :1729
        const DayConst d0{days[0].dr, days[0].sd, days[0].cd, days[0].ttm_s, days[0].inv_trm14};
        day_step(c, d0, day_prefetch(c, tc[0]), sw[0], tc[0], pn[0], snowfall[0], nxt, o);
        double diff_sm = nxt.wn - start_sm; // :1678-1688
        if (diff_sm < 0) diff_sm = start_sm - nxt.wn;
        if (!((diff_sm > 1.0) && (count < 1000))) break; // :1697
        w = cur;
        count++;
    }
    if (passes) *passes = count;
}

__attribute__((always_inline)) inline void mod_run_impl(const double* par, int nt, const DayRec* days, const double* sw,
                                                        const double* tc, const double* pn, const double* snowfall,
                                                        const double* st0, double* const dst[11])
{
    CellConst c;
    cell_setup(c, par, 1, 0, par[P_CELLOUT]);
    State st{st0[0], st0[1], st0[2], st0[3], st0[4]};
    for (int t = 0; t < nt; t++) {
        const DayConst dc{days[t].dr, days[t].sd, days[t].cd, days[t].ttm_s, days[t].inv_trm14};
        DayOut o;
        day_step(c, dc, day_prefetch(c, tc[t]), sw[t], tc[t], pn[t], snowfall[t], st, o);
        const double v[11] = {st.wn, o.ro, o.pet, o.aet, st.snow, o.cond, o.bflow, o.netr, st.qin, st.td, st.nd};
        for (int k = 0; k < 11; k++)
            if (dst[k]) dst[k][t] = v[k];
    }
}

// unSWC (R/unsSWC.R:14-88) on host arrays: wn [nt][ncell], soil [6][ncell] -> wn_z, wtd, theta [nt][ncell] (any NULL)
__attribute__((always_inline)) inline void unswc_impl(size_t ncell, int nt, const double* wn, const double* soil,
                                                      double uns_depth, double* wn_z, double* wtd, double* theta)
{
    for (size_t i = 0; i < ncell; i++) {
        UnswcCell u;
        unswc_setup(soil[i], soil[ncell + i], soil[2 * ncell + i], soil[3 * ncell + i], soil[4 * ncell + i],
                    soil[5 * ncell + i], uns_depth, u);
        for (int t = 0; t < nt; t++) {
            const size_t k = (size_t)t * ncell + i;
            double a, b, c;
            unswc_step(u, wn[k], a, b, c);
            if (wn_z) wn_z[k] = a;
            if (wtd) wtd[k] = b;
            if (theta) theta[k] = c;
        }
    }
}

// unSWC.grid (R/unsSWC.grid.R:14-141) on host arrays: w_z, wtd, Se, theta_mean [nt][ncell] (any NULL)
__attribute__((always_inline)) inline void unswc_grid_impl(size_t ncell, int nt, const double* wn, const double* soil,
                                                           double uns_depth, double* w_z, double* wtd, double* se,
                                                           double* theta_mean)
{
    for (size_t i = 0; i < ncell; i++) {
        UnswcCell u;
        unswc_grid_setup(soil[i], soil[ncell + i], soil[2 * ncell + i], soil[3 * ncell + i], soil[4 * ncell + i],
                         soil[5 * ncell + i], uns_depth, u);
        for (int t = 0; t < nt; t++) {
            const size_t k = (size_t)t * ncell + i;
            double a, b, c, d;
            unswc_grid_step(u, wn[k], a, b, c, d);
            if (w_z) w_z[k] = a;
            if (wtd) wtd[k] = b;
            if (se) se[k] = c;
            if (theta_mean) theta_mean[k] = d;
        }
    }
}

// K2 on host arrays (R/splash.point.R:120-122; k_snow_threshold): Tt = max(tc[p_snow >= 0.5]) per cell, NA if any tc is
template <class T>
__attribute__((always_inline)) inline void snow_threshold_impl(size_t ncell, int nt, const T* tc, const double* lat,
                                                               const double* elev, double* tt)
{
    for (size_t i = 0; i < ncell; i++) {
        const double e1 = elev[i] * 0.0004596581, e2 = std::fabs(lat[i]) * 0.0110592101;
        double best = -std::numeric_limits<double>::infinity();
        bool gap = false;
        for (int t = 0; t < nt; t++) {
            const double v = (double)tc[(size_t)t * ncell + i];
            const double x = snow_logit(v, e1, e2);
            if (x <= kSnowLogitMax && v > best) best = v;
            gap |= (x != x);
        }
        tt[i] = gap ? std::numeric_limits<double>::quiet_NaN() : best;
    }
}

// Two builds of every driver: the generic x86-64 one and one for cores with FMA (every x86 server core since
// 2013), chosen once at run time.  The Horner chains of splash_math.hpp are fma() calls: without the target
// attribute each is a call into libm, with it one vfmadd instruction (measured on the Bourne site: see
// profiles/r06_time_point.txt).
#if defined(__x86_64__)
#define SPL_FMA_TARGET __attribute__((target("avx2,fma")))
#else
#define SPL_FMA_TARGET
#endif
int cell_generic(const HostCellIn& in, HostCellOut& out) { return cell_impl(in, out); }
SPL_FMA_TARGET int cell_fma(const HostCellIn& in, HostCellOut& out) { return cell_impl(in, out); }
void mod_spin_generic(const double* par, int n, const DayRec* days, const double* sw, const double* tc, const double* pn,
                      const double* sf, double* const series[7], int* passes)
{
    mod_spin_impl(par, n, days, sw, tc, pn, sf, series, passes);
}
SPL_FMA_TARGET void mod_spin_fma(const double* par, int n, const DayRec* days, const double* sw, const double* tc,
                                 const double* pn, const double* sf, double* const series[7], int* passes)
{
    mod_spin_impl(par, n, days, sw, tc, pn, sf, series, passes);
}
void mod_run_generic(const double* par, int nt, const DayRec* days, const double* sw, const double* tc, const double* pn,
                     const double* sf, const double* st0, double* const dst[11])
{
    mod_run_impl(par, nt, days, sw, tc, pn, sf, st0, dst);
}
SPL_FMA_TARGET void mod_run_fma(const double* par, int nt, const DayRec* days, const double* sw, const double* tc,
                                const double* pn, const double* sf, const double* st0, double* const dst[11])
{
    mod_run_impl(par, nt, days, sw, tc, pn, sf, st0, dst);
}

void unswc_generic(size_t ncell, int nt, const double* wn, const double* soil, double d, double* a, double* b, double* c)
{
    unswc_impl(ncell, nt, wn, soil, d, a, b, c);
}
SPL_FMA_TARGET void unswc_fma(size_t ncell, int nt, const double* wn, const double* soil, double d, double* a, double* b,
                              double* c)
{
    unswc_impl(ncell, nt, wn, soil, d, a, b, c);
}

void unswc_grid_generic(size_t ncell, int nt, const double* wn, const double* soil, double d, double* a, double* b,
                        double* c, double* e)
{
    unswc_grid_impl(ncell, nt, wn, soil, d, a, b, c, e);
}
SPL_FMA_TARGET void unswc_grid_fma(size_t ncell, int nt, const double* wn, const double* soil, double d, double* a,
                                   double* b, double* c, double* e)
{
    unswc_grid_impl(ncell, nt, wn, soil, d, a, b, c, e);
}

void snow_threshold_generic(size_t ncell, int nt, const double* tc, const float* tc32, const double* lat,
                            const double* elev, double* tt)
{
    if (tc) snow_threshold_impl(ncell, nt, tc, lat, elev, tt);
    else snow_threshold_impl(ncell, nt, tc32, lat, elev, tt);
}
SPL_FMA_TARGET void snow_threshold_fma(size_t ncell, int nt, const double* tc, const float* tc32, const double* lat,
                                       const double* elev, double* tt)
{
    if (tc) snow_threshold_impl(ncell, nt, tc, lat, elev, tt);
    else snow_threshold_impl(ncell, nt, tc32, lat, elev, tt);
}

} // namespace

bool host_backend_fma()
{
#if defined(__x86_64__)
    static const bool have = __builtin_cpu_supports("fma") && __builtin_cpu_supports("avx2");
    return have;
#else
    return false;
#endif
}

int host_cell_run(const HostCellIn& in, HostCellOut& out)
{
    return host_backend_fma() ? cell_fma(in, out) : cell_generic(in, out);
}

void host_mod_spin_up(const double* par, int n, const DayRec* days, const double* sw, const double* tc, const double* pn,
                      const double* snowfall, double* const series[7], int* passes)
{
    if (host_backend_fma()) mod_spin_fma(par, n, days, sw, tc, pn, snowfall, series, passes);
    else mod_spin_generic(par, n, days, sw, tc, pn, snowfall, series, passes);
}

void host_mod_run_all(const double* par, int nt, const DayRec* days, const double* sw, const double* tc, const double* pn,
                      const double* snowfall, const double* st0, double* const dst[11])
{
    if (host_backend_fma()) mod_run_fma(par, nt, days, sw, tc, pn, snowfall, st0, dst);
    else mod_run_generic(par, nt, days, sw, tc, pn, snowfall, st0, dst);
}

void host_snow_threshold(size_t ncell, int nt, const double* tc, const float* tc32, const double* lat, const double* elev,
                         double* tt)
{
    if (host_backend_fma()) snow_threshold_fma(ncell, nt, tc, tc32, lat, elev, tt);
    else snow_threshold_generic(ncell, nt, tc, tc32, lat, elev, tt);
}

// K1 on host arrays: cellpar [P_NPAR][ncell] (soil_hydro_cell is the function k_soil_hydro runs)
void host_soil_hydro(size_t ncell, const double* lat, const double* elev, const double* slop, const double* asp,
                     const double* resolution, const double* soil, const double* au, double* cellpar)
{
    for (size_t i = 0; i < ncell; i++) {
        double col[P_NPAR];
        soil_hydro_cell(lat[i], elev[i], slop[i], asp[i], resolution[i], soil[i], soil[ncell + i], soil[2 * ncell + i],
                        soil[3 * ncell + i], soil[4 * ncell + i], soil[5 * ncell + i], au ? au[i] : 0.0,
                        au ? au[ncell + i] : 3.0, au ? au[2 * ncell + i] : 3.0, col);
        for (int k = 0; k < P_NPAR; k++) cellpar[(size_t)k * ncell + i] = col[k];
    }
}

void host_unswc(size_t ncell, int nt, const double* wn, const double* soil, double uns_depth, double* wn_z, double* wtd,
                double* theta)
{
    if (host_backend_fma()) unswc_fma(ncell, nt, wn, soil, uns_depth, wn_z, wtd, theta);
    else unswc_generic(ncell, nt, wn, soil, uns_depth, wn_z, wtd, theta);
}

void host_unswc_grid(size_t ncell, int nt, const double* wn, const double* soil, double uns_depth, double* w_z,
                     double* wtd, double* se, double* theta_mean)
{
    if (host_backend_fma()) unswc_grid_fma(ncell, nt, wn, soil, uns_depth, w_z, wtd, se, theta_mean);
    else unswc_grid_generic(ncell, nt, wn, soil, uns_depth, w_z, wtd, se, theta_mean);
}

double host_mod_moist_surf(double depth, double z, double bub_p, double wn, double SAT, double RES, double lambda)
{
    return moist_surf_literal(depth, z, bub_p, wn, SAT, RES, lambda);
}

double host_mod_inf_GA(double bub_press, double theta_i, double Ksat, double theta_s, double lambda, double P, double tdur,
                       double slop)
{
    return inf_GA_literal(bub_press, theta_i, Ksat, theta_s, lambda, P, tdur, slop);
}

} // namespace host
} // namespace splash
