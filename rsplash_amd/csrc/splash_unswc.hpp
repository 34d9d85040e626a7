// splash_unswc.hpp -- the unSWC epilogue as host+device functions: k_unswc (splash_statics.hip) and
// the host backend (splash_host_backend.hip) run the same two.  Not part of the day loop's translation unit.
#pragma once
#include "splash_device.hpp"

namespace splash {

// ================================================================ unSWC ==========
// R/unsSWC.R:14-88 as a pointwise epilogue of the column water content wn: water held between the surface and
// uns_depth (analytical integral of the Brooks-Corey profile, :41-43), water-table depth and mean volumetric content
// of that layer.  The soil-only terms (soil_hydro with fgravel = 0, :55) once per cell, then one evaluation per step:
// k_unswc (splash_statics.hip) and the host backend (splash_host_backend.hip) run these two functions.
struct UnswcCell {
    double theta_s, theta_r, bub, lambda, inv_lambda, depth, gfac, gout, uns_depth, u1000, inv_lm1, inv_d1000, inv_dth,
        inv_u1000, inv_bub;
};

SPL_DEV void unswc_setup(double sand, double clay, double OM, double gravel, double bd, double depth, double uns_depth,
                         UnswcCell& u)
{
    // soil_hydro(sand, clay, OM, fgravel = 0, bd): R/splash.point.R:261-380 (fields used here)
    const double fsand = sand / 100, fclay = clay / 100, fOM = OM / 100;
    const double dp = 1 / ((fOM / 1.3) + ((1 - fOM) / 2.65));
    if (isnan(bd))
        bd = (1.5 + (dp - 1.5 - 1.10 * (1 - fclay)) * (1 - exp(-0.022 * 30.0))) / (1 + 6.27 * fOM);
    if (bd < 0.81) bd = 0.81;
    const double theta_s = 1 - (bd / dp);
    const double sq_clay = sqrt(fclay);
    const double fc = (theta_s / bd) * (0.4760944 + (0.9402962 - 0.4760944) * sq_clay) *
                      exp(-1 * (0.05472678 * fsand - 0.01 * fOM) / (theta_s / bd));
    const double wp_Ball = fc * (0.2018522 + (0.7809203 - 0.2018522) * sq_clay);
    double wp = -2.464e-05 * sand + 3.650e-03 * clay + 8.680e-03 * OM + 9.393e-03 * bd;
    if (!isnan(wp) && wp >= fc) wp = wp_Ball;
    const double coef_B = (log(1500.0) - log(33.0)) / (log(fc) - log(wp));
    const double coef_A = exp(log(33.0) + coef_B * log(fc));
    double theta_r = (0.0285 + 0.00336 * (clay)) * bd;
    if (!isnan(theta_r) && theta_r > wp) theta_r = wp;
    const double m33i = 0.278 * fsand + 0.034 * fclay + 0.022 * fOM - 0.018 * (fsand * fOM) -
                        0.027 * (fclay * fOM) - 0.584 * (fsand * fclay) + 0.078;
    const double m33 = m33i + (0.636 * m33i - 0.107);
    const double bub_init = -21.6 * fsand - 27.93 * fclay - 81.97 * m33 + 71.12 * (fsand * m33) +
                            8.29 * (fclay * m33) + 14.05 * (fsand * fclay) + 27.16;
    double bub = (bub_init + (0.02 * (bub_init * bub_init) - 0.113 * bub_init - 0.7)) *
                 -101.97162129779;
    if (!isnan(bub) && bub > 0) bub = coef_A * -101.97162129779;
    const double lambda = 1 / coef_B; // R/unsSWC.R:62
    const double inv_lambda = 1 / lambda;
    const double d1000 = depth * 1000;
    const double dth = theta_s - theta_r;
    const double gfac = (100 + gravel) / 100, gout = (1 - (gravel / 100));
    const double u1000 = uns_depth * 1000;
    const double inv_lm1 = 1 / (lambda - 1);
    const double inv_d1000 = 1 / d1000, inv_dth = 1 / dth, inv_u1000 = 1 / u1000, inv_bub = 1 / bub;
    u = UnswcCell{theta_s, theta_r, bub, lambda, inv_lambda, depth, gfac, gout, uns_depth, u1000, inv_lm1, inv_d1000,
                  inv_dth, inv_u1000, inv_bub};
}

SPL_DEV void unswc_step(const UnswcCell& u, double wn, double& wn_z, double& wtd_out, double& theta_out)
{
    const double theta_s = u.theta_s, theta_r = u.theta_r, bub = u.bub, lambda = u.lambda, inv_lambda = u.inv_lambda;
    const double depth = u.depth, uns_depth = u.uns_depth, u1000 = u.u1000, inv_lm1 = u.inv_lm1, inv_d1000 = u.inv_d1000;
    const double inv_dth = u.inv_dth, inv_bub = u.inv_bub;
    const double w = wn * u.gfac; // :56
    double theta_i = w * inv_d1000;                                               // :64-66
    if (theta_i >= theta_s) theta_i = theta_s - 0.0001;
    if (theta_i <= theta_r) theta_i = theta_r + 0.0001;
    const double Theta = (theta_i - theta_r) * inv_dth; // in (0, 1) after the two clamps (or NaN)
    const double L = flog<kLogPositive>(Theta);
    const double Pm = fexp(-L * inv_lambda);          // Theta^(-1/lambda) >= 1
    const double psi_m = bub * Pm;                    // bub / Theta^(1/lambda), :67
    const double wtdini = (bub - psi_m) * 1.0e-3;     // UnsWater, :36-37
    const double wtd = (wtdini > depth) ? depth : ((wtdini < 0) ? 0 : wtdini);
    const double z_uns = (wtd <= uns_depth) ? wtd * 1000 : u1000; // :39
    // :41-43  (bub/(psi_m+z))^lambda = (Theta^(-1/lambda) + z/bub)^(-lambda): psi_m + z <= bub < 0, so
    // the bracket is >= 1 -- the logarithm of the reciprocal base, without the reciprocal
    const double pz = fexp(-lambda * flog<kLogPositive>(Pm + z_uns * inv_bub));
    const double p0 = Theta; // (bub/psi_m)^lambda at z = 0 is Theta itself
    const double w_uns_z = theta_r * z_uns + (((psi_m + z_uns) * (theta_r - theta_s) * pz) * inv_lm1);
    const double w_uns_0 = ((psi_m * (theta_r - theta_s) * p0) * inv_lm1);
    const double sat_swc = (wtd <= uns_depth) ? theta_s * (uns_depth - wtd) * 1000 : 0; // :46
    double swc_tot = (w_uns_z - w_uns_0) + sat_swc;
    if (swc_tot <= 0) swc_tot = theta_r * u1000; // :74
    theta_out = swc_tot * u.inv_u1000; // :80
    wn_z = swc_tot * u.gout;           // :82
    wtd_out = wtd;
}

// The GRIDDED function (R/unsSWC.grid.R:14-141) is not unSWC applied cell by cell: it leaves the stoniness correction out
// (:65 `fgravel = soil_data[[4]]*0`; the (1 - gravel/100) factor is commented out, :73), has no patch for a non-positive
// integral (R/unsSWC.R:74 has one), hands UnsWater the sensor depth as its `depth` argument (:115 against the signature
// at :29: the water table of the INTEGRAL is capped at uns_depth, the one it writes out -- calcwtd, :98-106 -- at the soil
// depth), and returns the layer's effective saturation (calc_Se, :120-125) and the column's clamped mean theta (:79-87).
// Same setup (the gravel factors of UnswcCell are not used), same logarithm/exponential forms as unswc_step().
SPL_DEV void unswc_grid_setup(double sand, double clay, double OM, double gravel, double bd, double depth,
                              double uns_depth, UnswcCell& u)
{
    unswc_setup(sand, clay, OM, 0.0, bd, depth, uns_depth, u);
    if (isnan(gravel)) { // soil_hydro(..., fgravel = NA * 0, ...): SAT and RES carry the factor (1 - fgravel), R/splash.point.R:315, 410
        u.theta_s = gravel;
        u.theta_r = gravel;
    }
}

SPL_DEV void unswc_grid_step(const UnswcCell& u, double wn, double& w_z, double& wtd_out, double& se_out,
                             double& theta_mean_out)
{
    const double theta_s = u.theta_s, theta_r = u.theta_r, bub = u.bub, lambda = u.lambda, inv_lambda = u.inv_lambda;
    const double uns_depth = u.uns_depth, u1000 = u.u1000, inv_lm1 = u.inv_lm1;
    const double theta_o = wn * u.inv_d1000;                                                    // calc_thetai, :79-86
    double theta_i = (theta_o >= theta_s) ? theta_s - 0.0001 : ((theta_o <= theta_r) ? theta_r + 0.0001 : theta_o);
    if (isnan(theta_o) || isnan(theta_s) || isnan(theta_r)) theta_i = theta_o + theta_s + theta_r; // ifelse(NA, ., .) is NA
    const double Theta = (theta_i - theta_r) * u.inv_dth;
    const double L = flog<kLogPositive>(Theta);
    const double Pm = fexp(-L * inv_lambda);      // Theta^(-1/lambda) >= 1
    const double psi_m = bub * Pm;                // :92
    const double wtdini = (bub - psi_m) * 1.0e-3; // calcwtd, :98-106 (against the soil depth) and UnsWater's own (against uns_depth)
    const double wtd = (wtdini > u.depth) ? u.depth : ((wtdini < 0) ? 0 : wtdini);
    const double wtd2 = (wtdini > uns_depth) ? uns_depth : ((wtdini < 0) ? 0 : wtdini);
    const double z_uns = (wtd2 <= uns_depth) ? wtd2 * 1000 : u1000;
    const double pz = fexp(-lambda * flog<kLogPositive>(Pm + z_uns * u.inv_bub));
    const double w_uns_z = theta_r * z_uns + (((psi_m + z_uns) * (theta_r - theta_s) * pz) * inv_lm1);
    const double w_uns_0 = ((psi_m * (theta_r - theta_s) * Theta) * inv_lm1);
    const double sat_swc = (wtd2 <= uns_depth) ? theta_s * (uns_depth - wtd2) * 1000 : 0;
    const double swc_tot = (w_uns_z - w_uns_0) + sat_swc;
    double Se = (swc_tot * u.inv_u1000) / theta_s; // calc_Se, :120-125
    Se = (Se > 1) ? 1 : ((Se < 0) ? 0 : Se);
    w_z = swc_tot;
    wtd_out = isnan(wtdini) ? wtdini : wtd;
    se_out = isnan(swc_tot) ? swc_tot : Se;
    theta_mean_out = theta_i;
}

} // namespace splash
