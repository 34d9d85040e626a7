"""Second, independent restatement (numpy, vectorised the way the R code is) of the R pieces of the
path that have no runnable reference in this image: soil_hydro (R/splash.point.R:232-416),
snowfall_prob / frain_func (:521-578) and the rain/snow split (:120-128).  Used only to cross-check
the oracle's C restatement of the same text (tests/test_oracle_golden.py)."""
import numpy as np


def soil_hydro(sand, clay, OM, fgravel=0.0, bd=np.nan):
    sand, clay, OM, fgravel, bd = (np.asarray(x, dtype=float) for x in (sand, clay, OM, fgravel, bd))
    fsand, fclay, fOM, fgravel = sand / 100, clay / 100, OM / 100, fgravel / 100      # :261-264
    depth = 30                                                                      # :268
    dp = 1 / ((fOM / 1.3) + ((1 - fOM) / 2.65))                                     # :269
    bd = np.where(np.isnan(bd), (1.5 + (dp - 1.5 - 1.10 * (1 - fclay)) * (1 - np.exp(-0.022 * depth)))
                  / (1 + 6.27 * fOM), bd)                                           # :276
    bd = np.where(bd < 0.81, 0.81, bd)                                              # :280
    sat = 1 - (bd / dp)                                                             # :285
    fc = (sat / bd) * (0.4760944 + (0.9402962 - 0.4760944) * fclay ** 0.5) * \
        np.exp(-1 * (0.05472678 * fsand - 0.01 * fOM) / (sat / bd))                 # :287
    wp_Ball = fc * (0.2018522 + (0.7809203 - 0.2018522) * fclay ** 0.5)             # :289
    wp = -2.464e-05 * sand + 3.650e-03 * clay + 8.680e-03 * OM + 9.393e-03 * bd     # :294
    sel = ~np.isnan(wp) & (wp >= fc)
    wp = np.where(sel, wp_Ball, wp)                                                 # :295
    coef_B = (np.log(1500) - np.log(33)) / (np.log(fc) - np.log(wp))                # :319
    coef_A = np.exp(np.log(33) + coef_B * np.log(fc))                               # :320
    coeff_c = 1000.0 / (997 * 9.80665)                                              # :326
    theta_c = (coeff_c * coef_A / 2.0) ** (1 / (1 + coef_B))                        # :327
    theta_r = (0.0285 + 0.00336 * clay) * bd                                        # :340
    theta_r = np.where(~np.isnan(theta_r) & (theta_r > wp), wp, theta_r)            # :341
    sat, fc, wp = sat * (1 - fgravel), fc * (1 - fgravel), wp * (1 - fgravel)       # :351-353
    ksat = 857.48454 / (1 + np.exp(-2.70927 * fsand + 3.62264 * bd + 7.33398 * fclay
                                   - 8.11795 * (sat - fc) + 18.75552 * fOM
                                   + 1.03319 * (1 / coef_B)))                       # :355-363
    m33i = 0.278 * fsand + 0.034 * fclay + 0.022 * fOM - 0.018 * (fsand * fOM) \
        - 0.027 * (fclay * fOM) - 0.584 * (fsand * fclay) + 0.078                   # :369
    m33 = m33i + (0.636 * m33i - 0.107)                                             # :370
    bub_init = -21.6 * fsand - 27.93 * fclay - 81.97 * m33 + 71.12 * (fsand * m33) \
        + 8.29 * (fclay * m33) + 14.05 * (fsand * fclay) + 27.16                    # :371
    bub = (bub_init + (0.02 * bub_init ** 2 - 0.113 * bub_init - 0.7)) * -101.97162129779  # :372-374
    bub = np.where(~np.isnan(bub) & (bub > 0), coef_A * -101.97162129779, bub)      # :379
    return dict(SAT=sat, FC=fc, WP=wp, bd=bd, AWC=fc - wp, Ksat=ksat, A=coef_A, B=coef_B,
                theta_c=theta_c, RES=theta_r * (1 - fgravel), bubbling_p=bub)       # :401-411


def snowfall_prob(tc, lat, elev):                                                   # :576
    return 1 / (1 + np.exp(-0.4710405934 + 1.0473543991 * tc - elev * 0.0004596581
                           - np.abs(lat) * 0.0110592101))


def frain_func(tc, Tt, Tr, month):                                                  # :547-555
    dsin = lambda d: np.sin(d * (np.pi / 180))
    Ttm = Tt + (Tt * dsin((month + 2) / 1.91))
    Trm = Tr * (0.55 + dsin(month + 4)) * 0.6
    x = (tc - Ttm) / (1.4 * Trm)
    fr = np.where(tc <= Ttm, 5 * x ** 3 + 6.76 * x ** 2 + 3.19 * x + 0.5,
                  5 * x ** 3 - 6.76 * x ** 2 + 3.19 * x + 0.5)
    return np.clip(fr, 0, 1)


def rain_snow(tc, pn, month, lat, elev):                                            # :120-128
    p = snowfall_prob(tc, lat, elev)
    Tt = np.max(tc[p >= 0.5]) if (p >= 0.5).any() else -np.inf
    with np.errstate(invalid="ignore"):
        f_rain = np.where(p >= 0.5, frain_func(tc, Tt, 13.3, month), 1.0)
    return Tt, pn * f_rain, pn * (1 - f_rain)
