"""The reference's only "test" (SURVEY.md 4): its README example (README.md:28-61) runs splash.point on the bundled
Bourne SNOTEL site and PLOTS simulated snow water equivalent and soil water content over the observations that ship
with the data set.  examples/readme_bourne.py is that example on this library; here what the plots show is asserted:
the simulated snowpack follows the observed one (8 years of daily SNOTEL pillow data), and the water content of the
measured part of the profile (unSWC to 0.49 m) follows the sensors'.  This is evidence of a kind the oracle cannot
give: measurements, not a restatement -- a wrong sign, unit or branch in the day step (radiation, melt, rain/snow
split, bucket) would show here even if product and oracle shared it."""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))


def test_readme_example_on_the_host_backend_follows_the_observed_snowpack():
    import rsplash_amd
    rsplash_amd.build_library()
    import readme_bourne
    res, run1 = readme_bourne.run("host", with_unswc=False)
    swe = res["swe"]
    print("SWE sim. vs obs.:", swe)
    assert swe["n"] == 2922
    assert swe["r"] > 0.95 and swe["nse"] > 0.85 and abs(swe["bias"]) < 30.0, swe
    assert set(run1) >= {"wn", "ro", "pet", "aet", "snow", "cond", "bflow", "netr", "sm_lim"}   # R/splash.point.R:182-224


@pytest.mark.gpu
def test_readme_example_on_the_gpu_follows_the_observed_snowpack_and_soil_water(api):
    import readme_bourne
    res, _ = readme_bourne.run("hip", with_unswc=True)
    host, _ = readme_bourne.run("host", with_unswc=False)
    print("GPU:", res, "host:", host)
    swe, swc = res["swe"], res["swc"]
    assert swe["r"] > 0.95 and swe["nse"] > 0.85 and abs(swe["bias"]) < 30.0, swe
    assert abs(swe["r"] - host["swe"]["r"]) < 1e-6 and abs(swe["nse"] - host["swe"]["nse"]) < 1e-6
    # soil water of the top 0.49 m against the sensors: the seasonal cycle is there (the README's second plot)
    assert swc["n"] > 2000 and swc["r"] > 0.8 and swc["nse"] > 0.4 and abs(swc["bias"]) < 20.0, swc   # measured: 0.866, 0.60, -5.1 mm
