"""Evidence for the fourth conditioning class: cells whose spin-up #2 never meets the
1 mm criterion and is not periodic.  After 1000 passes of such an orbit the state the reference
hands to run_all depends on its own last bits: for the given cells of tools/parity_sweep.py grids the
oracle's source is run as built (A), as another legal build (B: -O3 -march=x86-64-v3 -ffp-contract=fast)
and against "another correct libm" (C: every libm result moved by -1/0/+1 ulp), whole pipeline.
CPU only.     python tools/spinup_conditioning.py g:cell [g:cell ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle.binding import Oracle
from tools.conditioning_evidence import build_variant
nt, W, H = 730, 128, 64
libs = {"A": None,
        "B": build_variant("B", ["-O3", "-march=x86-64-v3", "-ffp-contract=fast"]),
        "C": build_variant("C", ["-O2", "-ffp-contract=off", "-DORC_PERTURB_LIBM"])}
orc = {k: Oracle(v) for k, v in libs.items()}
o = orc["A"]
doy, year, month, mid, nm = o.calendar(2001, nt)
for item in sys.argv[1:]:
    g, cell = (int(x) for x in item.split(":"))
    seed = 777000 + 7919 * g
    st = o.synth_statics(seed, H, W, cell, 1, 1)
    sw, tc, pn = o.synth_forcing(seed, H, W, cell, 1, 0, doy, st["lat"], st["elev"])
    res = {}
    for k, oo in orc.items():
        ref, aux = oo.grid(doy, year, month, mid, nm, sw, tc, pn, st["lat"], st["elev"], st["slop"], st["asp"],
                           st["soil"], st["au"], st["resolution"], monthly_out=False, n_threads=1)
        res[k] = (ref, aux)
    a_ref, a_aux = res["A"]
    line = [f"grid {g} cell {cell}: passes1 A/B/C {int(a_aux['passes1'][0])}/{int(res['B'][1]['passes1'][0])}/{int(res['C'][1]['passes1'][0])}",
            f"passes2 A/B/C {int(a_aux['passes2'][0])}/{int(res['B'][1]['passes2'][0])}/{int(res['C'][1]['passes2'][0])}",
            f"critical {int(a_aux['critical'][0])}"]
    for k in ("B", "C"):
        ref, aux = res[k]
        dw0 = abs(float(aux["state1"][0][0]) - float(a_aux["state1"][0][0]))
        worst = max(float(np.nanmax(np.abs(ref[s] - a_ref[s]) / np.maximum(np.abs(a_ref[s]), 1e-3))) for s in ("wn", "ro", "aet", "pet", "snow"))
        line.append(f"{k} vs A: |d wn| after the spin-up {dw0:.3e} mm, worst relative difference of the five outputs {worst:.2e}")
    print("; ".join(line), flush=True)
