"""Pins the PRODUCT's own bits: the exact int64 water-balance limbs and the sha256 of
every result array of a fixed 64 x 128 x 400-day terrain grid (seed 777000), as librsplash_hip.so
computes them on an MI355X.  tests/test_gpu_parity.py::test_product_bits_are_pinned compares them
exactly, so a commit that moves a bit anywhere in the pipeline (soil_hydro, snow threshold, spin-ups,
run_all, monthly totals) is seen -- and has to say so: re-run this on the GPU box, commit the new file
and add a line to DESIGN.md's "product bits changelog".

    python tools/make_product_bits.py "<one line: what moved the bits>"  > gpurun_out/product_bits.json
"""
import hashlib
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

SEED, H, W, NT = 777000, 64, 128, 400


def compute(api, oracle):
    """the pinned call: inputs from the oracle's counter-based generator (host arrays), everything else
    through splash_grid_run"""
    n = H * W
    doy, year, month, mid, nm = oracle.calendar(2001, NT)
    st = oracle.synth_statics(SEED, H, W, 0, n, 1)
    sw, tc, pn = oracle.synth_forcing(SEED, H, W, 0, n, 0, doy, st["lat"], st["elev"])
    dates = np.datetime64("2001-01-01") + np.arange(NT)
    basin = (np.arange(n) // (n // 4)).astype(np.int32).clip(0, 3)
    g = api.splash_grid(sw, tc, pn, st["elev"], st["soil"], st["lat"], st["slop"], st["asp"], st["au"],
                        st["resolution"], time_index=dates, sim_control=dict(monthly_out=False), want_aux=True,
                        want_diag=True, basin=basin, n_basin=4)
    sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
    out = {"grid": f"{H}x{W} terrain, seed {SEED}, {NT} days from 2001-01-01, 4 basins",
           "diag_acc_shape": list(g["diag_acc"].shape),
           "diag_acc": [int(x) for x in g["diag_acc"].ravel()],
           "sha256": {k: sha(g[k]) for k in api.OUT_NAMES}}
    out["sha256"]["state_end"] = sha(g["state_end"])
    out["sha256"]["AI"] = sha(g["aux"]["AI"])
    out["sha256"]["Tt"] = sha(g["aux"]["Tt"])
    out["passes2_sum"] = int(g["aux"]["passes2"].sum())
    return out


if __name__ == "__main__":
    from rsplash_amd import api
    from oracle.binding import Oracle
    r = compute(api, Oracle())
    r["why"] = sys.argv[1] if len(sys.argv) > 1 else ""
    print(json.dumps(r))
