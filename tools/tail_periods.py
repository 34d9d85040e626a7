"""Are the cells that run all 1000 spin-up passes periodic?  For given cells of the benchmark grid
the oracle's spin-up #2 is traced (state after day 1 of every pass) and the trace searched for the
first exact repeat: transient mu, period lam.  CPU only.   python tools/tail_periods.py cell ..."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle.binding import Oracle
o = Oracle()
cells = [int(x) for x in sys.argv[1:]]
nt = 365
doy, year, month, mid, nm = o.calendar(2001, nt)
for c in cells:
    st = o.synth_statics(20221130, 4096, 4096, c, 1, 1)
    sw, tc, pn = o.synth_forcing(20221130, 4096, 4096, c, 1, 0, doy, st["lat"], st["elev"])
    buf = np.full((1001, 5), np.nan)
    o.lib.orc_set_spin_trace(buf.ctypes.data_as(C.POINTER(C.c_double)), 1001)
    _, aux = o.grid(doy, year, month, mid, nm, sw, tc, pn, st["lat"], st["elev"], st["slop"], st["asp"],
                    st["soil"], st["au"], st["resolution"], n_threads=1)
    o.lib.orc_set_spin_trace(None, 0)
    p2 = int(aux["passes2"][0])
    tr = buf[1:p2].view(np.int64)          # rows 1..count-1 were written by spin-up #2 (the last one traced)
    seen, mu, lam = {}, None, None
    for k, row in enumerate(map(bytes, tr)):
        if row in seen:
            mu, lam = seen[row], k - seen[row]
            break
        seen[row] = k
    swing = float(np.nanmax(buf[max(1, p2 - 50):p2, 0]) - np.nanmin(buf[max(1, p2 - 50):p2, 0])) if p2 > 3 else 0.0
    print(f"cell {c}: passes {p2}, first exact repeat: transient {mu}, period {lam}; swing of wn over the last 50 passes {swing:.3f} mm")
