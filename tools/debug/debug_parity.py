"""Debug helper (GPU box): compare GPU run_all vs oracle on the first rows of the 4096-row grid and
print the worst cell-days per output."""
import os, sys, json
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import rsplash_amd
rsplash_amd.build_library()
from rsplash_amd import api
from oracle.binding import Oracle
o = Oracle()
dev = torch.device("cuda:0")
Htot, W, rows, nt, seed = 4096, 4096, int(sys.argv[1]) if len(sys.argv) > 1 else 16, 365, 20221130
n = rows * W
dates = np.datetime64("2001-01-01") + np.arange(nt)
cal = api.Calendar.from_dates(dates)
doy, year, month = api.time_fields(dates)
f64 = dict(dtype=torch.float64, device=dev)
lat, elev, slop, asp, res = (torch.empty(n, **f64) for _ in range(5))
soil, au = torch.empty(6, n, **f64), torch.empty(3, n, **f64)
api.dev_synth_statics(seed, Htot, W, 0, n, 1, lat, elev, slop, asp, res, soil, au)
sw, tc, pn = (torch.empty(nt, n, **f64) for _ in range(3))
api.dev_synth_forcing(seed, cal, 0, nt, 0, n, lat, elev, sw, tc, pn, n)
par = torch.empty(api.NPAR, n, **f64); Tt = torch.empty(n, **f64); state = torch.empty(5, n, **f64)
api.dev_soil_hydro(n, lat, elev, slop, asp, res, soil, au, par)
api.dev_snow_threshold(n, nt, tc, n, par, n, Tt)
work = torch.empty(api.spin_up_work_size(n), dtype=torch.uint8, device=dev)
api.dev_spin_up(cal, n, sw, tc, pn, n, Tt, par, n, state, n, work)
state0 = state.clone()
names = ("wn", "snow", "ro", "aet", "pet")
g = {k: torch.empty(nt, n, **f64) for k in names}
api.dev_run_all(cal, 0, nt, n, sw, tc, pn, n, Tt, par, n, state, n, g, n)
torch.cuda.synchronize()
hp = par.cpu().numpy()
hsw, htc, hpn = sw.cpu().numpy(), tc.cpu().numpy(), pn.cpu().numpy()
out, st_end = o.grid_transient(doy, year, month, hsw, htc, hpn, hp[0], hp[1], hp[2], hp[3], hp[4:16],
                               Tt.cpu().numpy(), state0.cpu().numpy(), n_threads=16)
for i, k in enumerate(names):
    a, b = g[k].cpu().numpy(), out[i]
    err = np.abs(a - b)
    rel = err / np.maximum(np.abs(b), 1e-300)
    big = np.abs(b) > 1e-9
    relm = np.where(big, rel, 0)
    t, c = np.unravel_index(np.argmax(relm), relm.shape)
    print(f"{k}: max rel {relm.max():.3e} at day {t} cell {c}: gpu {a[t,c]!r} cpu {b[t,c]!r}; max abs {err.max():.3e};"
          f" n(rel>1e-5)={int((relm>1e-5).sum())} n(rel>1e-8)={int((relm>1e-8).sum())} of {relm.size}")
    if k == "pet":
        print("  inputs: lat", hp[0][c], "elv", hp[1][c], "slop", hp[2][c], "asp-180", hp[3][c], "sw", hsw[t, c], "tc", htc[t, c],
              "wn prev gpu/cpu", g["wn"][t-1, c].item(), out[0][t-1, c], "snow prev", g["snow"][t-1, c].item(), out[1][t-1, c])
# first divergence of wn per cell: which day/cell first exceeds 1e-9 relative
a, b = g["wn"].cpu().numpy(), out[0]
bad = np.abs(a - b) > 1e-9 * np.abs(b)
cells = np.where(bad.any(0))[0]
from oracle.binding import grazing_mask
frag = o.fragile.copy()
graz = grazing_mask(o, hp[0], hp[2], hp[3], doy, year)
okc = ~graz.any(0) & (frag == 0)
print("cells:", n, "grazing:", int(graz.any(0).sum()), "fragile:", int((frag > 0).sum()), "ok:", int(okc.sum()))
a, b = g["wn"].cpu().numpy(), out[0]
bad = (np.abs(a - b) > 1e-7 * np.abs(b)) & okc[None, :]
cells = np.where(bad.any(0))[0]
print("ok cells with wn rel diff > 1e-7:", len(cells))
for c in cells[:8]:
    t = int(np.argmax(bad[:, c]))
    t0 = t
    for i, k in enumerate(names):
        aa, bb = g[k][:, c].cpu().numpy(), out[i][:, c]
        d = np.abs(aa - bb) > 1e-10 * np.abs(bb) + 1e-13
        if d.any():
            t0 = min(t0, int(np.argmax(d)))
    print(f" cell {c} lat {hp[0][c]:.2f} slop {hp[2][c]:.3f} depth {hp[9][c]:.3f} lam {hp[8][c]:.4f} bub {hp[10][c]:.1f} Au {hp[12][c]:.3e} cellout(AI) {hp[15][c]:.4f}: first >1e-7 day {t}; earliest tiny diff day {t0}")
    for i, k in enumerate(names):
        aa, bb = g[k][:, c].cpu().numpy(), out[i][:, c]
        print("     ", k, aa[max(t0-1,0):t0+3], bb[max(t0-1,0):t0+3])
    print("      forcing tc", htc[t0, c], "pn", hpn[t0, c], "sw", hsw[t0, c])
