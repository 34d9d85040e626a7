#!/usr/bin/env python3
"""bench.py -- cell-days/s of the gridded SPLASH loop on MI355X (BASELINE.json metric).

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json configs[2], the configuration the metric is quoted on): a synthetic
4096 x 4096 grid with slope/aspect/upslope area, 365 daily steps (calendar 2001), FP64.  With
N > 1 every rank owns a 4096-row block of a (4096*N) x 4096 grid (weak scaling; cells are
independent, so the only exchange is one all-reduce of the monthly water-balance totals).

One "step" = one pass of the hot path over the rank's whole block with all inputs resident in
HBM: K2 (snowfall threshold pre-pass over tc) + K5 (SPLASH::run_all: rain/snow split, SOLAR, EVAP,
infiltration, runoff, lateral flow, soil-water update for every cell and day, daily outputs wn,
snow, ro, aet, pet written to HBM) + the diagnostics totals.  Setup outside the timed region:
on-device input generation, K1 soil_hydro and K4 the two spin-ups (timed and reported separately
as spinup_s).  The grid is processed in row blocks so that forcing (147 GB) + one block of
outputs fit in 288 GB.

The JSON line carries `roofline` (algorithmic 64 B per cell-day over the measured duration of the
run_all kernel, against 8 TB/s) and `cpu_baseline` (the plain-C port of the reference's loop in
oracle/, timed on this box's host cores on a bounded sample of the same workload, and used to
check the GPU results on that sample).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALG_BYTES_PER_CELL_DAY = 64.0   # 3 forcing reads + 5 output writes, FP64
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: 8 TB/s spec
HBM_COPY_GBS = 6290.0           # same guide: measured copy rate
SEED = 20221130
OUT5 = ("wn", "snow", "ro", "aet", "pet")


def host_cores():
    """threads this process may really use: affinity mask capped by the cgroup CPU quota"""
    n = len(os.sched_getaffinity(0))
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(float(q) / float(p) + 0.5)))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, int(q / p + 0.5)))
        except Exception:
            pass
    return n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--rows", type=int, default=4096, help="grid rows per GPU")
    ap.add_argument("--cols", type=int, default=4096)
    ap.add_argument("--days", type=int, default=365)
    ap.add_argument("--block-rows", type=int, default=512, help="rows per run_all launch")
    ap.add_argument("--flat", action="store_true", help="flat-terrain config (configs[1] style)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="CPU baseline budget; 0 = skip")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    # rehearsal knobs (not used by the driver): run N ranks on ONE GPU with gloo to exercise the
    # multi-rank control flow, e.g.  SPLASH_BENCH_BACKEND=gloo SPLASH_BENCH_ONE_GPU=1 torchrun ...
    backend = os.environ.get("SPLASH_BENCH_BACKEND", "nccl")
    if os.environ.get("SPLASH_BENCH_ONE_GPU") == "1":
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    import rsplash_amd
    if rank == 0:
        rsplash_amd.build_library()  # no-op when the in-tree .so is current
    if world > 1:
        dist.barrier()               # other ranks load the library only after rank 0 has built it
    from rsplash_amd import api

    H, W, nt = args.rows, args.cols, args.days
    n = H * W
    terrain = 0 if args.flat else 1
    cell0 = rank * n
    Htot = H * world
    dates = np.datetime64("2001-01-01") + np.arange(nt)
    cal = api.Calendar.from_dates(dates)
    nm = cal.nmonth
    f64 = dict(dtype=torch.float64, device=dev)
    stream = torch.cuda.current_stream()
    sp = stream.cuda_stream  # every launch goes on torch's current stream

    # ---------------- setup (untimed): inputs resident in HBM, K1, K4
    t_setup = time.time()
    lat, elev, slop, asp, res = (torch.empty(n, **f64) for _ in range(5))
    soil, au = torch.empty(6, n, **f64), torch.empty(3, n, **f64)
    api.dev_synth_statics(SEED, Htot, W, cell0, n, terrain, lat, elev, slop, asp, res, soil, au, sp)
    sw, tc, pn = (torch.empty(nt, n, **f64) for _ in range(3))
    api.dev_synth_forcing(SEED, cal, 0, nt, cell0, n, lat, elev, sw, tc, pn, n, sp)
    par = torch.empty(api.NPAR, n, **f64)
    Tt = torch.empty(n, **f64)
    state = torch.empty(api.NSTATE, n, **f64)
    api.dev_soil_hydro(n, lat, elev, slop, asp, res, soil, au, par, sp)
    api.dev_snow_threshold(n, nt, tc, n, par, n, Tt, sp)
    del soil, au, res
    work = torch.empty(api.spin_up_work_size(n), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    api.dev_spin_up(cal, n, sw, tc, pn, n, Tt, par, n, state, n, work, stream=sp)
    e1.record(stream)
    torch.cuda.synchronize()
    spinup_s = e0.elapsed_time(e1) / 1e3
    del work
    state0 = state.clone()

    bn = min(args.block_rows, H) * W                       # cells per run_all launch
    blocks = [(c0, min(c0 + bn, n)) for c0 in range(0, n, bn)]
    outb = {k: torch.empty(nt, bn, **f64) for k in OUT5}
    diag = torch.zeros(nm, api.NDIAG, **f64)
    dwork = torch.empty(max(api.diag_work_size(bn, nm), 8), dtype=torch.uint8, device=dev)
    setup_s = time.time() - t_setup

    ev = []  # (start, stop) events around every run_all launch of the timed steps

    def step(timed):
        state.copy_(state0)
        diag.zero_()
        api.dev_snow_threshold(n, nt, tc, n, par, n, Tt, sp)                 # K2
        for c0, c1 in blocks:                                                # K5 (+K7)
            if timed:
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(stream)
            api.dev_run_all(cal, 0, nt, c1 - c0, sw[:, c0:], tc[:, c0:], pn[:, c0:], n, Tt[c0:],
                            par[:, c0:], n, state[:, c0:], n, outb, bn, monthly=False,
                            diag=diag, diag_work=dwork, stream=sp)
            if timed:
                b.record(stream)
                ev.append((a, b, c1 - c0))
        if world > 1:  # global monthly water-balance totals, once per step (RCCL over xGMI)
            if backend == "nccl":
                dist.all_reduce(diag)
            else:
                h = diag.cpu()
                dist.all_reduce(h)
                diag.copy_(h)

    for _ in range(args.warmup):
        step(False)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    cell_days_step = float(n) * nt * world
    value = cell_days_step * args.steps / dt
    k_ms = np.array([a.elapsed_time(b) for a, b, _ in ev])
    k_cells = np.array([c for _, _, c in ev], dtype=np.float64)
    full = k_cells == k_cells.max()
    avg_launch_s = float(k_ms[full].mean()) / 1e3
    units_per_launch = float(k_cells.max()) * nt
    achieved = ALG_BYTES_PER_CELL_DAY * units_per_launch / avg_launch_s / 1e9
    traffic = valu_busy = None
    tfile = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tfile):
        try:
            tj = json.load(open(tfile))
            key = f"{min(args.block_rows, H)}x{W}x{nt}"
            traffic = tj.get(key, {}).get("hbm_bytes_per_launch")
            valu_busy = tj.get(key, {}).get("valu_busy_frac")
        except Exception:
            traffic = None

    try:
        metric = json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
    except Exception:
        metric = "cell-days/sec on 4096x4096x365d synthetic grid; max |dwn| vs reference"
    line = {
        "metric": metric,
        "value": value, "unit": "cell-days/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {
            "workload": f"{H}x{W} synthetic grid per GPU, {'flat' if args.flat else 'slope/aspect + lateral flow'}, "
                        f"{nt} days, daily outputs wn/snow/ro/aet/pet, row blocks of {min(args.block_rows, H)}",
            "grid_rows_per_gpu": H, "grid_cols": W, "days": nt, "block_rows": min(args.block_rows, H),
            "timed": "K2 snow threshold + K5 run_all (+ monthly diagnostics), inputs resident in HBM",
            "parallelism": f"row-block x{world}",
        },
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "frac_of_measured_copy_rate": achieved / HBM_COPY_GBS,
                     "kernel": "k_run_all<daily,diag>", "avg_launch_ms": avg_launch_s * 1e3,
                     "units_per_launch": units_per_launch,
                     "alg_bytes_per_unit": ALG_BYTES_PER_CELL_DAY,
                     "valu_busy_frac": valu_busy,
                     "note": "the launch is FP64-VALU-issue bound, not HBM bound (~1.9k vector "
                             "instructions per wavefront-day); valu_busy_frac = share of SIMD cycles "
                             "issuing vector instructions, from the PMC pass in profiles/; see DESIGN.md"},
        "spinup": {"seconds": spinup_s, "cells": n,
                   "note": "both spin-ups of every cell (2..1000 passes of 365 days each, data "
                           "dependent); untimed setup, not part of value"},
        "setup_s": setup_s,
    }

    # ---------------- CPU baseline + parity on a bounded sample (rank 0, N = 1 only)
    if rank == 0 and world == 1 and args.cpu_seconds > 0:
        from oracle.binding import Oracle
        o = Oracle()
        cores = host_cores()
        doy, year, month = api.time_fields(dates)

        def cpu_run(ns, threads):
            sl = slice(0, ns)
            hsw, htc, hpn = (x[:, sl].contiguous().cpu().numpy() for x in (sw, tc, pn))
            hp = par[:, sl].cpu().numpy()
            t1 = time.perf_counter()
            out, st_end = o.grid_transient(doy, year, month, hsw, htc, hpn, hp[0], hp[1], hp[2],
                                           hp[3], hp[4:16], Tt[sl].cpu().numpy(),
                                           state0[:, sl].cpu().numpy(), n_threads=threads)
            return time.perf_counter() - t1, out, o.fragile.copy()

        ns = min(4096, n)
        tp, _, _ = cpu_run(ns, cores)
        rate = ns * nt / tp
        ns2 = int(min(n, bn, max(ns, rate * args.cpu_seconds / nt)))
        ns2 -= ns2 % 64 if ns2 > 64 else 0
        tcpu, oref, frag = cpu_run(ns2, cores)
        # the same cells on the GPU, for the parity figures
        gs = {k: torch.empty(nt, ns2, **f64) for k in OUT5}
        st = state0[:, :ns2].clone()
        api.dev_run_all(cal, 0, nt, ns2, sw, tc, pn, n, Tt, par, n, st, ns2, gs, ns2, stream=sp)
        torch.cuda.synchronize()
        # cell-days on which the reference itself is ill-conditioned (sun grazing the slope:
        # DESIGN.md "Conditioning"); a cell is set aside from the strict check once it has one
        from oracle.binding import grazing_mask
        hp = par[:, :ns2].cpu().numpy()
        graz = grazing_mask(o, hp[0], hp[2], hp[3], doy, year)
        okc = ~graz.any(0) & (frag == 0) & (hp[8] >= 0.01)
        par_ok, frac_all = {}, {}
        for i, k in enumerate(OUT5):
            a, b = gs[k].cpu().numpy(), oref[i]
            err = np.abs(a - b)
            exc = np.maximum(err - 1e-9, 0.0)  # |x-ref| <= tol*|ref| + 1e-9
            rel = exc / np.maximum(np.abs(b), 1e-300)
            par_ok[k] = float(rel[:, okc].max())
            frac_all[k] = float((err <= 1e-5 * np.abs(b) + 1e-9).mean())
        dwn = np.abs(gs["wn"].cpu().numpy() - oref[0])
        line["cpu_baseline"] = {
            "value": ns2 * nt / tcpu, "unit": "cell-days/s", "cores": cores, "kind": "port",
            "sample": f"first {ns2} cells x {nt} days of the same grid (run_all + rain/snow split "
                      f"from the GPU's spun-up state), OpenMP over cells, {tcpu:.1f} s",
        }
        line["parity_sample"] = {
            "cells": ns2, "tolerance": 1e-5,
            "well_conditioned_cells": int(okc.sum()),
            "max_rel_err": par_ok, "max_abs_dwn_mm": float(dwn[:, okc].max()),
            "ill_conditioned_cells": {"grazing_sun": int(graz.any(0).sum()),
                                      "noise_drainage_denominator": int((frag > 0).sum()),
                                      "pore_size_index_below_0.01": int((hp[8] < 0.01).sum())},
            "fraction_of_all_cell_days_within_tolerance": frac_all,
        }
    if rank == 0:
        print(json.dumps(line))
    cal.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
