#!/usr/bin/env python3
"""bench.py -- cell-days/s of the gridded SPLASH loop on MI355X (BASELINE.json metric).

    python bench.py [--gpus N --steps K --warmup W] [--config 1|2|3|4]

`--gpus N` with N > 1 launches itself: the parent (which never touches a GPU) starts N fresh rank
processes -- one per device, RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* set, rendezvous on 127.0.0.1 --
and prints rank 0's JSON line.  Under `python -m torch.distributed.run ... bench.py --gpus N` the
ranks already exist and nothing is spawned.

Workloads (BASELINE.json configs; --config selects, default 2 = the configuration the metric is
quoted on):
  0  splash.point on the bundled Bourne SNOTEL site (2 922 days + both spin-ups): the call's latency on the GPU and
     with the host backend (ABI 6: "CPU reference path (plumbing, no GPU)"), checked against the oracle
  1  256 x 256 flat-terrain grid, 365 days, 1 GPU
  2  4096 x 4096 grid with slope/aspect/upslope area, 365 days.  With N ranks (`--scaling`):
       strong (default)  THE 4096 x 4096 grid of the metric, row blocks of `--chunk-rows` rows dealt
                         round-robin to the ranks (rank r owns blocks r, r+N, ...: every rank gets
                         every latitude band, so the ranks finish together)
       weak              every rank owns a 4096-row block of a (4096*N)-row grid
  3  30 years of daily forcing with spin-up, one rank's share (512 x 4096 cells = 1/8 of 4096^2)
     streamed one calendar year at a time through two forcing buffers (a 30-year raster cannot be
     resident: 11.8 TB for the whole grid), state carried in HBM
  4  0.5 degree global grid (360 x 720), ~26 % land by a hash mask (layout of data/SA_cru.RData:
     NA ocean cells), 30 years, monthly water-balance totals per basin, summed over ranks with ONE
     all-reduce of the exact integer limbs

One "step" = one pass of the hot path over the rank's cells with the inputs resident in HBM:
K2 (snowfall threshold pre-pass over tc) + K5 (SPLASH::run_all: rain/snow split, SOLAR, EVAP,
infiltration, runoff, lateral flow, soil-water update for every cell and day, daily wn, snow, ro,
aet, pet written to HBM) + K7 (monthly water-balance totals).  Setup outside the timed region:
on-device input generation, K1 soil_hydro and K4 the two spin-ups (timed separately:
`spinup.seconds`, and folded into `value_inclusive`).  Configs 3/4 also time the per-year input
generation, which stands in for the stream of forcing a real 30-year run would have to be fed.

The JSON line carries `roofline` (algorithmic 64 B per cell-day over the measured duration of the
run_all kernel, against 8 TB/s) and `cpu_baseline` (the plain-C port of the reference's loop in
oracle/, timed on this box's host cores on a bounded sample of the same workload, which is also
used to check the GPU results on that sample: `parity_sample`).
"""
import argparse
import json
import os
import subprocess
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


def cpu_model():
    try:
        for l in open("/proc/cpuinfo"):
            if l.startswith("model name"):
                return l.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--config", type=int, default=2, choices=(0, 1, 2, 3, 4))
    ap.add_argument("--rows", type=int, default=None, help="grid rows per GPU")
    ap.add_argument("--cols", type=int, default=None)
    ap.add_argument("--days", type=int, default=None)
    ap.add_argument("--years", type=int, default=30, help="configs 3/4")
    ap.add_argument("--block-rows", type=int, default=512, help="rows per run_all launch")
    ap.add_argument("--flat", action="store_true", help="flat terrain (implied by --config 1)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="CPU baseline budget; 0 = skip")
    ap.add_argument("--scaling", choices=("strong", "weak"), default="strong",
                    help="configs 1/2 with N > 1: split THE grid over the ranks, or one grid per rank")
    ap.add_argument("--chunk-rows", type=int, default=64, help="strong scaling: rows per dealt row block")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------ launcher
def launch_ranks(n, argv):
    """Start n rank processes of this script (fresh interpreters: the parent has not touched the
    GPU and never will), relay rank 0's JSON line, fail if any rank fails."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    out0, _ = procs[0].communicate()
    rcs = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    lines = [l for l in (out0 or "").splitlines() if l.startswith("{")]
    if any(rcs) or len(lines) != 1:
        sys.stderr.write(f"bench.py: rank exit codes {rcs}; rank 0 printed {len(lines)} JSON line(s)\n")
        sys.exit(1)
    print(lines[0])


# ------------------------------------------------------------------------------------ one rank
def main():
    args = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return launch_ranks(args.gpus, sys.argv[1:])
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    import torch
    import torch.distributed as dist
    if os.environ.get("SPLASH_BENCH_DRYRUN") == "1":
        # launcher rehearsal for the CPU test suite: rendezvous + one collective, no GPU, no compute
        if world > 1:
            dist.init_process_group("gloo")
        t = torch.tensor([float(rank + 1)], dtype=torch.float64)
        ctx_collective = None
        tab = None
        if world > 1:
            dist.all_reduce(t)
            ctx_collective = collective_info(dist, t.numel() * t.element_size())
            # the per-rank table of the real line (`ranks`), with pretend figures: row r = [100 (r + 1), r]
            tab = rank_table(world, rank, [100.0 * (rank + 1), float(rank)], torch, dist.all_reduce)
            dist.barrier()
            dist.destroy_process_group()
        if rank == 0:
            line = {"dryrun": True, "n_gpus": world, "sum_of_ranks": float(t.item())}
            if world > 1:
                line["collective"] = ctx_collective
                line["ranks"] = {"cells": [int(x) for x in tab[:, 0]], "rank": [int(x) for x in tab[:, 1]]}
            print(json.dumps(line))
        return
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    # rehearsal knobs (not used by the driver): N ranks on ONE GPU with gloo to exercise the
    # multi-rank control flow:  SPLASH_BENCH_BACKEND=gloo SPLASH_BENCH_ONE_GPU=1 python bench.py --gpus 2
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

    ctx = dict(args=args, world=world, rank=rank, dev=dev, backend=backend, api=api, torch=torch, dist=dist)
    if args.config == 0:
        line = run_point(ctx)
    else:
        line = run_resident(ctx) if args.config in (1, 2) else run_streamed(ctx)
    if rank == 0:
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


def rank_table(world, rank, row, torch, allreduce, device=None):
    """[world, len(row)] table with every rank's `row` in its own line: each rank fills its line of a zero
    table and ONE sum all-reduce puts them together (no all-gather needed; the values are doubles)"""
    t = torch.zeros(world, len(row), dtype=torch.float64, device=device)
    t[rank] = torch.tensor(row, dtype=torch.float64, device=device)
    if world > 1:
        allreduce(t)
    return t.cpu().numpy()


def collective_info(dist, nbytes):
    """what the process group itself says (the driver checks that the backend saw N ranks: `n_gpus` alone
    is WORLD_SIZE from the environment)"""
    return {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "rank0_of": dist.get_rank(),
            "op": "all_reduce(sum) of the int64 water-balance limbs, once per step", "bytes": int(nbytes)}


def allreduce_(ctx, t, op=None):
    """in-place all-reduce of a device tensor: rsplash_amd.dist.all_reduce_ (RCCL, or through the host
    with the gloo rehearsal) -- the same function tests/test_dist_cpu.py exercises with two ranks"""
    if ctx["world"] == 1:
        return t
    from rsplash_amd.dist import all_reduce_
    dist = ctx["dist"]
    return all_reduce_(t, "max" if op == dist.ReduceOp.MAX else "sum")


def timed_region(ctx, step, steps, warmup):
    """W untimed steps, then exactly K steps between barrier + synchronize; MAX over ranks"""
    torch, dist, world = ctx["torch"], ctx["dist"], ctx["world"]
    for _ in range(warmup):
        step(False)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step(True)
    torch.cuda.synchronize()
    ctx["own_step_s"] = (time.perf_counter() - t0) / max(steps, 1)   # this rank's own clock, before the closing barrier
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=ctx["dev"])
        allreduce_(ctx, tmax, dist.ReduceOp.MAX)
        dt = float(tmax.item())
    return dt


def base_line(ctx, value, steps, warmup, dt, workload, extra_cfg):
    try:
        metric = json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
    except Exception:
        metric = "cell-days/sec on 4096x4096x365d synthetic grid; max |dwn| vs reference"
    world = ctx["world"]
    cfg = {"workload": workload, "baseline_config": ctx["args"].config, "parallelism": f"row-block x{world}"}
    cfg.update(extra_cfg)
    return {"metric": metric, "value": value, "unit": "cell-days/s", "n_gpus": world, "steps": steps,
            "warmup": warmup, "ms_per_step": dt / steps * 1e3, "higher_is_better": True,
            "scaling": extra_cfg.get("scaling", "weak"), "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": cfg}


def kernel_sources_sha16(root=None):
    """sha256 (first 16 hex digits) of the kernels' sources WITHOUT comments and white space: the PMC figures of
    profiles/traffic.json carry the hash of the sources they were measured on, so a constant that went stale with a
    kernel change shows in the line (and an edited comment does not raise a false alarm)"""
    import hashlib
    import re
    root = root or ROOT
    h = hashlib.sha256()
    # the translation unit of the day-loop kernels and what it includes -- not the host side of the library
    names = ("splash_kernels.hip", "splash_kernels.hpp", "splash_device.hpp", "splash_math.hpp", "splash_log_table.hpp")
    for f in [os.path.join(root, "rsplash_amd", "csrc", n) for n in names]:
        txt = open(f, encoding="utf-8", errors="replace").read()
        txt = re.sub(r"/\*.*?\*/", " ", txt, flags=re.S)      # block comments
        txt = re.sub(r"//[^\n]*", " ", txt)                    # line comments (no string of these files holds "//")
        txt = re.sub(r"\s+", "", txt)
        h.update(os.path.basename(f).encode())
        h.update(txt.encode())
    return h.hexdigest()[:16]


def roofline(avg_launch_s, units_per_launch, key, kernel):
    achieved = ALG_BYTES_PER_CELL_DAY * units_per_launch / avg_launch_s / 1e9
    traffic = valu_busy = None
    stamp = None
    tfile = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tfile):
        try:
            tj = json.load(open(tfile))
            traffic = tj.get(key, {}).get("hbm_bytes_per_launch")
            valu_busy = tj.get(key, {}).get("valu_busy_frac")
            stamp = tj.get(key, {}).get("kernel_sources_sha16")
        except Exception:
            traffic = None
    now = kernel_sources_sha16()
    return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
            "traffic_measured_on_these_kernel_sources": None if traffic is None else bool(stamp == now),
            "kernel_sources_sha16": now,
            "traffic_source": None if traffic is None else
            "profiles/traffic.json: separate rocprofv3 --pmc passes of this command on an earlier run "
            "(tools/profile_round.sh), not measured in this run; so is valu_busy_frac.  The file carries the hash of "
            "the kernel sources it was measured on (traffic_measured_on_these_kernel_sources)",
            "frac_of_measured_copy_rate": achieved / HBM_COPY_GBS, "kernel": kernel,
            "avg_launch_ms": avg_launch_s * 1e3, "units_per_launch": units_per_launch,
            "alg_bytes_per_unit": ALG_BYTES_PER_CELL_DAY, "valu_busy_frac": valu_busy,
            "note": "the launch is FP64-VALU-issue bound, not HBM bound (vector instructions per "
                    "wavefront-day and valu_busy_frac from the PMC pass in profiles/; DESIGN.md)"}


# ------------------------------------------------------------------------------------ config 0
def run_point(ctx):
    """BASELINE configs[0]: splash.point on the bundled Bourne site (tests/golden: decoded from data/Bourne.RData),
    the reference's README example (README.md:28-61).  One cell: a latency, not a throughput -- reported as cell-days/s of
    the call on the GPU (`value`) and with the host backend, both checked against the oracle."""
    import csv
    import datetime
    api = ctx["api"]
    from oracle.binding import Oracle
    gold = os.path.join(ROOT, "tests", "golden")
    rows = list(csv.DictReader(open(os.path.join(gold, "bourne_forcing.csv"))))
    site = json.load(open(os.path.join(gold, "bourne_site.json")))
    dates = np.array([datetime.date.fromisoformat(r["date"]) for r in rows], dtype="datetime64[D]")
    P, Ta, sw = (np.array([float(r[k]) for r in rows]) for k in ("P", "Ta", "sw_in"))
    soil = np.array([np.nan if v is None else v for v in site["soil"]])
    nt = len(dates)
    steps = ctx["args"].steps if ctx["args"].steps is not None else 10
    warmup = ctx["args"].warmup if ctx["args"].warmup is not None else 2
    call = lambda backend: api.splash_point(sw, Ta, P, site["latitude"], site["elev_m"], site["slop_250m"], site["asp_250m"],
                                            soil, site["Aups_250m"], site["resolution"], time_index=dates, backend=backend)
    res, secs = {}, {}
    for backend in ("hip", "host"):
        for _ in range(warmup):
            call(backend)
        t0 = time.perf_counter()
        for _ in range(steps):
            res[backend] = call(backend)
        secs[backend] = (time.perf_counter() - t0) / steps
    o = Oracle()
    doy, year, month = api.time_fields(dates)
    mid = api.month_ids(year, month)
    t0 = time.perf_counter()
    ref, aux = o.point(doy, year, month, mid, int(mid[-1]) + 1, sw, Ta, P, site["latitude"], site["elev_m"], site["slop_250m"],
                       site["asp_250m"], soil, [site["Aups_250m"], 3, 3], site["resolution"])
    t_orc = time.perf_counter() - t0
    rel = {b: max(float(np.max(np.abs(res[b][k] - ref[k]) / np.maximum(np.abs(ref[k]), 1e-6))) for k in OUT5) for b in res}
    line = base_line(ctx, nt / secs["hip"], steps, warmup, secs["hip"] * steps,
                     f"splash.point on the bundled Bourne SNOTEL site, {nt} days (+ both spin-ups), one cell: the call's latency",
                     {"scaling": "weak", "days": nt, "timed": "the whole splash_point_run call (host buffers in and out)"})
    line["config"].pop("scaling", None)
    line["latency_ms"] = {"gpu": secs["hip"] * 1e3, "host_backend": secs["host"] * 1e3, "oracle_plain_c_one_core": t_orc * 1e3}
    line["value_host_backend"] = nt / secs["host"]
    line["max_rel_err_vs_oracle"] = rel
    line["roofline"] = None
    line["cpu_baseline"] = {"value": nt / t_orc, "unit": "cell-days/s", "cores": 1, "kind": "port",
                            "sample": "the same call through the oracle's plain-C restatement, one core, one run"}
    assert max(rel.values()) < 1e-5, rel
    return line


# ------------------------------------------------------------------------------------ configs 1, 2
def run_resident(ctx):
    args, world, rank, dev, api, torch = (ctx[k] for k in ("args", "world", "rank", "dev", "api", "torch"))
    flat = args.flat or args.config == 1
    H = args.rows or (256 if args.config == 1 else 4096)
    W = args.cols or (256 if args.config == 1 else 4096)
    nt = args.days or 365
    steps = args.steps if args.steps is not None else 3
    warmup = args.warmup if args.warmup is not None else 1
    terrain = 0 if flat else 1
    strong = args.scaling == "strong" and world > 1
    if strong:
        # THE H x W grid, cut into row blocks of chunk_rows rows that are dealt round-robin
        from rsplash_amd.dist import row_blocks_round_robin
        Htot = H
        cr_eff = max(1, min(args.chunk_rows, H))
        chunks = row_blocks_round_robin(H, cr_eff, world, rank)
    else:
        Htot = H * world
        chunks = [(rank * H, H)]
        cr_eff = H
    n = sum(nr for _, nr in chunks) * W
    total_cells = float(Htot) * W
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
    sw, tc, pn = (torch.empty(nt, n, **f64) for _ in range(3))
    off = 0
    for r0, nr in chunks:  # counter-based generators: any range of cells of the grid, anywhere
        nc = nr * W
        if len(chunks) == 1:
            api.dev_synth_statics(SEED, Htot, W, r0 * W, nc, terrain, lat, elev, slop, asp, res, soil, au, sp)
        else:
            t5 = [torch.empty(nc, **f64) for _ in range(5)]
            tso, tau = torch.empty(6, nc, **f64), torch.empty(3, nc, **f64)
            api.dev_synth_statics(SEED, Htot, W, r0 * W, nc, terrain, *t5, tso, tau, sp)
            for dst, src in zip((lat, elev, slop, asp, res), t5):
                dst[off:off + nc] = src
            soil[:, off:off + nc] = tso
            au[:, off:off + nc] = tau
        api.dev_synth_forcing(SEED, cal, 0, nt, r0 * W, nc, lat[off:], elev[off:], sw[:, off:], tc[:, off:],
                              pn[:, off:], n, sp)
        off += nc
    par = torch.empty(api.NPAR, n, **f64)
    Tt = torch.empty(n, **f64)
    state = torch.empty(api.NSTATE, n, **f64)
    api.dev_soil_hydro(n, lat, elev, slop, asp, res, soil, au, par, sp)
    api.dev_snow_threshold(n, nt, tc, n, par, n, Tt, sp)
    del soil, au, res
    par0_cellout = par[15].clone()  # the spin-up overwrites ncellout with the aridity index
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

    bn = min(args.block_rows, n // W) * W                  # cells per run_all launch
    blocks = [(c0, min(c0 + bn, n)) for c0 in range(0, n, bn)]
    outb = {k: torch.empty(nt, bn, **f64) for k in OUT5}
    acc = torch.zeros(1, nm, api.DIAG_REC, dtype=torch.int64, device=dev)
    diag = torch.zeros(1, nm, api.NDIAG, **f64)
    setup_s = time.time() - t_setup
    ev = []  # (start, stop) events around every run_all launch of the timed steps

    def step(timed):
        state.copy_(state0)
        acc.zero_()
        api.dev_snow_threshold(n, nt, tc, n, par, n, Tt, sp)                 # K2
        for c0, c1 in blocks:                                                # K5 (+K7)
            if timed:
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(stream)
            api.dev_run_all(cal, 0, nt, c1 - c0, sw[:, c0:], tc[:, c0:], pn[:, c0:], n, Tt[c0:],
                            par[:, c0:], n, state[:, c0:], n, outb, bn, monthly=False,
                            diag_acc=acc, stream=sp)
            if timed:
                b.record(stream)
                ev.append((a, b, c1 - c0))
        # global monthly water-balance totals, once per step: exact integer limbs, so the sum over
        # ranks (RCCL over xGMI) does not depend on the reduction order
        allreduce_(ctx, acc)
        api.dev_diag_finalize(acc, 1, nm, diag, stream=sp)

    dt = timed_region(ctx, step, steps, warmup)
    cell_days_step = total_cells * nt
    value = cell_days_step * steps / dt
    k_ms = np.array([a.elapsed_time(b) for a, b, _ in ev])
    k_cells = np.array([c for _, _, c in ev], dtype=np.float64)
    full = k_cells == k_cells.max()
    avg_launch_s = float(k_ms[full].mean()) / 1e3
    units_per_launch = float(k_cells.max()) * nt

    brows = min(args.block_rows, n // W)
    kind = 'flat' if flat else 'slope/aspect + lateral flow'
    if world == 1:
        split = "one GPU"
        mode = args.scaling  # what --gpus N would do with the same flags: the label is the same at every N
    elif strong:
        split = (f"split over {world} GPUs: row blocks of {cr_eff} rows dealt round-robin, {n // W} rows "
                 f"({n} cells) per GPU")
        mode = "strong"
    else:
        split = f"one {H}x{W} block per GPU ({world} GPUs: a {Htot}x{W} grid in all)"
        mode = "weak"
    line = base_line(ctx, value, steps, warmup, dt,
                     f"{Htot}x{W} synthetic grid in total, {split}; {kind}, {nt} days, daily outputs "
                     f"wn/snow/ro/aet/pet, run_all launches of {brows} rows",
                     {"scaling": mode, "grid_rows_total": Htot, "grid_rows_per_gpu": n // W, "grid_cols": W,
                      "days": nt, "block_rows": brows,
                      "timed": "K2 snow threshold + K5 run_all + K7 monthly water-balance totals "
                               "(+ all-reduce), inputs resident in HBM"})
    line["config"].pop("scaling", None)
    if world > 1:
        line["collective"] = collective_info(ctx["dist"], acc.numel() * acc.element_size())
    line["roofline"] = roofline(avg_launch_s, units_per_launch, f"{brows}x{W}x{nt}", "k_run_all<daily,diag>")
    # both spin-ups of every cell + one transient pass: what a whole splash.grid call costs.
    # Measured as the library runs it for a host caller: the spin-up's long tail (a few cells per
    # thousand, over a second of sequential day steps on a handful of wavefronts) goes to a second
    # stream while the first one already runs the transient pass of every other cell; the late
    # cells follow in list mode into compact arrays.
    work = torch.empty(api.spin_up_work_size(n), dtype=torch.uint8, device=dev)
    mask_p, cells_p, ncells_p = api.dev_spin_up_pending(work, n)
    late_cap = max(4096, n // 64)
    late = {k: torch.empty(nt, late_cap, **f64) for k in OUT5}  # noqa
    s2 = torch.cuda.Stream(device=dev, priority=-1)   # the tail's stream: its few workgroups are dispatched first
    par_i = par.clone()
    par_i[15] = par0_cellout
    torch.cuda.synchronize()
    if world > 1:
        ctx["dist"].barrier()
    t0 = time.perf_counter()
    acc.zero_()
    api.dev_spin_up_begin(cal, n, sw, tc, pn, n, Tt, par_i, n, state, n, work, stream=sp)
    e_begin = torch.cuda.Event()
    e_begin.record(stream)
    for c0, c1 in blocks:
        api.dev_run_all(cal, 0, nt, c1 - c0, sw[:, c0:], tc[:, c0:], pn[:, c0:], n, Tt[c0:],
                        par_i[:, c0:], n, state[:, c0:], n, outb, bn, monthly=False, diag_acc=acc,
                        stream=sp, skip=mask_p + c0)
    s2.wait_event(e_begin)
    e_t0, e_t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e_t0.record(s2)
    api.dev_spin_up_finish(cal, n, sw, tc, pn, n, Tt, par_i, n, state, n, work, stream=s2.cuda_stream)
    e_t1.record(s2)
    torch.cuda.synchronize()
    tail_s = e_t0.elapsed_time(e_t1) / 1e3
    n_late = api.dev_spin_up_pending_count(work, n, sp)
    for k0 in range(0, n_late, late_cap):
        nl = min(late_cap, n_late - k0)
        api.dev_run_all(cal, 0, nt, nl, sw, tc, pn, n, Tt, par_i, n, state, n, late, late_cap,
                        monthly=False, diag_acc=acc, stream=sp, cells=cells_p + 4 * k0)
    allreduce_(ctx, acc)
    api.dev_diag_finalize(acc, 1, nm, diag, stream=sp)
    torch.cuda.synchronize()
    if world > 1:
        ctx["dist"].barrier()
    incl = time.perf_counter() - t0
    incl_own = incl
    incl_t = torch.tensor([incl], dtype=torch.float64, device=dev)
    allreduce_(ctx, incl_t, ctx["dist"].ReduceOp.MAX)
    incl = float(incl_t.item())
    del work, late
    sp_all = torch.tensor([spinup_s], dtype=torch.float64, device=dev)
    allreduce_(ctx, sp_all, ctx["dist"].ReduceOp.MAX)
    spinup_max = float(sp_all.item())
    line["value_inclusive"] = cell_days_step / incl
    tail_t = torch.tensor([tail_s], dtype=torch.float64, device=dev)
    allreduce_(ctx, tail_t, ctx["dist"].ReduceOp.MAX)
    # The whole-call figures also sit in `config`, which the driver's record keeps whole (round 5: they were
    # printed behind a 9 KB parity block and fell off the 8 KB tail the record keeps of the line).
    line["config"].update({"value_inclusive": cell_days_step / incl, "inclusive_seconds": incl,
                           "inclusive_tail_seconds": float(tail_t.item()), "spinup_seconds": spinup_max,
                           "late_cells": int(n_late)})
    # What every rank saw (one more all-reduce of a world x 5 table, each rank filling its row): cells, the
    # rank's own wall time per step, its spin-up alone, its tail, its whole call -- so that the first record on
    # an 8-GPU node shows imbalance and the constant tail without another round.
    rk = rank_table(world, rank, [float(n), ctx.get("own_step_s", dt / steps) * 1e3, float(k_ms.sum()) / steps, spinup_s,
                                  tail_s, incl_own], torch, lambda t: allreduce_(ctx, t), dev)
    line["ranks"] = {"cells": [int(x) for x in rk[:, 0]], "ms_per_step": [round(float(x), 3) for x in rk[:, 1]],
                     "run_all_ms_per_step": [round(float(x), 3) for x in rk[:, 2]],
                     "spinup_seconds": [round(float(x), 4) for x in rk[:, 3]],
                     "tail_seconds": [round(float(x), 4) for x in rk[:, 4]],
                     "inclusive_seconds": [round(float(x), 4) for x in rk[:, 5]],
                     "ms_per_step_min_max": [float(rk[:, 1].min()), float(rk[:, 1].max())],
                     "note": "per rank, in rank order: ms_per_step = the rank's own clock up to (not including) the closing "
                             "barrier (the steps' all-reduce couples the ranks: look at run_all_ms_per_step, the sum of "
                             "the rank's own k_run_all launches per step, for imbalance)"}
    line["inclusive"] = {"seconds": incl, "late_cells": n_late, "tail_seconds": float(tail_t.item()),
                         "tail_note": "the spin-up's last phase (splash_dev_spin_up_finish): one cell's chain of up to "
                                      "1000 passes x 365 sequential day steps on a lone wavefront; constant PER RANK, "
                                      "it does not shrink when the grid is split over more GPUs (DESIGN.md 7)",
                         "note": "spin-up + one transient pass of every cell with outputs, the spin-up's tail on "
                                 "a second stream overlapped with the transient pass (as splash_grid_run does it)"}
    line["spinup"] = {"seconds": spinup_max, "cells": n,
                      "note": "both spin-ups of every cell (2..1000 passes of 365 days each, data "
                              "dependent) run alone on one stream, untimed setup of the steps above"}
    line["setup_s"] = setup_s
    line["water_balance"] = {"annual_precip_mm_per_cell": float(diag[0, :, 0].sum().item()) / total_cells,
                             "annual_aet_mm_per_cell": float(diag[0, :, 1].sum().item()) / total_cells,
                             "annual_runoff_mm_per_cell": float(diag[0, :, 2].sum().item()) / total_cells}

    # rank 0 (at any N): CPU baseline on a bounded sample of ITS cells + the parity figures of the same
    # sample; the other ranks wait at the barrier so that the group is torn down together
    if rank == 0 and args.cpu_seconds > 0:
        cpu_leg(ctx, line, cal, dates, sw, tc, pn, par, Tt, state0, n, nt, bn, sp, W, chunks, Htot, par0_cellout)
    if world > 1:
        ctx["dist"].barrier()
    cal.close()
    return line


def sample_bands(n, W, cells_wanted, nbands=8):
    """[(first local cell, cells)] of `nbands` bands of whole grid rows spread evenly over the rank's rows
    (first band = its first rows, last band = its last rows), cells_wanted cells in all"""
    rows = n // W
    rb = max(1, cells_wanted // (nbands * W))
    if rows < 2 * nbands * rb or n % W:
        return [(0, min(n, max(W, cells_wanted - cells_wanted % 64)))]
    return [(int(round(i * (rows - rb) / (nbands - 1))) * W, rb * W) for i in range(nbands)]


def cpu_leg(ctx, line, cal, dates, sw, tc, pn, par, Tt, state0, n, nt, bn, sp, W, chunks, Htot, par0_cellout):
    """CPU baseline (the oracle's transient loop on the host cores) on a bounded sample of the same
    workload, and the parity figures of the GPU results on that sample (rank 0 only).  The sample is
    eight bands of whole rows spread over the rank's rows -- every latitude band of the grid, not its
    first rows (rows 0-127 of 4096 said nothing about the snow-free, ponding-prone low
    latitudes) -- and the same cells serve the timing and the comparison."""
    api, torch, args = ctx["api"], ctx["torch"], ctx["args"]
    from oracle.binding import Oracle
    from oracle.parity import SERIES11, merge_profiles, trajectory_summary
    o = Oracle()
    cores = host_cores()
    doy, year, month = api.time_fields(dates)

    def host(cells):   # cells: list of (c0, nc) local ranges -> numpy inputs of those cells, concatenated
        cat = lambda x: torch.cat([x[..., c0:c0 + nc] for c0, nc in cells], dim=-1).contiguous().cpu().numpy()
        return cat(sw), cat(tc), cat(pn), cat(par), cat(Tt), cat(state0)

    def cpu_run(cells, threads, store=True):
        hsw, htc, hpn, hp, hTt, hst = host(cells)
        t1 = time.perf_counter()
        o.grid_transient(doy, year, month, hsw, htc, hpn, hp[0], hp[1], hp[2], hp[3], hp[4:16], hTt, hst,
                         n_threads=threads, store=store)
        return time.perf_counter() - t1

    def global_row(c_local):   # grid row of a local cell (strong scaling deals row blocks round-robin)
        r, acc_rows = c_local // W, 0
        for r0, nr in chunks:
            if r < acc_rows + nr:
                return r0 + (r - acc_rows)
            acc_rows += nr
        return -1

    ns1 = min(1024, n)
    t1 = cpu_run([(0, ns1)], 1)
    rate1 = ns1 * nt / t1
    nsn = min(n, max(ns1, 64 * cores))
    tn = cpu_run([(0, nsn)], cores)
    rate_n = nsn * nt / tn
    # at most 2^19 cells, so that the sample -- and what it reports -- is the same on any host whose
    # cores get that far in the time budget
    want = int(min(n, 1 << 19, max(nsn, rate_n * args.cpu_seconds / nt)))
    bands = sample_bands(n, W, want)
    ns2 = sum(nc for _, nc in bands)
    tcpu = cpu_run(bands, cores)
    rows_txt = ", ".join(f"{global_row(c0)}-{global_row(c0 + nc - 1)}" for c0, nc in bands)
    line["cpu_baseline"] = {
        "value": ns2 * nt / tcpu, "unit": "cell-days/s", "cores": cores, "kind": "port",
        "single_thread_value": rate1, "threads_obtained": o.threads_obtained(cores), "omp_max_threads": o.max_threads(), "cpu_model": cpu_model(),
        "sample": f"{ns2} cells x {nt} days of the same grid in {len(bands)} bands of whole rows spread over the "
                  f"latitudes (grid rows {rows_txt}; run_all + rain/snow split from the GPU's spun-up state), OpenMP "
                  f"over cells with {cores} threads, {tcpu:.1f} s; one thread: {ns1} cells in {t1:.1f} s",
    }
    # the same cells on the GPU, all 11 series, checked on EVERY cell-day (oracle/parity.py), band by band
    f64 = dict(dtype=torch.float64, device=ctx["dev"])
    per_band, profs, viol, worst = [], [], [], []
    tot = dict(cell_days=0, set_aside_cell_days=0, grazing_cell_days=0, cells_after_a_grazing_day=0,
               critical_event_cells=0, critical_event_cells_that_differ=0,
               cells_on_which_the_reference_disagrees_with_itself=0)
    max_rel, max_dwn, oot = {}, 0.0, {}
    frac_all = {}
    for c0, nc in bands:
        gs = {k: torch.empty(nt, nc, **f64) for k in SERIES11}
        st = state0[:, c0:c0 + nc].clone()
        api.dev_run_all(cal, 0, nt, nc, sw[:, c0:], tc[:, c0:], pn[:, c0:], n, Tt[c0:], par[:, c0:], n, st, nc, gs, nc,
                        stream=sp)
        torch.cuda.synchronize()
        hsw, htc, hpn, hp, hTt, hst = host([(c0, nc)])
        p = dict(doy=doy, year=year, month=month, sw=hsw, tc=htc, pn=hpn, par=hp, Tt=hTt, state0=hst,
                 out={k: v.cpu().numpy() for k, v in gs.items()})
        r = trajectory_summary(o, p, n_threads=cores, n_worst=5)
        row0 = global_row(c0)
        per_band.append({"row0": row0, "rows": nc // W, "lat_deg": round(60.0 - 120.0 * (row0 + 0.5 * (nc // W)) / Htot, 2),
                         "cells": nc, "set_aside_cell_days": r["set_aside_cell_days"],
                         "critical_event_cells": r["critical_event_cells"],
                         "flips": r["critical_event_cells_that_differ"], "max_rel_err": r["max_rel_err"],
                         "max_abs_dwn_mm": r["max_abs_dwn_mm"], "out_of_tolerance": r["out_of_tolerance"],
                         "violations": r["violations"]})
        for k in tot:
            tot[k] += r[k]
        for k, v in r["max_rel_err"].items():
            max_rel[k] = max(max_rel.get(k, 0.0), v)
        for k, v in r["out_of_tolerance"].items():
            oot[k] = {q: oot.get(k, {}).get(q, 0) + v[q] for q in v}
        for k, v in r["fraction_of_all_cell_days_within_tolerance"].items():
            frac_all[k] = frac_all.get(k, 0.0) + v * nc / ns2
        max_dwn = max(max_dwn, r["max_abs_dwn_mm"])
        profs.append(r["profile"])
        viol += [f"rows {row0}+: {v}" for v in r["violations"]]
        for w in r["worst"]:
            w = dict(w, grid_row=global_row(c0 + w["cell"]), grid_col=int((c0 + w["cell"]) % W))
            worst.append(w)
        excuse = r["excuse_builds"]
        del gs, st
    worst.sort(key=lambda w: -w["rel"])
    profile = {k: merge_profiles([pr[k] for pr in profs]) for k in profs[0]}
    full = dict(tot, cells=ns2, tolerance=1e-5, bands=per_band, max_rel_err=max_rel, max_abs_dwn_mm=max_dwn,
                fraction_of_all_cell_days_within_tolerance=frac_all, out_of_tolerance=oot, violations=viol,
                excuse_builds=excuse, worst=worst[:5], profile=profile)
    # The line carries a COMPACT form (the driver's record keeps 8 KB of it; round 5's 9 KB block pushed the
    # whole-call figures out of the record); the full per-band / per-decade detail goes to stderr as one JSON line.
    sys.stderr.write("PARITY_DETAIL " + json.dumps(full) + "\n")
    r3 = lambda x: float(f"{x:.3g}")
    line["parity_sample"] = dict(
        tot, cells=ns2, tolerance=1e-5, max_rel_err={k: r3(v) for k, v in max_rel.items()}, max_abs_dwn_mm=r3(max_dwn),
        out_of_tolerance={k: [v["all"], v["inside_set_aside"], v["kept"]] for k, v in oot.items()},
        out_of_tolerance_columns=["all", "inside_set_aside", "kept"],
        min_fraction_of_all_cell_days_within_tolerance=r3(min(frac_all.values())),
        beyond_1e_9_1e_7={k: [profile[k]["beyond_1e_9"], profile[k]["beyond_1e_7"]] for k in profile},
        n_violations=len(viol), violations=[v[:160] for v in viol[:2]], excuse_builds=excuse,
        bands={"row0": [b["row0"] for b in per_band], "lat_deg": [b["lat_deg"] for b in per_band],
               "set_aside_cell_days": [b["set_aside_cell_days"] for b in per_band],
               "critical_event_cells": [b["critical_event_cells"] for b in per_band],
               "flips": [b["flips"] for b in per_band],
               "max_rel_err": [r3(max(b["max_rel_err"].values())) for b in per_band],
               "kept_out_of_tolerance": [sum(v["kept"] for v in b["out_of_tolerance"].values()) for b in per_band]},
        worst=[{"series": w["series"], "row": w["grid_row"], "col": w["grid_col"], "day": w["day"], "rel": r3(w["rel"]),
                "events": w["events"]} for w in worst[:3]],
        note="trajectory comparison from the GPU's spun-up state on every cell-day of eight bands of rows (oracle/parity.py); "
             "kept out-of-tolerance must be 0 or excused; per-band and per-decade detail: stderr, PARITY_DETAIL")
    # ---- the spin-up's RESULT against the oracle's own brute-force spin-up (VERDICT r05 #1c): one row of every band,
    # each side from its own spin-up (the comparison above starts the oracle from the GPU's spun-up state)
    t_sp = time.perf_counter()
    from oracle.parity import spinup_summary
    terrain = 0 if (args.flat or args.config == 1) else 1
    sel = [(c0, min(W, nc)) for c0, nc in bands]
    ns3 = sum(nc for _, nc in sel)
    mid = api.month_ids(year, month)
    nmon = int(mid[-1]) + 1
    hsw, htc, hpn, hp, hTt, hst = host(sel)
    st_h = {}
    for c0, nc in sel:
        one = o.synth_statics(SEED, Htot, W, global_row(c0) * W + c0 % W, nc, terrain)
        for k, v in one.items():
            st_h.setdefault(k, []).append(v)
    st_h = {k: np.concatenate(v, axis=-1) for k, v in st_h.items()}
    _, aux = o.grid(doy, year, month, mid, nmon, hsw, htc, hpn, st_h["lat"], st_h["elev"], st_h["slop"], st_h["asp"],
                    st_h["soil"], st_h["au"], st_h["resolution"], n_threads=cores)
    # the product's spin-up of the same cells, run again on their own WITH the pass counts (the grid run does not
    # count the passes of spin-up #1); its state and aridity index must be the grid run's, bit for bit
    cat = lambda x: torch.cat([x[..., c0:c0 + nc] for c0, nc in sel], dim=-1).contiguous()
    d_sw, d_tc, d_pn, d_Tt = cat(sw), cat(tc), cat(pn), cat(Tt)
    d_par = cat(par).clone()                 # the grid run's own cell parameters (the device generator's statics) ...
    d_par[15] = cat(par0_cellout)            # ... with ncellout as it was before the spin-up wrote the aridity index there
    d_state = torch.empty(api.NSTATE, ns3, **f64)
    d_pass = torch.zeros(2, ns3, dtype=torch.int32, device=ctx["dev"])
    d_work = torch.empty(api.spin_up_work_size(ns3), dtype=torch.uint8, device=ctx["dev"])
    api.dev_spin_up(cal, ns3, d_sw, d_tc, d_pn, ns3, d_Tt, d_par, ns3, d_state, ns3, d_work, passes=d_pass, stream=sp)
    torch.cuda.synchronize()
    same_bits = bool(torch.equal(d_state.view(torch.int64), cat(state0).view(torch.int64)) and
                     torch.equal(d_par[15].view(torch.int64), cat(par)[15].view(torch.int64)))
    rs = spinup_summary(o, d_state.cpu().numpy(), d_par[15].cpu().numpy(), d_pass[1].cpu().numpy(), aux, st_h,
                        (doy, year, month, mid, nmon, hsw, htc, hpn))
    line["spinup_parity"] = dict(
        cells=rs["cells"], capped=rs["capped"], capped_within_tolerance=rs["capped_within_tolerance"],
        passes_equal=rs["passes_equal"], nd_equal=rs["nd_equal"],
        max_rel_state=max(rs["max_rel_state"].values()), max_rel_state_capped=rs["max_rel_state_capped"],
        max_rel_aridity_index=rs["max_rel_ai"], critical_event_cells=rs["critical_event_cells"],
        critical_event_cells_that_differ=rs["critical_event_cells_that_differ"], excused=rs["excused"],
        grazing_cells_set_aside=rs["grazing_cells"], same_bits_as_the_grid_run=same_bits, n_violations=len(rs["violations"]),
        violations=[v[:200] for v in rs["violations"][:2]], seconds=round(time.perf_counter() - t_sp, 1),
        rows=[global_row(c0) for c0, _ in sel],
        note="state handed to run_all (wn, snow, qin, td within 1e-5; nd exactly), aridity index and pass counts of one "
             "row per band, the product's phased spin-up with exact-cycle detection against the oracle's brute-force "
             "1000-pass loop (src/SPLASH.cpp:1594-1749, R/splash.point.R:141-152), each side from its own spin-up")


# ------------------------------------------------------------------------------------ configs 3, 4
def run_streamed(ctx):
    """30-year runs: spin-up on the first 365 days, then the transient loop one calendar year at a
    time through two forcing buffers filled by the device generator on a second stream (the stand-in
    for the forcing stream of a real run), the 5-double state carried in HBM, daily outputs of each
    year written to one reused buffer (config 3) or monthly outputs + basin totals (config 4)."""
    args, world, rank, dev, api, torch, dist = (ctx[k] for k in ("args", "world", "rank", "dev", "api",
                                                                 "torch", "dist"))
    cfg4 = args.config == 4
    steps = args.steps if args.steps is not None else 1
    warmup = args.warmup if args.warmup is not None else 0
    years = args.years
    dates = np.arange(np.datetime64("2001-01-01"), np.datetime64(f"{2001 + years}-01-01"))
    nt = len(dates)
    cal = api.Calendar.from_dates(dates)
    doy, year, month = api.time_fields(dates)
    ybounds = [0] + [int(i) + 1 for i in np.where(np.diff(year) != 0)[0]] + [nt]
    nm = cal.nmonth
    f64 = dict(dtype=torch.float64, device=dev)
    stream = torch.cuda.current_stream()
    sp = stream.cuda_stream
    gen_stream = torch.cuda.Stream(device=dev)

    if cfg4:
        Htot, W = args.rows or 360, args.cols or 720
        # land mask by hash (~26 % valid); rows cut so that every rank gets the same land count
        from rsplash_amd.dist import row_block_partition
        rng = np.random.default_rng(SEED)
        land_all = rng.random(Htot * W) < 0.26
        part = row_block_partition(Htot, W, world, rank, land_all.reshape(Htot, W).sum(1))
        cell0, nfull = part["cell0"], part["ncell"]
        land_idx = np.where(land_all[cell0:cell0 + nfull])[0]
        n = len(land_idx)
        # basins: 30 x 30 degree tiles (6 x 12 = 72)
        rr, cc = np.divmod(cell0 + land_idx, W)
        basin_h = ((rr * 6 // Htot) * 12 + cc * 12 // W).astype(np.int32)
        n_basin = 72
        terrain = 1
    else:
        H, W = args.rows or 512, args.cols or 4096
        Htot = H * world
        n = nfull = H * W
        cell0 = rank * n
        land_idx, n_basin, terrain = None, 1, 1

    # statics (generated for the contiguous range, compacted to the land cells for config 4)
    stat = [torch.empty(nfull, **f64) for _ in range(5)]
    soil_f, au_f = torch.empty(6, nfull, **f64), torch.empty(3, nfull, **f64)
    api.dev_synth_statics(SEED, Htot, W, cell0, nfull, terrain, *stat, soil_f, au_f, sp)
    if cfg4:
        li = torch.tensor(land_idx, device=dev)
        lat, elev, slop, asp, res = (x[li].contiguous() for x in stat)
        soil, au = soil_f[:, li].contiguous(), au_f[:, li].contiguous()
        basin = torch.tensor(basin_h, device=dev)
    else:
        lat, elev, slop, asp, res = stat
        soil, au, basin = soil_f, au_f, None
    par = torch.empty(api.NPAR, n, **f64)
    api.dev_soil_hydro(n, lat, elev, slop, asp, res, soil, au, par, sp)
    lat_f, elev_f = stat[0], stat[1]

    bufs = [[torch.empty(366, n, **f64) for _ in range(3)] for _ in range(2)]
    full = [torch.empty(366, nfull, **f64) for _ in range(3)] if cfg4 else None
    ready = [torch.cuda.Event() for _ in range(2)]
    freed = [torch.cuda.Event() for _ in range(2)]

    def generate(t0, t1, slot):
        """forcing of days [t0, t1) into buffer `slot`, on the generator stream"""
        with torch.cuda.stream(gen_stream):
            gen_stream.wait_event(freed[slot])
            gp = gen_stream.cuda_stream
            if cfg4:
                api.dev_synth_forcing(SEED, cal, t0, t1 - t0, cell0, nfull, lat_f, elev_f, *full, nfull, gp)
                for v in range(3):
                    torch.index_select(full[v][:t1 - t0], 1, li, out=bufs[slot][v][:t1 - t0])
            else:
                api.dev_synth_forcing(SEED, cal, t0, t1 - t0, cell0, n, lat, elev, *bufs[slot], n, gp)
            ready[slot].record(gen_stream)

    # Tt needs the whole series: one streaming pass over tc, year by year (max of the yearly maxima)
    Tt = torch.full((n,), -np.inf, **f64)
    Ttc = torch.empty(n, **f64)
    for f in freed:
        f.record(stream)
    t_setup = time.time()
    for i, (t0, t1) in enumerate(zip(ybounds[:-1], ybounds[1:])):
        generate(t0, t1, i & 1)
        stream.wait_event(ready[i & 1])
        api.dev_snow_threshold(n, t1 - t0, bufs[i & 1][1], n, par, n, Ttc, sp)
        Tt = torch.maximum(Tt, Ttc)
        freed[i & 1].record(stream)
    # spin-up on the first 365 days
    generate(0, 365, 0)
    stream.wait_event(ready[0])
    state = torch.empty(api.NSTATE, n, **f64)
    work = torch.empty(api.spin_up_work_size(n), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    api.dev_spin_up(cal, n, *bufs[0], n, Tt, par, n, state, n, work, stream=sp)
    e1.record(stream)
    freed[0].record(stream)
    torch.cuda.synchronize()
    spinup_s = e0.elapsed_time(e1) / 1e3
    state0 = state.clone()
    del work
    setup_s = time.time() - t_setup

    acc = torch.zeros(n_basin, nm, api.DIAG_REC, dtype=torch.int64, device=dev)
    diag = torch.zeros(n_basin, nm, api.NDIAG, **f64)
    if cfg4:
        outs = {k: torch.zeros(nm, n, **f64) for k in OUT5}       # monthly, whole run
    else:
        outs = {k: torch.empty(366, n, **f64) for k in OUT5}      # daily, one year, reused
    ev = []

    def step(timed):
        state.copy_(state0)
        acc.zero_()
        generate(ybounds[0], ybounds[1], 0)
        for i, (t0, t1) in enumerate(zip(ybounds[:-1], ybounds[1:])):
            if i + 1 < len(ybounds) - 1:
                generate(ybounds[i + 1], ybounds[i + 2], (i + 1) & 1)  # next year in flight
            stream.wait_event(ready[i & 1])
            b = bufs[i & 1]
            if timed:
                a_, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a_.record(stream)
            if cfg4:
                o_ = {k: outs[k][int(cal.month_id[t0]):] for k in OUT5}
                api.dev_run_all(cal, t0, t1 - t0, n, b[0], b[1], b[2], n, Tt, par, n, state, n, o_, n,
                                monthly=True, diag_acc=acc, basin=basin, n_basin=n_basin, stream=sp)
            else:
                api.dev_run_all(cal, t0, t1 - t0, n, b[0], b[1], b[2], n, Tt, par, n, state, n, outs, n,
                                monthly=False, diag_acc=acc, stream=sp)
            if timed:
                b_.record(stream)
                ev.append((a_, b_, t1 - t0))
            freed[i & 1].record(stream)
        allreduce_(ctx, acc)  # ONE all-reduce for the whole 30-year run (exact integer limbs)
        api.dev_diag_finalize(acc, n_basin, nm, diag, stream=sp)

    dt = timed_region(ctx, step, steps, warmup)
    ncell_all = torch.tensor([float(n)], dtype=torch.float64, device=dev)
    allreduce_(ctx, ncell_all)
    cells_total = float(ncell_all.item())
    cell_days_step = cells_total * nt
    value = cell_days_step * steps / dt
    k_ms = np.array([a.elapsed_time(b) for a, b, _ in ev])
    k_days = np.array([d for _, _, d in ev], dtype=np.float64)
    avg_launch_s = float(k_ms.mean()) / 1e3
    units_per_launch = float(n) * float(k_days.mean())
    if cfg4:
        workload = (f"{Htot}x{W} (0.5 degree) global grid, {int(cells_total)} land cells (hash mask, NA "
                    f"elsewhere), {years} years daily, monthly outputs + monthly totals for {n_basin} basins")
    else:
        workload = (f"{n} cells per GPU ({args.rows or 512}x{W} = one rank's 1/8 share of 4096x4096), {years} "
                    f"years ({nt} days) streamed in calendar-year chunks, daily outputs wn/snow/ro/aet/pet")
    line = base_line(ctx, value, steps, warmup, dt, workload,
                     {"days": nt, "years": years, "cells_per_gpu": n,
                      "timed": "per year: forcing generation (second stream, overlapped) + K5 run_all + K7; "
                               "one all-reduce of the totals at the end"})
    if world > 1:
        line["collective"] = collective_info(dist, acc.numel() * acc.element_size())
        line["collective"]["op"] = "all_reduce(sum) of the int64 water-balance limbs, once per 30-year run"
    line["roofline"] = roofline(avg_launch_s, units_per_launch, f"stream{n}", "k_run_all (one calendar year)")
    # What the overlapped figure hides: the stand-in generator of the NEXT year's forcing runs on the
    # second stream beside every run_all launch and takes its share of the machine.  Measured apart,
    # after the timed region: one year of generation alone, one run_all launch with the generator idle.
    torch.cuda.synchronize()
    g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(gen_stream):
        g0.record(gen_stream)
    generate(ybounds[0], ybounds[1], 0)
    with torch.cuda.stream(gen_stream):
        g1.record(gen_stream)
    torch.cuda.synchronize()
    k0, k1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st_tmp = state0.clone()
    acc_tmp = torch.zeros_like(acc)
    k0.record(stream)
    if cfg4:
        api.dev_run_all(cal, 0, ybounds[1], n, *bufs[0], n, Tt, par, n, st_tmp, n, outs, n, monthly=True,
                        diag_acc=acc_tmp, basin=basin, n_basin=n_basin, stream=sp)
    else:
        api.dev_run_all(cal, 0, ybounds[1], n, *bufs[0], n, Tt, par, n, st_tmp, n, outs, n, monthly=False,
                        diag_acc=acc_tmp, stream=sp)
    k1.record(stream)
    freed[0].record(stream)
    torch.cuda.synchronize()
    idle_s = k0.elapsed_time(k1) / 1e3
    idle = ALG_BYTES_PER_CELL_DAY * float(n) * ybounds[1] / idle_s / 1e9
    line["generator"] = {"ms_per_year": g0.elapsed_time(g1), "run_all_ms_per_year_overlapped": avg_launch_s * 1e3,
                         "run_all_ms_per_year_generator_idle": idle_s * 1e3,
                         "roofline_frac_generator_idle": idle / HBM_PEAK_GBS,
                         "note": "the synthetic-forcing generator (stand-in for the input stream of a real 30-year "
                                 "run) shares the GPU with run_all in the timed region; `roofline` is the overlapped "
                                 "launch, this is the same launch (first year) with the second stream idle"}
    if cfg4:
        line["roofline"]["note"] = ("monthly outputs: algorithmic bytes are 24 B per cell-day, not 64; the "
                                    "figure is kept on the 64 B convention for comparability. "
                                    + line["roofline"]["note"])
    sp_all = torch.tensor([spinup_s], dtype=torch.float64, device=dev)
    allreduce_(ctx, sp_all, dist.ReduceOp.MAX)
    line["value_inclusive"] = cell_days_step / (float(sp_all.item()) + dt / steps)
    line["config"].update({"value_inclusive": line["value_inclusive"], "spinup_seconds": float(sp_all.item()),
                           "inclusive_seconds": float(sp_all.item()) + dt / steps})
    line["spinup"] = {"seconds": float(sp_all.item()), "cells": n}
    line["setup_s"] = setup_s
    d = diag.cpu().numpy()
    line["water_balance"] = {"basins": n_basin,
                             "mean_annual_precip_mm_per_cell": float(d[:, :, 0].sum() / cells_total / years),
                             "mean_annual_aet_mm_per_cell": float(d[:, :, 1].sum() / cells_total / years),
                             "mean_annual_runoff_mm_per_cell": float(d[:, :, 2].sum() / cells_total / years)}
    cal.close()
    return line


if __name__ == "__main__":
    main()
