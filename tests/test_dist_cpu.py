"""CPU suite, part 3: the N > 1 path (row-block partition + diagnostics all-reduce) with two gloo
ranks.  The per-rank compute is the oracle here (no GPU in this container); what is under test is
the partition/ownership logic and the collective that bench.py and multi-GPU drivers use."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def test_row_block_partition_covers_grid_once():
    from rsplash_amd.dist import row_block_partition
    for H, W, world in ((4096, 4096, 8), (139, 159, 8), (7, 3, 4), (360, 720, 8)):
        seen = np.zeros(H * W, int)
        for r in range(world):
            p = row_block_partition(H, W, world, r)
            seen[p["cell0"]:p["cell0"] + p["ncell"]] += 1
            assert p["ncell"] == (p["row1"] - p["row0"]) * W
        assert (seen == 1).all()
    # balanced by valid cells (land mask): 0.5 deg CRU-like, most land in the top rows
    valid = np.r_[np.full(100, 300), np.full(260, 10)]
    cnt = [row_block_partition(360, 720, 4, r, valid)["row1"] -
           row_block_partition(360, 720, 4, r, valid)["row0"] for r in range(4)]
    loads = []
    for r in range(4):
        p = row_block_partition(360, 720, 4, r, valid)
        loads.append(valid[p["row0"]:p["row1"]].sum())
    assert sum(cnt) == 360 and max(loads) - min(loads) <= 2 * 300  # within two rows of the ideal


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle.binding import Oracle
    from rsplash_amd.dist import allreduce_diag, row_block_partition
    o = Oracle()
    H, W, nt = 6, 16, 400
    doy, year, month, mid, nm = o.calendar(2001, nt)
    p = row_block_partition(H, W, world, rank)
    st = o.synth_statics(20221130, H, W, p["cell0"], p["ncell"], 1)
    sw, tc, pn = o.synth_forcing(20221130, H, W, p["cell0"], p["ncell"], 0, doy, st["lat"],
                                 st["elev"])
    res, _ = o.grid(doy, year, month, mid, nm, sw, tc, pn, st["lat"], st["elev"], st["slop"],
                    st["asp"], st["soil"], st["au"], st["resolution"], n_threads=2)
    # per-rank monthly totals in the layout of SPLASH_NDIAG (aet, ro, bflow columns)
    diag = torch.zeros(nm, 6, dtype=torch.float64)
    for j, k in ((1, "aet"), (2, "ro"), (3, "bflow")):
        for m in range(nm):
            diag[m, j] = res[k][mid == m].sum()
    allreduce_diag(diag)
    q.put((rank, p, diag.numpy(), res["wn"]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_partition_and_diag_allreduce(oracle):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=300) for _ in range(world)], key=lambda x: x[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # single-process answer over the whole grid
    H, W, nt = 6, 16, 400
    doy, year, month, mid, nm = oracle.calendar(2001, nt)
    st = oracle.synth_statics(20221130, H, W, 0, H * W, 1)
    sw, tc, pn = oracle.synth_forcing(20221130, H, W, 0, H * W, 0, doy, st["lat"], st["elev"])
    res, _ = oracle.grid(doy, year, month, mid, nm, sw, tc, pn, st["lat"], st["elev"], st["slop"],
                         st["asp"], st["soil"], st["au"], st["resolution"])
    # every rank holds the global totals; the rank blocks tile the grid exactly
    wn = np.concatenate([g[3] for g in got], axis=1)
    assert np.array_equal(wn, res["wn"])
    for _, _, d, _ in got:
        for j, k in ((1, "aet"), (2, "ro"), (3, "bflow")):
            ref = np.array([res[k][mid == m].sum() for m in range(nm)])
            assert np.allclose(d[:, j], ref, rtol=1e-12)
    assert np.array_equal(got[0][2], got[1][2])


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus N` (no torch.distributed.run around it) starts N fresh rank processes
    before anything touches a GPU, rendezvous on 127.0.0.1, and relays rank 0's single JSON line;
    a failing rank makes the launcher fail.  (Dry run: gloo, no GPU, no compute.)"""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["SPLASH_BENCH_DRYRUN"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d == {"dryrun": True, "n_gpus": 3, "sum_of_ranks": 6.0}
    # without a GPU the real ranks refuse loudly and the launcher reports the failure
    env.pop("SPLASH_BENCH_DRYRUN")
    if not torch.cuda.is_available():
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env,
                           capture_output=True, text=True, timeout=300)
        assert r.returncode != 0 and not [l for l in r.stdout.splitlines() if l.startswith("{")]
