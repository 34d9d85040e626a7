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


def test_strong_scaling_row_blocks_tile_the_grid_once():
    """bench.py --scaling strong: THE grid's rows in blocks dealt round-robin to the ranks"""
    from rsplash_amd.dist import row_blocks_round_robin
    for H, chunk, world in ((4096, 64, 8), (4096, 64, 1), (4096, 64, 3), (139, 16, 4), (5, 64, 2), (256, 1, 7)):
        seen = np.zeros(H, int)
        per_rank = []
        for r in range(world):
            blocks = row_blocks_round_robin(H, chunk, world, r)
            assert blocks == sorted(blocks)
            for r0, nr in blocks:
                seen[r0:r0 + nr] += 1
            per_rank.append(sum(nr for _, nr in blocks))
        assert (seen == 1).all()
        assert max(per_rank) - min(per_rank) <= chunk
    # 8 ranks of the benchmark grid: 512 rows each, every rank from the first to the last band
    b = row_blocks_round_robin(4096, 64, 8, 3)
    assert sum(nr for _, nr in b) == 512 and b[0] == (192, 64) and b[-1] == (3776, 64)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


NDIAG = 6


def _limbs(res, pn, mid, nm):
    """The exact fixed-point limbs k_run_all accumulates (include/splash_hip.h), formed
    here from the oracle's rasters: per cell and month precipitation, aet, ro, bflow summed, wn and
    snow at the month's last day; v = a/2^16 + r, a = rint(v*2^16), b = rint(r*2^48); int64 sums
    over cells.  Layout [1, nmonth, NDIAG+1, 2]; the last record counts cell-months."""
    acc = np.zeros((1, nm, NDIAG + 1, 2), np.int64)
    for m in range(nm):
        days = np.where(mid == m)[0]
        figs = [pn[days].sum(0), res["aet"][days].sum(0), res["ro"][days].sum(0), res["bflow"][days].sum(0),
                res["wn"][days[-1]], res["snow"][days[-1]]]
        for q, v in enumerate(figs):
            a = np.rint(v * 65536.0)
            r = v - a / 65536.0
            acc[0, m, q, 0] = a.astype(np.int64).sum()
            acc[0, m, q, 1] = np.rint(r * 281474976710656.0).astype(np.int64).sum()
        acc[0, m, NDIAG, 0] = v.size
    return acc


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
    # this rank's limbs, then THE collective of the gridded run (what bench.py issues per step)
    acc = torch.from_numpy(_limbs(res, pn, mid, nm))
    acc, diag = allreduce_diag(acc)
    with pytest.raises(TypeError):  # rounded doubles are refused: their sum depends on the order
        allreduce_diag(torch.zeros(1, nm, NDIAG))
    q.put((rank, p, acc.numpy().copy(), diag, res["wn"]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_partition_and_exact_limb_allreduce(oracle):
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
    # the rank blocks tile the grid exactly
    wn = np.concatenate([g[4] for g in got], axis=1)
    assert np.array_equal(wn, res["wn"])
    # every rank ends with the limbs of the single-process run, BITWISE, and so with the same doubles
    whole = _limbs(res, pn, mid, nm)
    for _, _, acc, diag, _ in got:
        assert acc.dtype == np.int64 and np.array_equal(acc, whole)
    assert np.array_equal(got[0][3], got[1][3])
    # the doubles (splash_diag_finalize_host) are the totals to the last bit or so
    from rsplash_amd import api
    d1, cnt = api.diag_finalize_host(whole)
    assert np.array_equal(d1, got[0][3]) and (cnt[0, :, 0] == H * W).all()
    for qi, k in ((1, "aet"), (2, "ro"), (3, "bflow")):
        ref = np.array([res[k][mid == m].astype(np.longdouble).sum() for m in range(nm)], dtype=np.longdouble)
        assert np.allclose(d1[0, :, qi], ref.astype(np.float64), rtol=1e-13, atol=1e-9)


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
    coll = d.pop("collective")
    # the per-rank table of the line (`ranks`: cells, ms per step, tail seconds ... of EVERY rank)
    # is put together by one sum all-reduce of a table each rank fills one line of: rehearsed with pretend figures
    assert d.pop("ranks") == {"cells": [100, 200, 300], "rank": [0, 1, 2]}
    assert d == {"dryrun": True, "n_gpus": 3, "sum_of_ranks": 6.0}
    # the line says what the process group itself saw, not just WORLD_SIZE from the environment
    assert coll["backend"] == "gloo" and coll["world_size"] == 3 and coll["bytes"] == 8
    # without a GPU the real ranks refuse loudly and the launcher reports the failure
    env.pop("SPLASH_BENCH_DRYRUN")
    if not torch.cuda.is_available():
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env,
                           capture_output=True, text=True, timeout=300)
        assert r.returncode != 0 and not [l for l in r.stdout.splitlines() if l.startswith("{")]
