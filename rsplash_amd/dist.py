"""Multi-GPU layout of the gridded run: one process per GPU, contiguous row blocks, no data-path
collective (cells are independent: the reference's "lateral flow" is a per-cell closed form,
src/SPLASH.cpp:1286-1372).  The only exchange is an all-reduce (RCCL via torch.distributed) of the
monthly grid totals used for the global water-balance diagnostics.

This replaces the SOCK-cluster block scheduler of R/splash.grid.R:260-314, 359-400.
"""
import numpy as np


def row_block_partition(n_rows, n_cols, world, rank, valid_per_row=None):
    """Rows [r0, r1) owned by `rank`.  With `valid_per_row` (count of non-NA cells per raster row,
    e.g. a land mask) the cut points balance valid cells instead of rows."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad world/rank")
    if valid_per_row is None:
        bounds = [(n_rows * k) // world for k in range(world + 1)]
    else:
        w = np.asarray(valid_per_row, dtype=np.float64)
        if len(w) != n_rows:
            raise ValueError("valid_per_row must have one entry per row")
        cum = np.r_[0.0, np.cumsum(w)]
        targets = cum[-1] * np.arange(world + 1) / world
        bounds = [int(np.searchsorted(cum, t, side="left")) for t in targets]
        bounds[0], bounds[-1] = 0, n_rows
        for k in range(1, world + 1):
            bounds[k] = max(bounds[k], bounds[k - 1])
    r0, r1 = bounds[rank], bounds[rank + 1]
    return dict(row0=r0, row1=r1, cell0=r0 * n_cols, ncell=(r1 - r0) * n_cols)


def row_blocks_round_robin(n_rows, chunk_rows, world, rank):
    """Strong scaling of ONE grid: its rows in blocks of `chunk_rows`, dealt round-robin -- rank r owns
    blocks r, r + world, ...  Every rank gets every latitude band (the day step costs more at low
    latitudes: a contiguous split leaves the equatorial rank ~5 % behind), still a row-block
    partition with no data-path exchange.  Returns [(row0, nrows), ...] in ascending order."""
    if world < 1 or not (0 <= rank < world) or chunk_rows < 1:
        raise ValueError("bad world/rank/chunk_rows")
    starts = range(0, n_rows, chunk_rows)
    return [(r0, min(chunk_rows, n_rows - r0)) for k, r0 in enumerate(starts) if k % world == rank]


def all_reduce_(t, op="sum", group=None):
    """In-place all-reduce of a tensor over the ranks of torch.distributed: directly on the device
    with the nccl (= RCCL) backend, through a host copy with gloo (CPU rehearsals, CPU tensors).
    No-op outside a process group or with one rank.  This is the ONE collective of the gridded run
    (bench.py and the multi-rank tests both go through it)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return t
    rop = {"sum": dist.ReduceOp.SUM, "max": dist.ReduceOp.MAX, "min": dist.ReduceOp.MIN}[op]
    if t.is_cuda and dist.get_backend(group) != "nccl":
        h = t.cpu()
        dist.all_reduce(h, op=rop, group=group)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=rop, group=group)
    return t


def allreduce_diag(diag_acc, group=None):
    """Global water-balance totals: sums the EXACT fixed-point limbs `diag_acc` (int64
    [n_basin, nmonth, NDIAG+1, 2], include/splash_hip.h: what k_run_all adds into
    splash_dev_out.diag_acc / what splash_grid_out.diag_acc returns) over all ranks, in place.
    Integer addition is associative: every rank ends with bitwise the same limbs, equal to those of
    a single-process run over the whole grid, whatever the partition and the reduction order -- a
    sum of doubles would have neither property.  A few KB per call: latency bound, xGMI bandwidth is
    irrelevant.  Returns (diag_acc, diag) with diag = the totals as doubles [n_basin, nmonth, NDIAG]
    (splash_diag_finalize_host: one rounding per total)."""
    import torch
    if diag_acc.dtype != torch.int64:
        raise TypeError("allreduce_diag reduces the int64 limbs (diag_acc), not rounded doubles")
    all_reduce_(diag_acc, "sum", group)
    from . import api
    return diag_acc, api.diag_finalize_host(diag_acc.cpu().numpy())[0]
