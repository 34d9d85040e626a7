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


def allreduce_diag(diag, group=None):
    """Sum the [nmonth, NDIAG] monthly totals over all ranks, in place (one small all-reduce per
    time chunk: latency bound, so xGMI bandwidth is irrelevant)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(diag, op=dist.ReduceOp.SUM, group=group)
    return diag
