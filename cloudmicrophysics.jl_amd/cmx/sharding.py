"""Multi-GPU layout: grid points shard embarrassingly, one process per GPU.

Every point is independent, so rank r of G owns one contiguous, 256-point-aligned range of every
state column and runs the same single-GPU kernel on it: no halo, no exchange, no data-path
collective.  The only communication is OPTIONAL: the all-reduce (RCCL over xGMI when the backend is
"nccl", gloo in the CPU tests) of a handful of diagnostic sums — ≤16 doubles, latency-bound.
The reference has no counterpart (it has no parallel layer at all: SURVEY §2a).
"""
from __future__ import annotations

from typing import Tuple

import torch
import torch.distributed as dist

ALIGN = 256  # points; keeps every shard's columns 16-byte aligned for both f32 (×4 B) and f64 (×8 B)


def shard_bounds(n: int, rank: int, world: int, align: int = ALIGN) -> Tuple[int, int]:
    """[lo, hi) of rank's contiguous shard of n points; shards tile [0, n) exactly, sizes differ by
    at most one aligned block, interior boundaries are multiples of `align`."""
    if not (0 <= rank < world) or n < 0:
        raise ValueError("bad shard request")
    blocks = (n + align - 1) // align
    base, extra = divmod(blocks, world)
    lo_b = rank * base + min(rank, extra)
    hi_b = lo_b + base + (1 if rank < extra else 0)
    return min(lo_b * align, n), min(hi_b * align, n)


def allreduce_sums(local_sums: torch.Tensor, group=None) -> torch.Tensor:
    """Sum the per-rank diagnostic sums over all ranks (in place, returns the tensor).  A no-op when
    torch.distributed is not initialised (single-GPU run)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(local_sums, op=dist.ReduceOp.SUM, group=group)
    return local_sums


def global_diagnostics(columns, group=None) -> torch.Tensor:
    """Σ over ALL ranks of each device column: per-rank deterministic two-stage reduction (cmx_column_sums_*: one launch over all
    columns + a fixed-tree finish, no atomics), then one all-reduce of len(columns) doubles."""
    from .bulk_tendencies import column_sums
    return allreduce_sums(column_sums(columns), group)
