"""Multi-GPU sharding of independent streams (one process per GPU, torch.distributed).

Streams never exchange data (each is a private closure, flowz.hpp:1181-1230), so the data path
has NO collective: rank r owns a contiguous range of global stream ids and generates / receives
only those.  The single collective is the reduction of run statistics at the end (RCCL over
xGMI when the backend is "nccl", gloo on CPU): a handful of doubles.
"""
from __future__ import annotations

from typing import Dict, Tuple


def shard_range(n_streams_total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [begin, end) of global stream ids for `rank`; sizes differ by at most one."""
    q, r = divmod(int(n_streams_total), int(world))
    begin = rank * q + min(rank, r)
    return begin, begin + q + (1 if rank < r else 0)


def reduce_stats(seconds: float, samples: float, checksum: int, device=None) -> Dict[str, float]:
    """max(seconds), sum(samples), sum(checksum) over all ranks (identity when not distributed).

    `checksum` is an INTEGER (bench.py: the sum of the uint32 bit patterns of the last output row of every
    stream of the rank), reduced in int64: exact and independent of how the streams are sharded, so an
    N-rank run can be compared with a single-process run over the union of the global stream ids."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return {"seconds": float(seconds), "samples": float(samples), "checksum": int(checksum), "world": 1}
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    s = torch.tensor([samples], dtype=torch.float64, device=device)
    c = torch.tensor([int(checksum)], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(s, op=dist.ReduceOp.SUM)
    dist.all_reduce(c, op=dist.ReduceOp.SUM)
    return {"seconds": float(t[0]), "samples": float(s[0]), "checksum": int(c[0]), "world": dist.get_world_size()}


def gather_floats(value: float, device=None):
    """[value of rank 0, value of rank 1, ...] on every rank ([value] when not distributed)."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return [float(value)]
    mine = torch.tensor([value], dtype=torch.float64, device=device)
    parts = [torch.empty_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, mine)
    return [float(t[0]) for t in parts]


def sum_ints(values, device=None):
    """element-wise sum of a short list of integers over all ranks, exact in int64 (identity when not distributed)"""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return [int(v) for v in values]
    t = torch.tensor([int(v) for v in values], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [int(v) for v in t]


def bits_checksum(last_row) -> int:
    """Sum of the uint32 bit patterns of a float32 CUDA/CPU tensor (exact in int64 for < 2^31 values)."""
    import torch

    v = last_row.contiguous().view(torch.int32).to(torch.int64) & 0xFFFFFFFF
    return int(v.sum().item())
