"""Query sharding across the GPUs of one node: the index is replicated in every GPU's HBM, a query
batch is split into contiguous slices, each rank searches its slice, results are concatenated.
There is NO collective on the data path ("replicas only" for the index); the only
communication is the optional gather of results and the max-over-ranks of a timing.

`torch.distributed` is plumbing here: backend "nccl" (= RCCL over xGMI) on GPUs, "gloo" on CPU
(tests/test_sharded_gloo.py runs this module at world_size 2 without a GPU).
"""
from __future__ import annotations

import numpy as np


def split_range(n: int, world: int) -> list[tuple[int, int]]:
    """Balanced contiguous slices: the first n % world ranks get one extra row."""
    base, extra = divmod(n, world)
    out, b = [], 0
    for r in range(world):
        e = b + base + (1 if r < extra else 0)
        out.append((b, e))
        b = e
    return out


def my_slice(n: int, rank: int, world: int) -> tuple[int, int]:
    return split_range(n, world)[rank]


def _dist():
    import torch.distributed as dist

    return dist


def max_over_ranks(value: float, device=None) -> float:
    """The bench contract's timing reduction."""
    import torch

    dist = _dist()
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_rows(local: np.ndarray, n_total: int, device=None) -> np.ndarray:
    """Concatenate every rank's slice (in rank order) on every rank.  Slices may be ragged."""
    import torch

    dist = _dist()
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return local
    world = dist.get_world_size()
    sizes = [e - b for b, e in split_range(n_total, world)]
    width = int(np.prod(local.shape[1:])) if local.ndim > 1 else 1
    pad = max(sizes)
    buf = torch.zeros((pad, width), dtype=torch.from_numpy(local.reshape(local.shape[0], -1)[:0]).dtype, device=device)
    if local.shape[0]:
        buf[: local.shape[0]] = torch.from_numpy(np.ascontiguousarray(local).reshape(local.shape[0], -1)).to(buf.device)
    outs = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(outs, buf)
    parts = [o[:s].cpu().numpy() for o, s in zip(outs, sizes)]
    return np.concatenate(parts, axis=0).reshape((n_total,) + local.shape[1:])


def sharded_search(search_fn, queries: np.ndarray, k: int, device=None):
    """search_fn(queries_slice, k) -> (labels[n,k] u64, dists[n,k] f32) on this rank's replica.
    Returns the full (labels, dists) for all queries on every rank."""
    dist = _dist()
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    b, e = my_slice(queries.shape[0], rank, world)
    labels, dists = search_fn(queries[b:e], k)
    # u64 labels travel as i64 bit patterns (gloo/nccl have no u64)
    lab = gather_rows(np.ascontiguousarray(labels).view(np.int64), queries.shape[0], device).view(np.uint64)
    dst = gather_rows(np.ascontiguousarray(dists), queries.shape[0], device)
    return lab, dst
