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


# ---- work-sharded index build: communicators for lantern_gpu_add_sharded ------------------------


def torch_allgatherv(group=None):
    """An in-place host all-gather with per-rank sizes over torch.distributed (gloo on CPU tensors), in the shape
    lantern_amd.capi.Comm.host expects.  Segments are padded to the largest one (gloo has no all-gather-v)."""
    import torch

    dist = _dist()

    def allgatherv(buf: np.ndarray, offsets, counts):
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        pad = max(max(counts), 1)
        mine = torch.zeros(pad, dtype=torch.uint8)
        if counts[rank]:
            mine[: counts[rank]] = torch.from_numpy(buf[offsets[rank]: offsets[rank] + counts[rank]].copy())
        outs = [torch.empty(pad, dtype=torch.uint8) for _ in range(world)]
        dist.all_gather(outs, mine, group=group)
        for r in range(world):
            if r != rank and counts[r]:
                buf[offsets[r]: offsets[r] + counts[r]] = outs[r][: counts[r]].numpy()

    return allgatherv


def host_comm(group=None):
    """A lantern_gpu communicator whose exchange runs over torch.distributed host tensors (gloo): the debug /
    test transport.  Production multi-GPU builds use rccl_comm()."""
    from lantern_amd import capi

    dist = _dist()
    return capi.Comm.host(dist.get_rank(group), dist.get_world_size(group), torch_allgatherv(group))


def rccl_comm(group=None):
    """A lantern_gpu communicator over RCCL (xGMI): rank 0 draws the unique id, torch.distributed carries it to
    the peers (128 bytes, host side), every rank then joins with its current HIP device."""
    from lantern_amd import capi

    dist = _dist()
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    box = [capi.Comm.unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0, group=group)
    return capi.Comm.rccl(rank, world, box[0])
