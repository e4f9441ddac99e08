"""bench_cache_model.py -- Python side of lantern_amd/tools/cache_model.c: the LRU replay of a search launch's memory-object trace
(lantern_gpu_search_row_trace) through eight 4 MiB L2s and the 256 MiB Infinity Cache, behind bench.py's roofline.frac_dram_model.

A measurement harness (bench.py, tests/test_cache_model.py); nothing of the search path uses it."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(ROOT, "lantern_amd", "lib", "libcache_model.so")
L2_BYTES_PER_XCD = 4 << 20     # /opt/skills/guides/MI355X_MICROARCH.md: 4 MiB L2 per XCD, eight XCDs
MALL_BYTES = 256 << 20         # 256 MiB Infinity Cache (memory side, shared)
XCDS = 8
LIST0, LISTU = 0x80000000, 0xC0000000  # trace entry flags (include/lantern_gpu.h, lantern_gpu_search_row_trace)


class Config(C.Structure):
    _fields_ = [("l2_bytes_per_xcd", C.c_uint64), ("mall_bytes", C.c_uint64), ("xcds", C.c_uint32), ("walkers", C.c_uint32),
                ("row_bytes", C.c_uint32), ("list0_bytes", C.c_uint32), ("listu_bytes", C.c_uint32), ("reserved", C.c_uint32)]


class Result(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("accesses", "access_bytes", "row_accesses", "list_accesses", "l2_miss_bytes", "mall_miss_bytes",
                                          "l2_hits", "mall_hits", "dropped_entries")]


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(LIB)
        _lib.cache_model_replay.restype = C.c_int
        _lib.cache_model_replay.argtypes = [C.POINTER(Config), C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_uint32), C.c_uint32,
                                            C.POINTER(Result)]
    return _lib


def replay(traces, counts, walkers, row_bytes, list0_bytes, listu_bytes, l2_bytes_per_xcd=L2_BYTES_PER_XCD, mall_bytes=MALL_BYTES, xcds=XCDS):
    """traces: list of (nq, cap) u32 arrays, one per launch, replayed back to back through ONE set of caches; counts: list of (nq,) u32.
    Returns one dict per launch: accesses, access_bytes, fabric_bytes (L2 misses), dram_bytes (Infinity-Cache misses), hit counts."""
    L = len(traces)
    assert L == len(counts) and L > 0
    cap = traces[0].shape[1]
    tr = [np.ascontiguousarray(t, dtype=np.uint32) for t in traces]
    ct = [np.ascontiguousarray(c, dtype=np.uint32) for c in counts]
    assert all(t.ndim == 2 and t.shape[1] == cap and t.shape[0] == c.shape[0] for t, c in zip(tr, ct))
    cfg = Config(int(l2_bytes_per_xcd), int(mall_bytes), int(xcds), int(walkers), int(row_bytes), int(list0_bytes), int(listu_bytes), 0)
    tp = (C.c_void_p * L)(*[t.ctypes.data for t in tr])
    cp = (C.c_void_p * L)(*[c.ctypes.data for c in ct])
    nq = (C.c_uint32 * L)(*[t.shape[0] for t in tr])
    out = (Result * L)()
    rc = lib().cache_model_replay(C.byref(cfg), L, tp, cp, nq, cap, out)
    if rc != 0:
        raise RuntimeError("cache_model_replay failed (allocation or arguments)")
    res = []
    for r in out:
        res.append({"accesses": r.accesses, "access_bytes": r.access_bytes, "row_accesses": r.row_accesses, "list_accesses": r.list_accesses,
                    "fabric_bytes": r.l2_miss_bytes, "dram_bytes": r.mall_miss_bytes, "l2_hit_objects": r.l2_hits, "mall_hit_objects": r.mall_hits,
                    "dropped_entries": r.dropped_entries})
    return res


def distinct_bytes(trace, count, row_bytes, list0_bytes, listu_bytes):
    """Bytes of the distinct objects one launch asks for (what DRAM must deliver with perfect caches that start empty)."""
    cap = trace.shape[1]
    valid = np.arange(cap, dtype=np.uint32)[None, :] < np.minimum(count, cap)[:, None]
    u = np.unique(trace[valid])
    kind = u >> 30
    return float((kind < 2).sum()) * row_bytes + float((kind == 2).sum()) * list0_bytes + float((kind == 3).sum()) * listu_bytes
