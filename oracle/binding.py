"""ctypes binding of the CPU oracle (oracle/lantern_oracle.h).

TEST INFRASTRUCTURE.  Importable only from tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  Nothing under lantern_amd/ may import this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "_build", "liblantern_oracle.so")
NATIVE_LIB_PATH = os.path.join(HERE, "_build_native", "liblantern_oracle.so")

METRIC_COS, METRIC_L2SQ, METRIC_HAMMING, METRIC_COS_B1 = 1, 3, 8, 9
SUM_SEQ, SUM_WAVE64, SUM_FAST, SUM_WAVE64_F16, SUM_I8 = 0, 1, 2, 3, 4
EMPTY = 0xFFFFFFFF

METRICS = {"cos": METRIC_COS, "l2sq": METRIC_L2SQ, "hamming": METRIC_HAMMING, "cos_b1": METRIC_COS_B1}
BIT_METRICS = (METRIC_HAMMING, METRIC_COS_B1)  # rows are u32 words of bits


def build(force: bool = False) -> str:
    """Compile the oracle with its Makefile (gcc only)."""
    if force or not os.path.exists(LIB_PATH) or any(
        os.path.getmtime(os.path.join(HERE, f)) > os.path.getmtime(LIB_PATH)
        for f in ("hnsw.c", "metrics.c", "metrics_fast.c", "lantern_oracle.h", "Makefile")
    ):
        subprocess.check_call(["make", "-s", "-C", HERE])
    return LIB_PATH


def build_native() -> bool:
    """Rebuild the oracle with -march=native INTO A SEPARATE DIRECTORY on the machine that will time it
    (the reference's own build offers -march=native: lantern_hnsw/CMakeLists.txt:134-136).  Used by
    bench.py's cpu_baseline leg so the CPU gets its best code; the portable x86-64-v3 build stays the
    one the tests use.  Returns False when no compiler is available."""
    try:
        # -B: always recompile here -- objects built for another host's `native` must never be reused
        subprocess.check_call(["make", "-s", "-B", "-C", HERE, "ORACLE_MARCH=native", "BUILD=_build_native"],
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        return os.path.exists(NATIVE_LIB_PATH)
    except Exception:
        return False


_lib = None
_use_native = False


def use_native(flag: bool = True):
    """Select the -march=native build for subsequently loaded handles (call before lib())."""
    global _use_native, _lib
    _use_native = flag and os.path.exists(NATIVE_LIB_PATH)
    _lib = None
    return _use_native


def lib() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        build()
    L = C.CDLL(NATIVE_LIB_PATH if _use_native else LIB_PATH)
    vp, sz, u32, u64, i32 = C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint64, C.c_int
    L.lo_distance.restype = C.c_float
    L.lo_distance.argtypes = [vp, vp, sz, i32, i32]
    L.lo_wave_group_lanes.restype = i32
    L.lo_wave_group_lanes.argtypes = [sz]
    L.lo_level_for.restype = i32
    L.lo_level_for.argtypes = [u64, u64, u32]
    L.lo_bruteforce.restype = None
    L.lo_bruteforce.argtypes = [vp, sz, sz, i32, i32, vp, sz, sz, vp, vp, i32]
    L.lo_create.restype = vp
    L.lo_create.argtypes = [i32, sz, u32, u32, u32, u64, i32]
    L.lo_free.argtypes = [vp]
    L.lo_reserve.argtypes = [vp, sz]
    L.lo_size.restype = sz
    L.lo_size.argtypes = [vp]
    L.lo_capacity.restype = sz
    L.lo_capacity.argtypes = [vp]
    L.lo_max_level.argtypes = [vp]
    L.lo_entry_slot.restype = u32
    L.lo_entry_slot.argtypes = [vp]
    L.lo_add.argtypes = [vp, u64, vp]
    L.lo_add_with_level.argtypes = [vp, u64, vp, i32]
    L.lo_add_batch.argtypes = [vp, vp, vp, sz]
    L.lo_add_batch_cand.restype = i32
    L.lo_add_batch_cand.argtypes = [vp, vp, vp, sz, vp, vp, vp, sz]
    L.lo_set_pq_view.restype = i32
    L.lo_set_pq_view.argtypes = [vp, u32, u32, vp, vp]
    L.lo_set_wave_simd.restype = None
    L.lo_set_wave_simd.argtypes = [i32]
    L.lo_set_build_threads.restype = None
    L.lo_set_build_threads.argtypes = [vp, i32]
    L.lo_plan_batch.restype = sz
    L.lo_plan_batch.argtypes = [sz, i32, vp, sz, sz, sz]
    L.lo_search.restype = sz
    L.lo_search.argtypes = [vp, vp, sz, sz, sz, vp, vp, vp]
    L.lo_search_batch.restype = None
    L.lo_search_batch.argtypes = [vp, vp, sz, sz, sz, vp, vp, vp, vp, vp, i32]
    L.lo_last_distance_evals.restype = u64
    L.lo_last_distance_evals.argtypes = [vp]
    L.lo_last_expansions.restype = u64
    L.lo_last_expansions.argtypes = [vp]
    L.lo_upper_blocks.restype = sz
    L.lo_upper_blocks.argtypes = [vp]
    L.lo_export_graph.restype = None
    L.lo_export_graph.argtypes = [vp, vp, vp, vp, vp, vp]
    L.lo_import_graph.restype = vp
    L.lo_import_graph.argtypes = [i32, sz, u32, u32, u32, u64, i32, sz, vp, vp, vp, vp, vp, vp, u32, i32, i32]
    L.lo_estimate_visited_tuples.restype = u64
    L.lo_estimate_visited_tuples.argtypes = [C.c_double, u32, u32]
    _lib = L
    return L


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _rows(x, metric):
    """f32 rows for cos/l2sq, u32 words for hamming; C-contiguous 2-D."""
    dt = np.uint32 if metric in BIT_METRICS else np.float32
    a = np.ascontiguousarray(x, dtype=dt)
    return a.reshape(1, -1) if a.ndim == 1 else a


def distance(a, b, metric: str | int, sum_mode: int = SUM_SEQ) -> float:
    m = METRICS.get(metric, metric)
    A, B = _rows(a, m)[0], _rows(b, m)[0]
    if A.shape != B.shape:
        raise ValueError("expected equally sized arrays")
    dims = A.size * 32 if m in BIT_METRICS else A.size
    return float(lib().lo_distance(_ptr(A), _ptr(B), dims, m, sum_mode))


def set_wave_simd(on: bool):
    """SUM_WAVE64 over f32: the AVX2 form (default where available) or the scalar restatement -- identical bits."""
    lib().lo_set_wave_simd(1 if on else 0)


def round_f16(x) -> np.ndarray:
    """f32 -> f16 -> f32, round-to-nearest-even: what an f16 index stores and what it casts a query to."""
    return np.ascontiguousarray(x, dtype=np.float32).astype(np.float16).astype(np.float32)


def quantize_i8(x) -> np.ndarray:
    """f32 -> the integers an i8 index stores, returned as f32: trunc(clamp(x * 100, -100, 100)) in f32 arithmetic
    (usearch's i8 storage; lantern_hnsw/test/sql/hnsw_sq.sql:33-34), NaN -> 0.  Feed these to SUM_I8."""
    v = np.ascontiguousarray(x, dtype=np.float32) * np.float32(100.0)
    v = np.where(np.isnan(v), np.float32(0), v)
    return np.trunc(np.clip(v, np.float32(-100.0), np.float32(100.0))).astype(np.float32)


def level_for(seed: int, slot: int, M: int) -> int:
    return int(lib().lo_level_for(seed, slot, M))


def levels_for(seed: int, first_slot: int, count: int, M: int) -> np.ndarray:
    """lo_level_for of `count` consecutive slots (vectorised restatement of oracle/hnsw.c lo_level_for: splitmix64 hash of
    (seed, slot) -> U in (0, 1] -> floor(-ln(U) / ln(M)); checked against the C function in tests/test_oracle_golden.py)."""
    m64 = np.uint64(0xFFFFFFFFFFFFFFFF)

    def mix(x):
        with np.errstate(over="ignore"):
            x = (x + np.uint64(0x9E3779B97F4A7C15)) & m64
            x = ((x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & m64
            x = ((x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & m64
            return x ^ (x >> np.uint64(31))

    with np.errstate(over="ignore"):
        slots = np.arange(first_slot, first_slot + count, dtype=np.uint64)
        h = mix(np.uint64(seed) ^ mix(slots + np.uint64(0x632BE59BD9B4E019)))
    u = ((h >> np.uint64(11)).astype(np.float64) + 1.0) * (1.0 / 9007199254740992.0)
    level = -np.log(u) * (1.0 / np.log(float(M)))
    return np.minimum(level, 255.0).astype(np.int32)


def plan_batch(size, max_level, pending_levels, max_batch, min_ratio) -> int:
    lv = np.ascontiguousarray(pending_levels, dtype=np.int32)
    return int(lib().lo_plan_batch(size, max_level, _ptr(lv), lv.size, max_batch, min_ratio))


def bruteforce(rows, queries, k, metric, sum_mode=SUM_SEQ, nthreads=1):
    m = METRICS.get(metric, metric)
    R, Q = _rows(rows, m), _rows(queries, m)
    dims = R.shape[1] * 32 if m in BIT_METRICS else R.shape[1]
    ids = np.empty((Q.shape[0], k), dtype=np.uint32)
    dists = np.empty((Q.shape[0], k), dtype=np.float32)
    lib().lo_bruteforce(_ptr(R), R.shape[0], dims, m, sum_mode, _ptr(Q), Q.shape[0], k, _ptr(ids), _ptr(dists), nthreads)
    return ids, dists


class OracleIndex:
    """The oracle's HNSW index (usearch semantics; see oracle/hnsw.c)."""

    def __init__(self, metric, dims, M=16, ef_construction=128, ef=64, seed=42, sum_mode=SUM_SEQ, _handle=None,
                 _keep=None):
        self.metric = METRICS.get(metric, metric)
        self.dims = dims  # f32 scalars, or u32 WORDS for hamming (bits = 32*dims, scan.c:84-88)
        self.M, self.efc, self.ef, self.seed, self.sum_mode = M, ef_construction, ef, seed, sum_mode
        self._keep = _keep
        bits_or_dims = dims * 32 if self.metric in BIT_METRICS else dims
        self.h = _handle or lib().lo_create(self.metric, bits_or_dims, M, ef_construction, ef, seed, sum_mode)
        if not self.h:
            raise ValueError("lo_create failed")

    def __del__(self):
        if getattr(self, "h", None):
            lib().lo_free(self.h)
            self.h = None

    def __len__(self):
        return int(lib().lo_size(self.h))

    @property
    def max_level(self):
        return int(lib().lo_max_level(self.h))

    @property
    def entry_slot(self):
        return int(lib().lo_entry_slot(self.h))

    def reserve(self, n):
        lib().lo_reserve(self.h, n)

    def add(self, label, vec, level=None):
        v = _rows(vec, self.metric)[0]
        assert v.size == self.dims, "dimension mismatch"
        if level is None:
            rc = lib().lo_add(self.h, int(label), _ptr(v))
        else:
            rc = lib().lo_add_with_level(self.h, int(label), _ptr(v), int(level))
        assert rc == 0

    def add_many(self, labels, vecs):
        for l, v in zip(labels, _rows(vecs, self.metric)):
            self.add(l, v)

    def set_pq_view(self, codebook, codes):
        """Searches evaluate rows by ADC over `codes` ([n][num_subvectors] u8) with `codebook` ([num_centroids][dims] f32); the
        index's vectors must be the decoded rows."""
        cb = np.ascontiguousarray(codebook, dtype=np.float32)
        cd = np.ascontiguousarray(codes, dtype=np.uint8)
        assert cd.shape[0] == len(self) and cb.shape[1] == self.dims
        rc = lib().lo_set_pq_view(self.h, cd.shape[1], cb.shape[0], _ptr(cb), _ptr(cd))
        assert rc == 0, rc

    def set_build_threads(self, nthreads):
        """Threads for the two phases of add_batch / add_planned (the graph does not depend on the number)."""
        lib().lo_set_build_threads(self.h, int(nthreads))

    def add_batch(self, labels, vecs):
        V = _rows(vecs, self.metric)
        lab = np.ascontiguousarray(labels, dtype=np.uint64)
        rc = lib().lo_add_batch(self.h, _ptr(lab), _ptr(V), V.shape[0])
        assert rc == 0, rc

    def add_batch_cand(self, labels, vecs, cand_slot, cand_d, cand_n):
        """One batch of the row-sharded build: node i's level-0 candidates are cand_slot / cand_d [i][:cand_n[i]] (lo_add_batch_cand)."""
        V = _rows(vecs, self.metric)
        lab = np.ascontiguousarray(labels, dtype=np.uint64)
        cs = np.ascontiguousarray(cand_slot, dtype=np.uint32)
        cd = np.ascontiguousarray(cand_d, dtype=np.float32)
        cn = np.ascontiguousarray(cand_n, dtype=np.uint32)
        assert cs.shape == cd.shape and cs.shape[0] == V.shape[0] == cn.size
        rc = lib().lo_add_batch_cand(self.h, _ptr(lab), _ptr(V), V.shape[0], _ptr(cs), _ptr(cd), _ptr(cn), cs.shape[1])
        assert rc == 0, rc

    def add_planned(self, labels, vecs, max_batch=4096, min_ratio=16):
        """Insert with the device builder's batch plan (lo_plan_batch)."""
        V = _rows(vecs, self.metric)
        lab = np.ascontiguousarray(labels, dtype=np.uint64)
        i, n = 0, V.shape[0]
        while i < n:
            size = len(self)
            look = min(n - i, max_batch)
            lv = levels_for(self.seed, size, look, self.M)
            b = plan_batch(size, self.max_level, lv, max_batch, min_ratio)
            self.add_batch(lab[i:i + b], V[i:i + b])
            i += b

    def search(self, query, k, ef=0, skip=0):
        q = _rows(query, self.metric)[0]
        labels = np.zeros(k, dtype=np.uint64)
        dists = np.zeros(k, dtype=np.float32)
        slots = np.zeros(k, dtype=np.uint32)
        n = lib().lo_search(self.h, _ptr(q), k, ef, skip, _ptr(labels), _ptr(dists), _ptr(slots))
        return labels[:n], dists[:n], slots[:n]

    def last_counters(self):
        return int(lib().lo_last_distance_evals(self.h)), int(lib().lo_last_expansions(self.h))

    def search_batch(self, queries, k, ef=0, nthreads=1):
        Q = _rows(queries, self.metric)
        nq = Q.shape[0]
        labels = np.zeros((nq, k), dtype=np.uint64)
        dists = np.zeros((nq, k), dtype=np.float32)
        slots = np.zeros((nq, k), dtype=np.uint32)
        D = np.zeros(nq, dtype=np.uint64)
        E = np.zeros(nq, dtype=np.uint64)
        lib().lo_search_batch(self.h, _ptr(Q), nq, k, ef, _ptr(labels), _ptr(dists), _ptr(slots), _ptr(D), _ptr(E),
                              nthreads)
        return labels, dists, slots, D, E

    def export_graph(self):
        n, M = len(self), self.M
        blocks = int(lib().lo_upper_blocks(self.h))
        g = {
            "levels": np.zeros(n, dtype=np.uint8),
            "nbr0": np.zeros((n, 2 * M), dtype=np.uint32),
            "upper_off": np.zeros(n, dtype=np.uint32),
            "upper_nbr": np.zeros((max(blocks, 1), M), dtype=np.uint32),
            "labels": np.zeros(n, dtype=np.uint64),
        }
        lib().lo_export_graph(self.h, _ptr(g["levels"]), _ptr(g["nbr0"]), _ptr(g["upper_off"]), _ptr(g["upper_nbr"]),
                              _ptr(g["labels"]))
        g["upper_nbr"] = g["upper_nbr"][:blocks]
        g["entry_slot"] = self.entry_slot
        g["max_level"] = self.max_level
        return g

    @classmethod
    def from_graph(cls, metric, vectors, graph, M, ef_construction=128, ef=64, seed=42, sum_mode=SUM_SEQ,
                   borrow=True):
        m = METRICS.get(metric, metric)
        V = _rows(vectors, m)
        n, dims = V.shape
        bits_or_dims = dims * 32 if m in BIT_METRICS else dims
        levels = np.ascontiguousarray(graph["levels"], dtype=np.uint8)
        nbr0 = np.ascontiguousarray(graph["nbr0"], dtype=np.uint32)
        upper_off = np.ascontiguousarray(graph["upper_off"], dtype=np.uint32)
        upper_nbr = np.ascontiguousarray(graph["upper_nbr"], dtype=np.uint32)
        labels = np.ascontiguousarray(graph["labels"], dtype=np.uint64) if graph.get("labels") is not None else None
        h = lib().lo_import_graph(m, bits_or_dims, M, ef_construction, ef, seed, sum_mode, n, _ptr(V), _ptr(labels),
                                  _ptr(levels), _ptr(nbr0), _ptr(upper_off), _ptr(upper_nbr),
                                  int(graph["entry_slot"]), int(graph["max_level"]), 1 if borrow else 0)
        return cls(metric, dims, M, ef_construction, ef, seed, sum_mode, _handle=h, _keep=V if borrow else None)


def row_shard_plan(sizes, levels, max_batch, min_ratio):
    """[(first, b, [rows of the batch that come from shard r, ...]), ...]: the batches of the row-sharded build and where their
    members come from -- the usual plan over the GLOBAL level draw, position p of the global order goes to the
    shard that is furthest behind its proportional share of the rows handed out so far, ties to the lower rank (lantern_amd/csrc/index.cpp add_row_sharded_locked)."""
    W, N = len(sizes), int(sum(sizes))
    taken, out, pi, max_level = [0] * W, [], 0, 0
    while pi < N:
        look = min(N - pi, max_batch)
        b = plan_batch(pi, max_level, levels[pi:pi + look], max_batch, min_ratio)
        if pi == 0 or (b == 1 and levels[pi] > max_level):
            max_level = int(levels[pi])
        share = [0] * W
        for j in range(b):
            behind = [int(sizes[r]) * (pi + j + 1) - taken[r] * N for r in range(W)]
            best = behind.index(max(behind))
            taken[best] += 1
            share[best] += 1
        out.append((pi, b, share))
        pi += b
    return out


def row_sharded_build(metric, dims, shards, M=16, ef_construction=128, ef=64, seed=42, max_batch=8192, min_ratio=16, sum_mode=SUM_SEQ,
                      per_shard=None):
    """CPU restatement of lantern_gpu_add_row_sharded (lantern_amd/csrc/index.cpp add_row_sharded_locked; SURVEY.md 8e as written).
    shards = [(labels, rows), ...] in rank order.  Every shard keeps a graph over ITS rows (labels there = global slot + 1), grown
    batch by batch with the device's plan; a batch's rows are searched in every shard's graph AS IT STOOD BEFORE THE BATCH (k = ef =
    per_shard; the batch's members join their shards afterwards), the answers are merged by (distance, slot), the best
    ef_construction of them are the level-0 candidates
    of lo_add_batch_cand on the global graph.  Returns (the global index, labels in slot order)."""
    W = len(shards)
    sizes = [len(lab) for lab, _ in shards]
    N = int(sum(sizes))
    levels = levels_for(seed, 0, N, M)
    K = per_shard or min(ef_construction, max(2 * M + 1, 2 * ef_construction // W))
    glob = OracleIndex(metric, dims, M=M, ef_construction=ef_construction, ef=ef, seed=seed, sum_mode=sum_mode)
    glob.reserve(max(N, 1))
    locs = [OracleIndex(metric, dims, M=M, ef_construction=ef_construction, ef=ef, seed=seed, sum_mode=sum_mode) for _ in range(W)]
    for r in range(W):
        locs[r].reserve(max(sizes[r], 1))
    cur = [0] * W
    by_slot = np.zeros(N, dtype=np.uint64)
    for first, b, share in row_shard_plan(sizes, levels, max_batch, min_ratio):
        rows = np.zeros((b, np.asarray(shards[0][1]).shape[1]), dtype=np.asarray(shards[0][1]).dtype)
        labs = np.zeros(b, dtype=np.uint64)
        at = 0
        joins = []  # (shard, global slots + 1, rows): the batch's members join their shards' graphs AFTER the batch's candidate searches
        for r in range(W):
            n_r = share[r]
            if n_r:
                lab_r, rows_r = shards[r]
                seg = slice(cur[r], cur[r] + n_r)
                joins.append((r, np.arange(first + at, first + at + n_r, dtype=np.uint64) + 1, rows_r[seg]))
                rows[at:at + n_r] = rows_r[seg]
                labs[at:at + n_r] = lab_r[seg]
                cur[r] += n_r
                at += n_r
        cs = np.zeros((b, ef_construction), dtype=np.uint32)
        cd = np.zeros((b, ef_construction), dtype=np.float32)
        cn = np.zeros(b, dtype=np.uint32)
        if first:
            for i in range(b):
                L, D = [], []
                for r in range(W):
                    if len(locs[r]):
                        l, d, _ = locs[r].search(rows[i], K, K)
                        keep = l <= first  # label = slot + 1: the slots before this batch (all of them: the shards hold nothing newer)
                        L.append(l[keep])
                        D.append(d[keep])
                La = np.concatenate(L) if L else np.zeros(0, dtype=np.uint64)
                Da = np.concatenate(D) if D else np.zeros(0, dtype=np.float32)
                order = np.lexsort((La, Da))[:ef_construction]
                cn[i] = order.size
                cs[i, :order.size] = (La[order] - 1).astype(np.uint32)
                cd[i, :order.size] = Da[order]
        for r, slots1, rws in joins:
            locs[r].add_planned(slots1, rws, max_batch, min_ratio)
        glob.add_batch_cand(labs, rows, cs, cd, cn)
        by_slot[first:first + b] = labs
    return glob, by_slot


def recall_at_k(found_ids, true_ids) -> float:
    """|top-k_ann ∩ top-k_exact| / k averaged over queries (index_autotune/mod.rs:239-247,
    test/sql/utils/calculate_recall.sql:9-24)."""
    f, t = np.asarray(found_ids), np.asarray(true_ids)
    hits = 0
    for a, b in zip(f, t):
        hits += len(set(int(x) for x in a if x != EMPTY) & set(int(x) for x in b if x != EMPTY))
    return hits / float(t.shape[0] * t.shape[1])
