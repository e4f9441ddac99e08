"""`python tests/env_switch_probe.py` in a process of its own (several switches are read once per process): a small build, searches in
the three launch regimes, an exact k-NN, a compact pq index -- and one JSON line of checksums.  tests/test_gpu_env_switches.py runs it
once with the default environment and once per result-neutral switch and requires the same line."""
import hashlib
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lantern_amd import capi  # noqa: E402


def digest(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()[:16]


def main():
    rng = np.random.default_rng(5)
    out = {}
    for name, metric, n, d in (("l2sq_128", "l2sq", 4000, 128), ("cos_768", "cos", 2500, 768)):
        base = rng.standard_normal((n, d), dtype=np.float32)
        queries = rng.standard_normal((600, d), dtype=np.float32)
        ix = capi.GpuIndex(metric, d, M=16, ef_construction=64, ef=64, seed=3)
        ix.set_add_batch(512, 16)
        ix.add_many(np.arange(n, dtype=np.uint64) + 1, base)
        ix.flush()
        out[name + "_graph"] = f"{ix.checksum():016x}"
        for nq in (1, 40, 600):  # lone query | a batch that cannot fill the chip | one that can
            lab, dist, cnt = ix.search_batch(queries[:nq], 10)
            out[f"{name}_search_{nq}"] = digest(lab, dist, cnt)
        lab1, dist1 = ix.search(queries[0], 10)
        out[name + "_search_ef"] = digest(lab1, dist1)
        slots, dists = ix.exact_search(queries[:64], 10)
        out[name + "_exact"] = digest(slots, dists)
        ix.add(10**6, base[0] * np.float32(0.5))  # one ldb_aminsert-sized insertion on top
        out[name + "_graph_after_insert"] = f"{ix.checksum():016x}"
    # a pq index, expanded and compact
    n, d, S, C = 2000, 128, 16, 64
    base = rng.standard_normal((n, d), dtype=np.float32)
    cb = np.zeros((C, d), dtype=np.float32)
    for s in range(S):
        cb[:, s * 8:(s + 1) * 8] = base[rng.choice(n, size=C, replace=False), s * 8:(s + 1) * 8]
    ix = capi.GpuIndex("l2sq", d, M=8, ef_construction=48, ef=40, seed=3, pq_codebook=cb, num_subvectors=S)
    ix.add_many(np.arange(n, dtype=np.uint64) + 1, base)
    ix.flush()
    q = rng.standard_normal((100, d), dtype=np.float32)
    lab, dist, _ = ix.search_batch(q, 10)
    out["pq_expanded"] = digest(lab, dist)
    ix.pq_compact()
    lab2, dist2, _ = ix.search_batch(q, 10)
    out["pq_compact_labels"] = digest(lab2)
    out["pq_compact_equals_expanded"] = bool(np.array_equal(lab, lab2))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
