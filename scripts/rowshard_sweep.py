"""Row-sharded build: recall against the one-GPU build over world size / candidates per shard (argv: n d efc metric)."""
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, ".")
from lantern_amd import capi, synth  # noqa: E402

n, d, efc, metric = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
rng = np.random.default_rng(77)
make = synth.query_maker("clustered", d)
base, queries = make(rng, n), make(rng, 1000)
labels = np.arange(n, dtype=np.uint64) + 1


def truth_of(base, queries, k):
    b, q = base.astype(np.float64), queries.astype(np.float64)
    if metric == "cos":
        b, q = b / np.linalg.norm(b, axis=1, keepdims=True), q / np.linalg.norm(q, axis=1, keepdims=True)
        dd = 1.0 - q @ b.T
    else:
        dd = (q * q).sum(1)[:, None] - 2.0 * q @ b.T + (b * b).sum(1)[None, :]
    return np.argsort(dd, axis=1, kind="stable")[:, :k]


truth = truth_of(base, queries, 10)


def recall(ix, ef):
    lab, _, _ = ix.search_batch(queries, 10, ef)
    return float(np.mean([len(set(lab[i].tolist()) & set((truth[i] + 1).tolist())) / 10 for i in range(len(queries))]))


t = time.time()
one = capi.GpuIndex(metric, d, M=16, ef_construction=efc, ef=64, seed=21)
one.add_many(labels, base)
one.flush()
print(f"one GPU: {time.time() - t:.2f} s  recall@10 ef32 {recall(one, 32):.4f} ef64 {recall(one, 64):.4f} ef128 {recall(one, 128):.4f}", flush=True)
SWEEP = [(1, 0, 0), (2, 0, 0), (2, 64, 0), (2, 33, 0), (3, 0, 0), (4, 0, 0), (4, 33, 0)]
if len(sys.argv) > 5:  # world:K:ef,...
    SWEEP = [tuple(int(x) for x in t.split(":")) for t in sys.argv[5].split(",")]
for world, K, ef_shard in SWEEP:
    for name, v in (("LANTERN_GPU_ROW_SHARD_K", K), ("LANTERN_GPU_ROW_SHARD_EF", ef_shard)):
        if v:
            os.environ[name] = str(v)
        else:
            os.environ.pop(name, None)
    comms = capi.Comm.local_world(world)
    out, errs = [None] * world, []

    def run(r):
        try:
            comms[r].set_timeout(600)
            ix = capi.GpuIndex(metric, d, M=16, ef_construction=efc, ef=64, seed=21)
            lo, hi = n * r // world, n * (r + 1) // world
            ix.add_row_sharded(comms[r], labels[lo:hi], base[lo:hi])
            out[r] = ix
        except Exception as e:  # noqa: BLE001
            errs.append(repr(e))

    t = time.time()
    ts = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    [x.start() for x in ts]
    [x.join() for x in ts]
    assert not errs, errs
    print(f"world {world} K {K or 'default'} ef {ef_shard or 'efc'}: {time.time() - t:.2f} s  recall@10 ef32 {recall(out[0], 32):.4f} ef64 {recall(out[0], 64):.4f} ef128 {recall(out[0], 128):.4f}", flush=True)
    del out
    [c.free() for c in comms]
