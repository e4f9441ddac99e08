"""Build-to-build spread of recall on the clustered set: one-GPU and row-sharded builds over seeds (argv: n d efc metric world)."""
import sys
import threading

import numpy as np

sys.path.insert(0, ".")
from lantern_amd import capi, synth  # noqa: E402

n, d, efc, metric, world = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], int(sys.argv[5])
rng = np.random.default_rng(77)
make = synth.query_maker("clustered", d)
base, queries = make(rng, n), make(rng, 2000)
labels = np.arange(n, dtype=np.uint64) + 1
b, q = base.astype(np.float64), queries.astype(np.float64)
dd = (q * q).sum(1)[:, None] - 2.0 * q @ b.T + (b * b).sum(1)[None, :]
truth = np.argsort(dd, axis=1, kind="stable")[:, :10]
centres = np.random.default_rng(synth.CLUSTER_SEED).standard_normal((synth.CLUSTERS, d), dtype=np.float32)
qc = np.argmin(((queries[:, None, :] - centres[None, :, :]) ** 2).sum(2), axis=1)


def recall(ix, ef):
    lab, _, _ = ix.search_batch(queries, 10, ef)
    per = np.array([len(set(lab[i].tolist()) & set((truth[i] + 1).tolist())) / 10 for i in range(len(queries))])
    by_cluster = [float(per[qc == c].mean()) for c in range(synth.CLUSTERS)]
    return float(per.mean()), min(by_cluster), int(np.argmin(by_cluster))


for seed in (21, 22, 23, 24, 25, 26):
    one = capi.GpuIndex(metric, d, M=16, ef_construction=efc, ef=64, seed=seed)
    one.add_many(labels, base)
    one.flush()
    comms = capi.Comm.local_world(world)
    out, errs = [None] * world, []

    def run(r):
        try:
            comms[r].set_timeout(600)
            ix = capi.GpuIndex(metric, d, M=16, ef_construction=efc, ef=64, seed=seed)
            lo, hi = n * r // world, n * (r + 1) // world
            ix.add_row_sharded(comms[r], labels[lo:hi], base[lo:hi])
            out[r] = ix
        except Exception as e:  # noqa: BLE001
            errs.append(repr(e))

    ts = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    [x.start() for x in ts]
    [x.join() for x in ts]
    assert not errs, errs
    a, b2 = recall(one, 64), recall(out[0], 64)
    print(f"seed {seed}: one GPU {a[0]:.4f} (worst cluster {a[1]:.3f} #{a[2]})   row-sharded x{world} {b2[0]:.4f} (worst cluster {b2[1]:.3f} #{b2[2]})", flush=True)
    del out, one
    [c.free() for c in comms]
