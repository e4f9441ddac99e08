"""`python tests/fake_rccl/run_world.py WORLD` in a process of its own, with LANTERN_GPU_RCCL_LIB pointing at the test double: WORLD
ranks as threads of this process run the sharded builds through the RCCL transport of lantern_amd/csrc/comm.cpp -- unique id,
ncclCommInitRank per rank, the metadata exchange (allgatherv_host over the device path), the grouped-broadcast all-gather-v with
ragged segment sizes on the index streams, the deadline poll -- and print one JSON line the test asserts on.
(A process of its own: comm.cpp binds its RCCL once per process.)"""
import json
import os
import sys
import threading

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lantern_amd import capi, hip  # noqa: E402


def threads(world, fn):
    out, errs = [None] * world, []

    def run(r):
        try:
            out[r] = fn(r)
        except Exception as e:  # noqa: BLE001
            errs.append((r, repr(e)))

    ts = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    [t.start() for t in ts]
    [t.join(300) for t in ts]
    return out, errs


def main():
    world = int(sys.argv[1])
    res = {"world": world, "rccl_lib": os.environ.get("LANTERN_GPU_RCCL_LIB")}
    uid = capi.Comm.unique_id()
    res["uid_is_the_doubles"] = uid[:8] == b"FAKERCCL"
    comms, errs = threads(world, lambda r: capi.Comm.rccl(r, world, uid))  # ncclCommInitRank is a rendezvous: one thread per rank
    res["comm_errors"] = errs
    if errs:
        print(json.dumps(res))
        return
    [c.set_timeout(120) for c in comms]
    res["rccl_ranks_seen"] = sum(1 for c in comms if c.world == world)

    # ---- the exchange primitive itself: ragged segments, one empty, on a stream per rank
    seg = [((r * 977) % 5000) + (0 if r == 1 else 1) * 37 for r in range(world)]
    if world > 1:
        seg[1] = 0
    off = np.concatenate([[0], np.cumsum(seg)]).astype(int)
    want = np.concatenate([np.full(seg[r], r + 1, dtype=np.uint8) for r in range(world)]) if off[-1] else np.zeros(0, np.uint8)

    def exchange(r):
        st = hip.Stream()
        mine = np.zeros(max(int(off[-1]), 1), dtype=np.uint8)
        mine[off[r]:off[r + 1]] = r + 1
        buf = hip.Buffer.from_numpy(mine)
        for _ in range(3):
            comms[r].allgatherv_device(buf.ptr, [int(x) for x in off[:-1]], [int(x) for x in seg], st.handle)
        return bool(np.array_equal(buf.download(int(off[-1]), np.uint8), want)) if off[-1] else True

    ok, errs = threads(world, exchange)
    res["allgatherv_ok"], res["allgatherv_errors"] = ok, errs

    # ---- the work-sharded build: replicas bit-identical to the one-GPU graph; ragged shards, one of them empty at world >= 3
    n, d, M, efc, plan = 2400, 96, 8, 40, (256, 8)
    rng = np.random.default_rng(77)
    base = rng.standard_normal((n, d), dtype=np.float32)
    labels = np.arange(n, dtype=np.uint64) + 1
    ref = capi.GpuIndex("l2sq", d, M=M, ef_construction=efc, ef=32, seed=21)
    ref.set_add_batch(*plan)
    ref.add_many(labels, base)
    ref.flush()
    cuts = sorted(int(x) for x in rng.choice(np.arange(1, n), size=world - 1, replace=False)) if world > 1 else []
    cuts = [0] + cuts + [n]
    if world >= 3:
        cuts[2] = cuts[1]  # rank 1 contributes nothing and still takes its share of the work
    res["shards"] = [cuts[r + 1] - cuts[r] for r in range(world)]

    def work_sharded(r):
        ix = capi.GpuIndex("l2sq", d, M=M, ef_construction=efc, ef=32, seed=21)
        ix.set_add_batch(*plan)
        ix.add_sharded(comms[r], labels[cuts[r]:cuts[r + 1]], base[cuts[r]:cuts[r + 1]])
        ix.flush()
        return {"checksum": f"{ix.checksum():016x}", "size": len(ix), "add_dist_evals": ix.counters()["add_dist_evals"]}

    out, errs = threads(world, work_sharded)
    res["work_sharded"] = {"errors": errs, "ranks": out, "reference_checksum": f"{ref.checksum():016x}", "reference_add_dist_evals": ref.counters()["add_dist_evals"]}

    # ---- the row-sharded build (SURVEY 8e as written): every rank's replica identical, recall as the one-GPU build's
    queries = rng.standard_normal((200, d), dtype=np.float32)
    truth, _ = ref.exact_search(queries, 10)

    def recall(ix):
        lab, _, _ = ix.search_batch(queries, 10)
        return float(np.mean([len(set((l.astype(np.int64) - 1).tolist()) & set(t.tolist())) / 10 for l, t in zip(lab, truth)]))

    def row_sharded(r):
        ix = capi.GpuIndex("l2sq", d, M=M, ef_construction=efc, ef=32, seed=21)
        ix.set_add_batch(*plan)
        ix.add_row_sharded(comms[r], labels[cuts[r]:cuts[r + 1]], base[cuts[r]:cuts[r + 1]])
        ix.flush()
        return {"checksum": f"{ix.checksum():016x}", "size": len(ix), "recall": recall(ix)}

    out, errs = threads(world, row_sharded)
    res["row_sharded"] = {"errors": errs, "ranks": out, "one_gpu_recall": recall(ref)}
    res["stats"] = [c.stats() for c in comms]
    [c.free() for c in comms]
    print(json.dumps(res))


if __name__ == "__main__":
    main()
