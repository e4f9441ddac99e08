#!/usr/bin/env python
"""Where a hop of the ONE-WAVE walk (lantern_amd/csrc/walk_solo.hpp, LANTERN_GPU_SPEC=4) spends its time: the instrumented
instantiation on the lone 100k x 128 query of BASELINE config[1] -- shader-clock cycles per section of a hop, executed by ONE
wave with no barrier anywhere: the dependent chain of a hop, section by section -- beside the same walk's plain timing and the
3 + 8 wave shape's (LANTERN_GPU_SPEC=2).

    python scripts/profile_solo_hops.py > profiles/r05_one_wave_hop_phases.json
"""
import ctypes as C
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lantern_amd import capi, hip  # noqa: E402

SECTIONS = ["list look-up (LDS cache, else HBM)", "issuing 16 row + 4 list (+ norm) loads", "merge of the previous hop's keys + expanded mark",
            "visited filter (ds_or_rtn) + cache claims", "wait for the rows + 8 x 2 chunk chains + 4 DPP steps + keys", "cache writes", "next node"]


def timed(ix, dq, row_bytes, n, st, lab, dst):
    e0, e1 = hip.Event(), hip.Event()
    for i in range(20):
        ix.search_batch_device(dq.ptr + i * row_bytes, 1, 10, 64, 0, lab.ptr, dst.ptr, None, None, None, None, st.handle)
    st.synchronize()
    e0.record(st.handle)
    for i in range(n):
        ix.search_batch_device(dq.ptr + i * row_bytes, 1, 10, 64, 0, lab.ptr, dst.ptr, None, None, None, None, st.handle)
    e1.record(st.handle)
    st.synchronize()
    return e0.elapsed_ms(e1) * 1e3 / n


def main():
    base = np.random.default_rng(1).standard_normal((100_000, 128), dtype=np.float32)
    q = np.random.default_rng(2).standard_normal((1000, 128), dtype=np.float32)
    ix = capi.GpuIndex("l2sq", 128, M=16, ef_construction=128, ef=64, seed=42)
    ix.add_many(np.arange(base.shape[0], dtype=np.uint64) + 1, base)
    ix.flush()
    dq = hip.Buffer.from_numpy(hip.padded_rows(q, False))
    row_bytes = hip.padded_rows(q[:1], False).shape[1] * 4
    lab, dst = hip.Buffer(80), hip.Buffer(40)
    st = hip.Stream()
    ix.set_search_shape(0, 0)
    out = {"config": "100000x128 f32 l2sq M=16 efc=128 ef=64 k=10, one query per launch, 1000 launches"}
    os.environ["LANTERN_GPU_SPEC"] = "2"
    out["three_role_plus_eight_row_waves_us_per_query"] = timed(ix, dq, row_bytes, 1000, st, lab, dst)
    os.environ["LANTERN_GPU_SPEC"] = "4"
    before = ix.counters()["search_solo_launches"]
    out["one_wave_us_per_query"] = timed(ix, dq, row_bytes, 1000, st, lab, dst)
    assert ix.counters()["search_solo_launches"] == before + 1020
    raw = (C.c_ulonglong * 32)()
    capi._call("lantern_gpu_spec_profile", ix.h, 1, raw)
    out["one_wave_us_per_query_instrumented"] = timed(ix, dq, row_bytes, 1000, st, lab, dst)
    capi._call("lantern_gpu_spec_profile", ix.h, 0, raw)
    p = [int(x) for x in raw]
    hops = max(p[7], 1)
    out["hops"] = hops
    out["hops_per_query"] = hops / 1020
    out["cycles_per_hop_by_section"] = {SECTIONS[i]: p[i] / hops for i in range(7)}
    out["cycles_per_hop_total_instrumented"] = sum(p[:7]) / hops
    out["list_cache_miss_rate"] = p[8] / hops
    out["note"] = ("ONE wave executes every section in order, no barrier: this is the dependent chain of a hop, cycle by cycle.  "
                   "Each stamp adds an s_memtime + s_waitcnt (~40 cycles).")
    print(json.dumps(out))


if __name__ == "__main__":
    main()
