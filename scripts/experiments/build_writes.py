"""WRITE_SIZE (and FETCH_SIZE) per build kernel over a re-execution of the 1M x 768 build: are there writes the algorithmic count does not know about?
    python scripts/experiments/build_writes.py      (on the GPU box)"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import bench_pmc
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
child = [sys.executable, os.path.join(ROOT, "bench.py"), "--pmc-child", "--no-pmc", "--no-cpu", "--truth-queries", "0", "--build-quality-rows", "0", "--steps", "1", "--warmup", "1",
         "--queries", "64", "--query-batches", "1"] + sys.argv[1:]
out = {}
for ctr in ("WRITE_SIZE", "FETCH_SIZE"):
    r = bench_pmc.run_pass(child, "", "k_", [ctr], "pmc_child", timeout=600, by_kernel=True)
    if "error" in r:
        out[ctr] = r["error"]
        continue
    mult = 1024 * (2 if ctr == "FETCH_SIZE" else 1)
    out[ctr] = {k: {"launches": len(v[ctr]), "GB": float(np.sum(v[ctr])) * mult / 1e9} for k, v in sorted(r["values"].items(), key=lambda kv: -np.sum(kv[1][ctr]))[:14]}
print(json.dumps(out, indent=1))
