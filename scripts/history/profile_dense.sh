#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
OUT=gpurun_out/dense; rm -rf "$OUT"; mkdir -p "$OUT"
timeout 200 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o trace -- python scripts/bench_dense.py > "$OUT/bench.json" 2> "$OUT/trace.log"
timeout 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 -d "$OUT/pmc" -o pmc -- python scripts/bench_dense.py > "$OUT/bench_pmc.json" 2> "$OUT/pmc.log"
cat "$OUT/bench.json"; tail -2 "$OUT/pmc.log"
python - <<'PY'
import glob, sqlite3
for f in glob.glob("gpurun_out/dense/trace/**/*.db", recursive=True):
    cur = sqlite3.connect(f).cursor()
    for r in cur.execute("select name,total_calls,total_duration,average from top_kernels limit 6"): print(r)
for f in glob.glob("gpurun_out/dense/pmc/**/*.db", recursive=True):
    cur = sqlite3.connect(f).cursor()
    for r in cur.execute("select counter_name,count(*),avg(value) from counters_collection where kernel_name like '%k_dense_f32%' group by counter_name"): print(r)
PY
