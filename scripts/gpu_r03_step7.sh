#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r03s7
rm -rf "$OUT"; mkdir -p "$OUT"
timeout 100 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1; echo "rc=$?" >> "$OUT/smoke.log"
timeout 100 python scripts/bench_dense.py > "$OUT/r03_dense_bench.json" 2> "$OUT/dense.err"
timeout 100 python scripts/bench_dense.py l2sq >> "$OUT/r03_dense_bench.json" 2>> "$OUT/dense.err"
timeout 300 python bench.py --no-cpu --pq-subvectors 96 > "$OUT/r03_bench_line_pq96_compact.json" 2> "$OUT/pq.err"
timeout 300 python bench.py --no-cpu --pq-subvectors 96 --data clustered > "$OUT/r03_bench_line_pq96_compact_clustered.json" 2>> "$OUT/pq.err"
timeout 300 python bench.py --no-cpu --pq-subvectors 32 --data clustered > "$OUT/r03_bench_line_pq32_compact_clustered.json" 2>> "$OUT/pq.err"
tail -n 2 "$OUT/smoke.log"; cat "$OUT/r03_dense_bench.json"; tail -c 600 "$OUT/pq.err"
