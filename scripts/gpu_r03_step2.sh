#!/bin/bash
# round 3, GPU step 2: the latency-bound walk (parity + timings), the services end to end, the clustered set
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r03
mkdir -p "$OUT"
timeout 420 python -m pytest tests/test_gpu_parity.py -q -x -p no:cacheprovider -k "latency_bound or lone_query or gather_in_the_walks or i8_rows_wider" > "$OUT/t_spec.log" 2>&1; echo "rc=$?" >> "$OUT/t_spec.log"
timeout 200 python -m pytest tests/test_gpu_scans_and_inserts.py -q -x -p no:cacheprovider -k "mirror" > "$OUT/t_mirror.log" 2>&1; echo "rc=$?" >> "$OUT/t_mirror.log"
timeout 120 python scripts/bench_single_query.py > "$OUT/r03_single_query_100kx128.json" 2> "$OUT/single.err"
timeout 200 python bench.py --no-cpu --metric cos --queries 1024 --steps 40 > "$OUT/r03_bench_line_cos_q1024.json" 2> "$OUT/q1024.err"
LANTERN_GPU_SPEC=0 timeout 200 python bench.py --no-cpu --metric cos --queries 1024 --steps 40 > "$OUT/r03_bench_line_cos_q1024_classic.json" 2>> "$OUT/q1024.err"
for c in 64 256 1024; do timeout 60 lantern_amd/lib/lantern-scan-load --connections $c --seconds 4 >> "$OUT/r03_scan_load_100kx128.jsonl" 2>> "$OUT/scanload.err"; done
timeout 200 lantern_amd/lib/lantern-index-load --rows 1000000 --dim 1536 > "$OUT/r03_index_load_1Mx1536.json" 2> "$OUT/indexload.err"
timeout 300 python bench.py --no-cpu --data clustered > "$OUT/r03_bench_line_clustered.json" 2> "$OUT/clustered.err"
timeout 300 python -m pytest tests/test_gpu_fullsize.py -q -x -p no:cacheprovider -k "clustered" -s > "$OUT/t_clustered.log" 2>&1; echo "rc=$?" >> "$OUT/t_clustered.log"
tail -3 "$OUT"/t_*.log
