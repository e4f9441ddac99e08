#!/bin/bash
# round 3, GPU step 3: one-barrier descent + ADC + mirror fix (parity), hop sections of the latency-bound walk, shape sweep of the
# 1024-query cosine batch, the indexing server with the buffered socket reader, the clustered set
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r03
mkdir -p "$OUT"
timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -p no:cacheprovider -k "latency_bound or lone_query" > "$OUT/t3_spec.log" 2>&1; echo "rc=$?" >> "$OUT/t3_spec.log"
timeout 200 python -m pytest tests/test_gpu_scans_and_inserts.py -q -x -p no:cacheprovider -k "mirror" > "$OUT/t3_mirror.log" 2>&1; echo "rc=$?" >> "$OUT/t3_mirror.log"
timeout 300 python -m pytest tests/test_gpu_quantized_indexes.py -q -x -p no:cacheprovider -k "compact" > "$OUT/t3_adc.log" 2>&1; echo "rc=$?" >> "$OUT/t3_adc.log"
timeout 200 python scripts/profile_spec_hops.py > "$OUT/r03_spec_hop_phases.json" 2> "$OUT/spec_hops.err"
timeout 120 python scripts/bench_single_query.py > "$OUT/r03_single_query_100kx128.json" 2> "$OUT/single.err"
timeout 200 lantern_amd/lib/lantern-index-load --rows 1000000 --dim 1536 > "$OUT/r03_index_load_1Mx1536.json" 2> "$OUT/indexload.err"
for sw in "1 4" "1 8" "2 11" "2 7"; do set -- $sw
  LANTERN_GPU_SPEC=$1 LANTERN_GPU_SPEC_WAVES=$2 timeout 200 python bench.py --no-cpu --metric cos --queries 1024 --steps 40 > "$OUT/r03_q1024_cos_spec$1_w$2.json" 2>> "$OUT/q1024.err"
done
timeout 300 python bench.py --no-cpu --data clustered > "$OUT/r03_bench_line_clustered.json" 2> "$OUT/clustered.err"
timeout 300 python bench.py --no-cpu --data clustered --metric cos > "$OUT/r03_bench_line_clustered_cos.json" 2>> "$OUT/clustered.err"
timeout 400 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_baseline_configs.py -q -x -p no:cacheprovider -k "clustered" -s > "$OUT/t3_clustered.log" 2>&1; echo "rc=$?" >> "$OUT/t3_clustered.log"
tail -n 3 "$OUT"/t3_*.log
