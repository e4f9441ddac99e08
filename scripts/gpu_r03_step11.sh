#!/bin/bash
# per-role instantiations of the latency-bound walk: parity of every launch shape, the lone query, the ADC lines
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r03step11; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_quantized_indexes.py tests/test_gpu_baseline_configs.py -q -x -p no:cacheprovider > $OUT/t.log 2>&1; echo "rc=$?" >> $OUT/t.log; tail -3 $OUT/t.log
timeout 200 python scripts/bench_single_query.py > $OUT/r03_single_query_100kx128.json 2> $OUT/single.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r03step11/r03_single_query_100kx128.json')); print(d['us_per_query_wall'], d['kernel_only']['latency_bound_shape'], d['identical_to_batch_search'])
PY
timeout 300 python bench.py --no-cpu --pq-subvectors 96 --data clustered > $OUT/r03_bench_line_pq96_compact_clustered.json 2>> $OUT/pq.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r03step11/r03_bench_line_pq96_compact_clustered.json')); print('pq96 clustered', round(d['value']), d['recall_at_10'])
PY
