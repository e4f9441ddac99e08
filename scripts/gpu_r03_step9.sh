#!/bin/bash
# ADC search with the latency-bound walk: parity tests, then the compact bench lines with either walk
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r03step9; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_quantized_indexes.py -q -x -p no:cacheprovider > $OUT/t.log 2>&1; echo "rc=$?" >> $OUT/t.log
tail -3 $OUT/t.log
for spec in 1 0; do
  LANTERN_GPU_ADC_SPEC=$spec timeout 300 python bench.py --no-cpu --pq-subvectors 96 --data clustered > $OUT/pq96_clustered_spec$spec.json 2>> $OUT/pq.err
  LANTERN_GPU_ADC_SPEC=$spec timeout 300 python bench.py --no-cpu --pq-subvectors 32 --data clustered > $OUT/pq32_clustered_spec$spec.json 2>> $OUT/pq.err
done
for f in $OUT/pq*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print(sys.argv[1].split('/')[-1], round(d['value']), d['ms_per_step'], d.get('recall_at_10'))
except Exception as e: print(sys.argv[1], 'unreadable', e)
PY
done
tail -3 $OUT/pq.err
