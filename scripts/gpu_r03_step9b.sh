#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r03step9; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_quantized_indexes.py -q -x -p no:cacheprovider > $OUT/t.log 2>&1; echo "rc=$?" >> $OUT/t.log
tail -3 $OUT/t.log
timeout 300 python bench.py --no-cpu --pq-subvectors 96 --data clustered > $OUT/r03_bench_line_pq96_compact_clustered.json 2>> $OUT/pq.err
timeout 300 python bench.py --no-cpu --pq-subvectors 96 > $OUT/r03_bench_line_pq96_compact.json 2>> $OUT/pq.err
timeout 300 python bench.py --no-cpu --pq-subvectors 32 --data clustered > $OUT/r03_bench_line_pq32_compact_clustered.json 2>> $OUT/pq.err
for f in $OUT/r03_bench_line_pq*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print(sys.argv[1].split('/')[-1], round(d['value']), d['ms_per_step'], d.get('recall_at_10'))
except Exception as e: print(sys.argv[1], 'unreadable', e)
PY
done
tail -3 $OUT/pq.err
