#!/bin/bash
# sanity of the shipped library after the last rebuild + the compact pq96 line on the prescribed (Gaussian) set
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r03last; mkdir -p $OUT
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_build_parity_production_batch.py -q -x -p no:cacheprovider > $OUT/t.log 2>&1; echo "rc=$?" >> $OUT/t.log; tail -2 $OUT/t.log
timeout 300 python bench.py --no-cpu --pq-subvectors 96 > $OUT/r03_bench_line_pq96_compact.json 2>> $OUT/err.log
python - <<'PY'
import json
d=json.load(open('gpurun_out/r03last/r03_bench_line_pq96_compact.json')); print('pq96 gaussian', round(d['value']), d['recall_at_10'], d['roofline']['kernel'])
PY
