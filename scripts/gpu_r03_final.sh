#!/bin/bash
# the round's closing run: the whole GPU suite, smoke(), the default bench line
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r03final; mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $OUT/t_gpu.log 2>&1; echo "rc=$?" >> $OUT/t_gpu.log
tail -4 $OUT/t_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
timeout 600 python bench.py > $OUT/bench_line.json 2> $OUT/bench.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/r03final/bench_line.json'))
print({k:d[k] for k in ('value','ms_per_step','recall_at_10','build_vectors_per_s')}, {k:d['roofline'][k] for k in ('frac','frac_traffic','frac_of_measured_ceiling')}, d['cpu_baseline'] and d['cpu_baseline']['value'])
PY
