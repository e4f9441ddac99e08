#!/bin/bash
# the round's closing run: the whole GPU suite, smoke(), the default bench line, the lone query, the scan service
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r03final; mkdir -p $OUT; rm -f $OUT/*.jsonl
timeout 1200 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $OUT/t_gpu.log 2>&1; echo "rc=$?" >> $OUT/t_gpu.log
grep -n "passed\|failed\|error" $OUT/t_gpu.log | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
timeout 600 python bench.py > $OUT/bench_line.json 2> $OUT/bench.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/r03final/bench_line.json'))
print({k:d[k] for k in ('value','ms_per_step','recall_at_10','build_vectors_per_s')}, {k:d['roofline'][k] for k in ('frac','frac_traffic','frac_of_measured_ceiling')}, d['cpu_baseline'] and d['cpu_baseline']['value'])
PY
for c in 64 256 512; do timeout 60 lantern_amd/lib/lantern-scan-load --connections $c --seconds 3 >> $OUT/r03_scan_load_100kx128.jsonl 2>> $OUT/scanload.err; done
for c in 256 1024; do timeout 60 lantern_amd/lib/lantern-scan-load --connections $c --client-threads 4 --seconds 3 >> $OUT/r03_scan_load_100kx128_multiplexed.jsonl 2>> $OUT/scanload.err; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03final/*.jsonl')):
    for l in open(f):
        d=json.loads(l); print(f.split('/')[-1][:40], d['connections'], d.get('client_threads'), round(d['queries_per_s']), d['latency_us']['mean'], d['latency_us']['p99'], d['service']['mean_batch'], d['failures'])
PY
