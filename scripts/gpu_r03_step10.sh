#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r03step10; mkdir -p $OUT; rm -f $OUT/r03_scan_load_100kx128.jsonl
timeout 400 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "scan or service" > $OUT/t.log 2>&1; echo "rc=$?" >> $OUT/t.log; tail -3 $OUT/t.log
for c in 16 64 256 512; do timeout 60 lantern_amd/lib/lantern-scan-load --connections $c --seconds 4 >> $OUT/r03_scan_load_100kx128.jsonl 2>> $OUT/scanload.err; done
python - <<'PY'
import json
for l in open('gpurun_out/r03step10/r03_scan_load_100kx128.jsonl'):
    d=json.loads(l); print(d['connections'], round(d['queries_per_s']), d['latency_us'], d['service']['mean_batch'])
PY
