#!/bin/bash
# the scan service under load: thread-per-connection clients (what backends are) and multiplexed clients (what the service can take)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r03step10; mkdir -p $OUT; rm -f $OUT/*.jsonl
timeout 400 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "scan or service" > $OUT/t.log 2>&1; echo "rc=$?" >> $OUT/t.log; tail -3 $OUT/t.log
for c in 64 256 512; do timeout 60 lantern_amd/lib/lantern-scan-load --connections $c --seconds 3 >> $OUT/r03_scan_load_100kx128.jsonl 2>> $OUT/scanload.err; done
for c in 64 256 1024; do timeout 60 lantern_amd/lib/lantern-scan-load --connections $c --client-threads 4 --seconds 3 >> $OUT/r03_scan_load_100kx128_multiplexed.jsonl 2>> $OUT/scanload.err; done
LANTERN_SCAN_ONE_AT_A_TIME=1 timeout 60 lantern_amd/lib/lantern-scan-load --connections 256 --seconds 3 >> $OUT/one_at_a_time.jsonl 2>> $OUT/scanload.err
LANTERN_SCAN_ONE_AT_A_TIME=1 timeout 60 lantern_amd/lib/lantern-scan-load --connections 256 --client-threads 4 --seconds 3 >> $OUT/one_at_a_time.jsonl 2>> $OUT/scanload.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03step10/*.jsonl')):
    print(f.split('/')[-1])
    for l in open(f):
        d=json.loads(l); print('  ', d['connections'], d.get('client_threads'), round(d['queries_per_s']), d['latency_us'], d['service']['mean_batch'], d['failures'])
PY
tail -2 $OUT/scanload.err
