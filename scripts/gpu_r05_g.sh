#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/r05g; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_scan_server.py -m gpu -x -q -p no:cacheprovider > $OUT/service_tests.log 2>&1; echo "rc $?" >> $OUT/service_tests.log
timeout 600 python scripts/scan_load_headline.py --connections 1,16,64,256,1024 > $OUT/scan_load.jsonl 2> $OUT/scan_load.err
LANTERN_BENCH_SECONDARY=headline_scan_service timeout 600 python bench.py --no-pmc --no-cpu --build-quality-rows 0 > $OUT/bench_service_only.json 2> $OUT/bench_service_only.err
LANTERN_BENCH_SECONDARY=headline_scan_service timeout 600 python bench.py --no-pmc --no-cpu --build-quality-rows 0 --add-batch 8192 > $OUT/bench_service_only_plan8192.json 2> $OUT/bench_service_only2.err
