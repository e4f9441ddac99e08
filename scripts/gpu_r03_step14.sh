#!/bin/bash
# the latency-bound insert kernel: build / insert parity (the builds start from an empty index and pass through it), lone-insert latency
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r03step14; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_build_parity_production_batch.py tests/test_gpu_scans_and_inserts.py tests/test_gpu_sharded_build.py -q -x -p no:cacheprovider > $OUT/t.log 2>&1; echo "rc=$?" >> $OUT/t.log; tail -2 $OUT/t.log
timeout 300 python scripts/bench_single_insert.py > $OUT/r03_single_insert_100kx128.json 2> $OUT/ins.err; cat $OUT/r03_single_insert_100kx128.json
LANTERN_GPU_INSERT_SPEC=0 timeout 300 python scripts/bench_single_insert.py --no-cpu > $OUT/r03_single_insert_100kx128_classic.json 2>> $OUT/ins.err; cat $OUT/r03_single_insert_100kx128_classic.json
tail -2 $OUT/ins.err
