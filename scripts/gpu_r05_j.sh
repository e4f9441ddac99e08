#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/r05j; mkdir -p $OUT
export TMPDIR=/tmp
GPU_MAX_HW_QUEUES=16 timeout 600 python scripts/scan_load_headline.py --connections 64,256,1024 > $OUT/scan_load_q16.jsonl 2> $OUT/e1.err
GPU_MAX_HW_QUEUES=16 LANTERN_SCAN_LANES=8 timeout 600 python scripts/scan_load_headline.py --connections 64,256,1024 > $OUT/scan_load_q16_8lanes.jsonl 2> $OUT/e2.err
GPU_MAX_HW_QUEUES=16 LANTERN_BENCH_SECONDARY=headline_scan_service timeout 600 python bench.py --no-pmc --no-cpu --build-quality-rows 0 --steps 3 > $OUT/bench_service_only_q16.json 2> $OUT/e3.err
