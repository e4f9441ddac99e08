#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/r05h; mkdir -p $OUT
export TMPDIR=/tmp
LANTERN_BENCH_SECONDARY=headline_scan_service timeout 600 python bench.py --no-pmc --no-cpu --build-quality-rows 0 --steps 3 > $OUT/bench_service_only.json 2> $OUT/e1.err
timeout 600 python scripts/scan_load_headline.py --connections 256 > $OUT/scan_load_256.jsonl 2> $OUT/e2.err
