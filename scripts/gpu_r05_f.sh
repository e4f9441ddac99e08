#!/bin/bash
# round 5, sixth GPU call: the whole -m gpu suite, the bench line (default command), the kernel trace of the same command, the scan
# service's per-leg timing at the headline shape
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/r05f; mkdir -p $OUT
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider --durations=15 ) > $OUT/gpu_suite.log 2>&1
echo "rc $?" >> $OUT/gpu_suite.log
( time timeout 900 python bench.py ) > $OUT/bench_line.json 2> $OUT/bench.err
echo "rc $?" >> $OUT/bench.err
rm -rf gpurun_out/trace_only
bash scripts/trace_only.sh --no-secondary --build-quality-rows 0 --no-pmc > $OUT/trace_summary.txt 2>&1
cp -r gpurun_out/trace_only/summary $OUT/trace_summary 2>/dev/null
cp gpurun_out/trace_only/bench_trace.json $OUT/bench_trace.json 2>/dev/null
timeout 600 python scripts/scan_load_headline.py --connections 16,64,256,1024 > $OUT/scan_load_timing.jsonl 2> $OUT/scan_load.err
