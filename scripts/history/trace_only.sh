#!/bin/bash
# kernel-trace --stats only (one pass).  bash scripts/trace_only.sh [bench args]
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
OUT=gpurun_out/trace_only; rm -rf "$OUT"; mkdir -p "$OUT"
timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o trace -- python bench.py --no-cpu --steps 3 --warmup 1 "$@" > "$OUT/bench_trace.json" 2> "$OUT/trace.log"
python scripts/summarize_prof.py "$OUT" | head -20
