#!/bin/bash
# scripts/gpu.sh -- the GPU-box jobs of this repo, one entry per job (run under gpurun: `gpurun --timeout N -- bash scripts/gpu.sh JOB`).
# Everything a job writes goes under gpurun_out/<job>/ ; summaries worth keeping are copied into profiles/ by hand.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
JOB="${1:-help}"; shift || true
OUT="gpurun_out/$JOB"; mkdir -p "$OUT"
case "$JOB" in
  tests)        # the whole -m gpu suite
    timeout 1500 python -m pytest tests -m gpu -x -q --durations=25 2>&1 | tail -60 > "$OUT/pytest.log"; tail -40 "$OUT/pytest.log" ;;
  tests-new)    # a named subset: bash scripts/gpu.sh tests-new "expr for -k"
    timeout 900 python -m pytest tests -m gpu -x -q -k "$1" 2>&1 | tail -40 | tee "$OUT/pytest.log" ;;
  bench)        # the driver's command; extra flags pass through
    LANTERN_BENCH_PMC_LOG="$OUT" timeout ${BENCH_TIMEOUT:-1200} python bench.py "$@" > "$OUT/line.json" 2> "$OUT/stderr.log"; echo "rc=$?"; tail -5 "$OUT/stderr.log"; head -c 600 "$OUT/line.json" ;;
  bench-trace)  # rocprofv3 --kernel-trace --stats of the search leg only (no counters, no secondary legs)
    timeout 900 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o trace -- python bench.py --no-pmc --no-secondary --no-cpu --no-dram-model --build-quality-rows 0 "$@" > "$OUT/line.json" 2> "$OUT/trace.log"
    python scripts/prof_dump.py "$OUT/trace" > "$OUT/kernel_stats.md"; head -30 "$OUT/kernel_stats.md" ;;
  *) echo "jobs: tests | tests-new EXPR | bench [flags] | bench-trace [flags]" ;;
esac
