#!/bin/bash
# Run on the GPU box (through gpurun): kernel-trace stats and, in SEPARATE passes, the PMC counters
# (gpurun refuses --pmc combined with other trace domains; the guide wants counters in their own run).
#   bash scripts/profile_bench.sh [extra bench.py args]
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/prof
rm -rf "$OUT"; mkdir -p "$OUT"
ARGS="--no-cpu --steps 5 --warmup 2 $*"
timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o trace -- python bench.py $ARGS > "$OUT/bench_trace.json" 2> "$OUT/trace.log"
timeout 300 rocprofv3 --pmc FETCH_SIZE -d "$OUT/pmc_fetch" -o pmc -- python bench.py $ARGS > "$OUT/bench_pmc_fetch.json" 2> "$OUT/pmc_fetch.log"
timeout 300 rocprofv3 --pmc WRITE_SIZE -d "$OUT/pmc_write" -o pmc -- python bench.py $ARGS > "$OUT/bench_pmc_write.json" 2> "$OUT/pmc_write.log"
timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -d "$OUT/pmc_l2" -o pmc -- python bench.py $ARGS > "$OUT/bench_pmc_l2.json" 2> "$OUT/pmc_l2.log"
python scripts/summarize_prof.py "$OUT"
