#!/bin/bash
# Kernel trace of a build + SQ counters of the re-prune kernel (separate --pmc passes; gpurun refuses --pmc with other traces).
#   bash scripts/profile_reprune.sh [kernel regex] [extra bench.py args]
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
KRE="${1:-k_revlink_pairs}"; shift || true
OUT=gpurun_out/prof_reprune
rm -rf "$OUT"; mkdir -p "$OUT"
ARGS="--no-cpu --steps 2 --warmup 1 $*"
timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o trace -- python bench.py $ARGS > "$OUT/bench_trace.json" 2> "$OUT/trace.log"
timeout 300 rocprofv3 --kernel-include-regex "$KRE" --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT -d "$OUT/pmc_sq1" -o pmc -- python bench.py $ARGS > "$OUT/bench_pmc1.json" 2> "$OUT/pmc1.log"
timeout 300 rocprofv3 --kernel-include-regex "$KRE" --pmc SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVES SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SMEM GRBM_GUI_ACTIVE -d "$OUT/pmc_sq2" -o pmc -- python bench.py $ARGS > "$OUT/bench_pmc2.json" 2> "$OUT/pmc2.log"
python scripts/prof_dump.py "$OUT" > "$OUT/summary.md" 2>&1
cat "$OUT/summary.md"
tail -3 "$OUT/pmc1.log" "$OUT/pmc2.log"
