#!/bin/bash
# clock and matrix-pipe occupancy of the harness variants: GRBM_GUI_ACTIVE / duration = clock, MFMA busy / (SIMDs * cycles) = occupancy
export TMPDIR=/tmp
OUT=gpurun_out/dense_variants
mkdir -p $OUT
for v in "$@"; do
  rm -rf $OUT/prof_$v
  timeout 120 rocprofv3 --kernel-trace --stats -d $OUT/prof_$v/trace -o trace -- scripts/experiments/bin/dv_$v > /dev/null 2>&1
  timeout 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES -d $OUT/prof_$v/pmc -o pmc -- scripts/experiments/bin/dv_$v > /dev/null 2>&1
  echo "## dv_$v" >> $OUT/clocks.md
  python scripts/prof_dump.py $OUT/prof_$v k_dense >> $OUT/clocks.md 2>&1
  rm -rf $OUT/prof_$v
done
cat $OUT/clocks.md
