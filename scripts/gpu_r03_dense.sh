#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r03dense
rm -rf "$OUT"; mkdir -p "$OUT/dense"
timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_quantized_indexes.py -q -x -p no:cacheprovider -k "exact_search or mfma or assign_to_clusters or distance_matrix or pq_index" > "$OUT/t_dense.log" 2>&1; echo "rc=$?" >> "$OUT/t_dense.log"
timeout 200 rocprofv3 --kernel-trace --stats -d "$OUT/dense/trace" -o trace -- python scripts/bench_dense.py > "$OUT/dense_bench.json" 2> /dev/null
timeout 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 -d "$OUT/dense/pmc" -o pmc -- python scripts/bench_dense.py > /dev/null 2> /dev/null
python scripts/prof_dump.py "$OUT/dense" k_dense > "$OUT/r03_dense_mfma.md" 2>&1
rm -rf "$OUT/dense"
tail -n 3 "$OUT/t_dense.log"; head -20 "$OUT/r03_dense_mfma.md"; cat "$OUT/dense_bench.json"
