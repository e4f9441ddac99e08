#!/bin/bash
# round 3, GPU step 5: cosine over b1 + the re-laid-out dense kernel (parity), dense MFMA profile, counters of the build kernels
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r03s5
rm -rf "$OUT"; mkdir -p "$OUT"
timeout 300 python -m pytest tests/test_gpu_quantized_indexes.py -q -x -p no:cacheprovider -k "quant_bits_1 or bulk_adds" > "$OUT/t_b1.log" 2>&1; echo "rc=$?" >> "$OUT/t_b1.log"
timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -p no:cacheprovider -k "exact_search or mfma or assign_to_clusters or distance_matrix" > "$OUT/t_dense.log" 2>&1; echo "rc=$?" >> "$OUT/t_dense.log"
# ---- dense contraction: kernel trace + MFMA counters
mkdir -p "$OUT/dense"
timeout 200 rocprofv3 --kernel-trace --stats -d "$OUT/dense/trace" -o trace -- python scripts/bench_dense.py > "$OUT/dense_bench.json" 2> "$OUT/dense_trace.log"
timeout 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 -d "$OUT/dense/pmc" -o pmc -- python scripts/bench_dense.py > "$OUT/dense_bench_pmc.json" 2> "$OUT/dense_pmc.log"
python scripts/prof_dump.py "$OUT/dense" k_dense > "$OUT/r03_dense_mfma.md" 2>&1
# ---- the build's kernels: waves resident, busy cycles, fabric bytes (k_connect has had no counter-based account so far)
mkdir -p "$OUT/build"
timeout 300 rocprofv3 --kernel-include-regex "k_connect|k_insert|k_revlink" --pmc SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES -d "$OUT/build/pmc_sq" -o pmc -- python bench.py --no-cpu --steps 2 --warmup 1 > /dev/null 2> "$OUT/build_sq.log"
timeout 300 rocprofv3 --kernel-include-regex "k_connect|k_insert|k_revlink" --pmc FETCH_SIZE -d "$OUT/build/pmc_fetch" -o pmc -- python bench.py --no-cpu --steps 2 --warmup 1 > /dev/null 2> "$OUT/build_fetch.log"
python scripts/prof_dump.py "$OUT/build" k_ > "$OUT/r03_build_kernels_pmc.md" 2>&1
for d in dense build; do find "$OUT/$d" -type f -delete; find "$OUT/$d" -type d -empty -delete; done
for l in "$OUT"/*.log; do case "$l" in *t_*.log) ;; *) head -c 500 "$l" > "$l.head"; rm -f "$l";; esac; done
du -sh "$OUT"; tail -n 3 "$OUT"/t_*.log; cat "$OUT/dense_bench.json"
