#!/bin/bash
# Round-2 measurement pass (run through gpurun): rocprofv3 kernel trace + separate PMC passes of the headline bench,
# the gather-ceiling calibration, and the secondary configs.  Outputs under gpurun_out/r02/; summaries are copied to profiles/.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r02
rm -rf "$OUT"; mkdir -p "$OUT/bench" "$OUT/gather"
ARGS="--no-cpu --steps 5 --warmup 2"
# ---- headline bench: kernel trace, then FETCH / WRITE / L2 passes (each its own run)
timeout 400 rocprofv3 --kernel-trace --stats -d "$OUT/bench/trace" -o trace -- python bench.py $ARGS > "$OUT/bench/bench_trace.json" 2> "$OUT/bench_trace.log"
timeout 400 rocprofv3 --kernel-include-regex k_search --pmc FETCH_SIZE -d "$OUT/bench/pmc_fetch" -o pmc -- python bench.py $ARGS > /dev/null 2> "$OUT/bench_pmc_fetch.log"
timeout 400 rocprofv3 --kernel-include-regex k_search --pmc WRITE_SIZE -d "$OUT/bench/pmc_write" -o pmc -- python bench.py $ARGS > /dev/null 2> "$OUT/bench_pmc_write.log"
timeout 400 rocprofv3 --kernel-include-regex k_search --pmc TCC_HIT_sum TCC_MISS_sum -d "$OUT/bench/pmc_l2" -o pmc -- python bench.py $ARGS > /dev/null 2> "$OUT/bench_pmc_l2.log"
python scripts/summarize_prof.py "$OUT/bench" "$OUT/r02_bench_1Mx768" > "$OUT/bench_summary.txt" 2>&1
# ---- the cosine line of the same workload: fabric traffic next to its algorithmic bytes (is the lower fraction fewer cache hits?)
mkdir -p "$OUT/cos"
timeout 400 rocprofv3 --kernel-trace --stats -d "$OUT/cos/trace" -o trace -- python bench.py $ARGS --metric cos > "$OUT/cos/bench_trace.json" 2> "$OUT/cos_trace.log"
timeout 400 rocprofv3 --kernel-include-regex k_search --pmc FETCH_SIZE -d "$OUT/cos/pmc_fetch" -o pmc -- python bench.py $ARGS --metric cos > /dev/null 2> "$OUT/cos_pmc_fetch.log"
timeout 400 rocprofv3 --kernel-include-regex k_search --pmc TCC_HIT_sum TCC_MISS_sum -d "$OUT/cos/pmc_l2" -o pmc -- python bench.py $ARGS --metric cos > /dev/null 2> "$OUT/cos_pmc_l2.log"
python scripts/summarize_prof.py "$OUT/cos" "$OUT/r02_bench_1Mx768_cos" > "$OUT/cos_summary.txt" 2>&1
# ---- gather ceiling
timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/gather/trace" -o trace -- python scripts/bench_gather_ceiling.py > "$OUT/gather.json" 2> "$OUT/gather_trace.log"
timeout 300 rocprofv3 --kernel-include-regex k_gather --pmc FETCH_SIZE -d "$OUT/gather/pmc_fetch" -o pmc -- python scripts/bench_gather_ceiling.py > /dev/null 2> "$OUT/gather_pmc_fetch.log"
timeout 300 rocprofv3 --kernel-include-regex k_gather --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -d "$OUT/gather/pmc_l2" -o pmc -- python scripts/bench_gather_ceiling.py > /dev/null 2> "$OUT/gather_pmc_l2.log"
python scripts/prof_dump.py "$OUT/gather" k_gather > "$OUT/r02_gather_ceiling.md" 2>&1
# ---- the unprofiled lines
python bench.py > "$OUT/r02_bench_line.json" 2> "$OUT/bench_line.err"
python bench.py --no-cpu --metric cos --queries 1024 --steps 40 > "$OUT/r02_bench_line_cos_q1024.json" 2>/dev/null
python bench.py --no-cpu --metric cos --queries 1024 --steps 40 --streams 2 > "$OUT/r02_bench_line_cos_q1024_2streams.json" 2>/dev/null
python bench.py --no-cpu --streams 2 > "$OUT/r02_bench_line_2streams.json" 2>/dev/null
python bench.py --no-cpu --metric cos > "$OUT/r02_bench_line_cos.json" 2>/dev/null
python bench.py --no-cpu --data lowrank > "$OUT/r02_bench_line_lowrank.json" 2>/dev/null
python bench.py --no-cpu --quant f16 > "$OUT/r02_bench_line_f16.json" 2>/dev/null
python bench.py --no-cpu --quant i8 --data-scale 0.3 > "$OUT/r02_bench_line_i8.json" 2>/dev/null
python bench.py --no-cpu --dim 1536 --steps 5 > "$OUT/r02_bench_line_1Mx1536.json" 2>/dev/null
python scripts/bench_single_query.py > "$OUT/r02_single_query_100kx128.json" 2>/dev/null
python scripts/profile_hop_phases.py > "$OUT/r02_hop_phases.json" 2>/dev/null
METRIC768=cos python scripts/profile_hop_phases.py > "$OUT/r02_hop_phases_cos.json" 2>/dev/null
python scripts/ab_list_placement.py > "$OUT/r02_list_placement_ab.json" 2>/dev/null
python scripts/bench_configs.py > "$OUT/r02_configs.jsonl" 2>/dev/null
python scripts/bench_scan_server.py > "$OUT/r02_scan_server_100kx128.json" 2>/dev/null
python bench.py --gpus 2 --dist-backend files --no-cpu --rows 500000 --steps 10 > "$OUT/r02_bench_line_2ranks_one_gpu_files.json" 2>/dev/null
python bench.py --no-cpu --quant b1 > "$OUT/r02_bench_line_b1.json" 2>/dev/null
timeout 900 python bench.py --rows 10000000 --ef 128 --steps 5 --no-cpu --truth-queries 256 > "$OUT/r02_bench_line_10Mx768_ef128.json" 2>/dev/null
ls -la "$OUT"
