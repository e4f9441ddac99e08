#!/bin/bash
# Round-3 measurement pass (run through gpurun): rocprofv3 kernel trace + separate PMC passes of the headline bench and of its
# cosine line (incl. the DRAM-destined share of the fabric reads), the walk-shaped gather calibration, the lines of the other
# configurations, the services end to end.  Outputs under gpurun_out/r03p/; summaries are copied to profiles/.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r03p
rm -rf "$OUT"; mkdir -p "$OUT/bench" "$OUT/cos" "$OUT/gather" "$OUT/clustered"
timeout 200 python -m pytest tests/test_gpu_parity.py -q -x -p no:cacheprovider -k "latency_bound or lone_query" > "$OUT/t_spec.log" 2>&1; echo "rc=$?" >> "$OUT/t_spec.log"
timeout 120 python scripts/bench_single_query.py > "$OUT/r03_single_query_100kx128.json" 2> "$OUT/single.err"
timeout 200 python scripts/profile_spec_hops.py > "$OUT/r03_spec_hop_phases.json" 2> "$OUT/spec_hops.err"
ARGS="--no-cpu --steps 5 --warmup 2"
prof() {  # dir, extra bench args
  local d="$1"; shift
  timeout 300 rocprofv3 --kernel-trace --stats -d "$d/trace" -o trace -- python bench.py $ARGS "$@" > "$d/bench_trace.json" 2> "$d/trace.log"
  timeout 300 rocprofv3 --kernel-include-regex k_search --pmc FETCH_SIZE -d "$d/pmc_fetch" -o pmc -- python bench.py $ARGS "$@" > /dev/null 2> "$d/pmc_fetch.log"
  timeout 300 rocprofv3 --kernel-include-regex k_search --pmc WRITE_SIZE -d "$d/pmc_write" -o pmc -- python bench.py $ARGS "$@" > /dev/null 2> "$d/pmc_write.log"
  timeout 300 rocprofv3 --kernel-include-regex k_search --pmc TCC_HIT_sum TCC_MISS_sum -d "$d/pmc_l2" -o pmc -- python bench.py $ARGS "$@" > /dev/null 2> "$d/pmc_l2.log"
  timeout 300 rocprofv3 --kernel-include-regex k_search --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum -d "$d/pmc_dram" -o pmc -- python bench.py $ARGS "$@" > /dev/null 2> "$d/pmc_dram.log"
}
prof "$OUT/bench"
python scripts/summarize_prof.py "$OUT/bench" "$OUT/r03_bench_1Mx768" > "$OUT/bench_summary.txt" 2>&1
prof "$OUT/cos" --metric cos
python scripts/summarize_prof.py "$OUT/cos" "$OUT/r03_bench_1Mx768_cos" > "$OUT/cos_summary.txt" 2>&1
prof "$OUT/clustered" --data clustered
python scripts/summarize_prof.py "$OUT/clustered" "$OUT/r03_bench_1Mx768_clustered" > "$OUT/clustered_summary.txt" 2>&1
# ---- gather calibration in the walk's launch shape
export LANTERN_GPU_GATHER_WALKSHAPE=1
timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/gather/trace" -o trace -- python scripts/bench_gather_ceiling.py > "$OUT/gather.json" 2> "$OUT/gather_trace.log"
timeout 300 rocprofv3 --kernel-include-regex k_gather --pmc FETCH_SIZE -d "$OUT/gather/pmc_fetch" -o pmc -- python scripts/bench_gather_ceiling.py > /dev/null 2> "$OUT/gather_pmc_fetch.log"
timeout 300 rocprofv3 --kernel-include-regex k_gather --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum -d "$OUT/gather/pmc_dram" -o pmc -- python scripts/bench_gather_ceiling.py > /dev/null 2> "$OUT/gather_pmc_dram.log"
unset LANTERN_GPU_GATHER_WALKSHAPE
python scripts/prof_dump.py "$OUT/gather" k_gather > "$OUT/r03_gather_ceiling.md" 2>&1
# the rocpd databases are tens of MB each and gpurun brings back at most 64 MiB: keep the summaries, drop the raw files
for d in bench cos clustered gather; do
  for l in "$OUT/$d"/*.log; do head -c 600 "$l" > "$l.head" 2>/dev/null; rm -f "$l"; done
  find "$OUT/$d" -type f ! -name "*.head" ! -name "bench_trace.json" -delete; find "$OUT/$d" -type d -empty -delete
done
du -sh "$OUT"
# ---- the lines
python bench.py > "$OUT/r03_bench_line.json" 2> "$OUT/bench_line.err"
python bench.py --no-cpu --streams 2 > "$OUT/r03_bench_line_2streams.json" 2>/dev/null
python bench.py --no-cpu --metric cos > "$OUT/r03_bench_line_cos.json" 2>/dev/null
python bench.py --no-cpu --metric cos --queries 1024 --steps 40 > "$OUT/r03_bench_line_cos_q1024.json" 2>/dev/null
python bench.py --no-cpu --metric cos --queries 1024 --steps 40 --streams 2 > "$OUT/r03_bench_line_cos_q1024_2streams.json" 2>/dev/null
python bench.py --data clustered --cpu-seconds 10 --build-quality-rows 0 > "$OUT/r03_bench_line_clustered.json" 2>/dev/null
python bench.py --no-cpu --data clustered --metric cos > "$OUT/r03_bench_line_clustered_cos.json" 2>/dev/null
python bench.py --no-cpu --quant f16 > "$OUT/r03_bench_line_f16.json" 2>/dev/null
python bench.py --no-cpu --quant i8 --data-scale 0.3 > "$OUT/r03_bench_line_i8.json" 2>/dev/null
python bench.py --no-cpu --quant b1 > "$OUT/r03_bench_line_b1.json" 2>/dev/null
python bench.py --no-cpu --dim 1536 --steps 5 > "$OUT/r03_bench_line_1Mx1536.json" 2>/dev/null
for c in 64 256 512; do timeout 60 lantern_amd/lib/lantern-scan-load --connections $c --seconds 4 >> "$OUT/r03_scan_load_100kx128.jsonl" 2>> "$OUT/scanload.err"; done
timeout 200 lantern_amd/lib/lantern-index-load --rows 1000000 --dim 1536 > "$OUT/r03_index_load_1Mx1536.json" 2> "$OUT/indexload.err"
timeout 200 lantern_amd/lib/lantern-index-load --rows 1000000 --dim 1536 --tuples-per-write 64 > "$OUT/r03_index_load_1Mx1536_64_per_write.json" 2>> "$OUT/indexload.err"
timeout 900 python bench.py --rows 10000000 --ef 128 --steps 5 --no-cpu --truth-queries 256 > "$OUT/r03_bench_line_10Mx768_ef128.json" 2>/dev/null
du -sh "$OUT"; ls "$OUT"; tail -n 3 "$OUT/t_spec.log"
