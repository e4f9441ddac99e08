#!/bin/bash
# Round-4 GPU passes (run through gpurun: `gpurun --timeout S -- 'bash scripts/gpu_r04.sh <step>'`).  One file, one function per
# pass; outputs under gpurun_out/r04/, summaries are copied to profiles/ by hand.
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r04; mkdir -p $OUT

probe() {
  { nproc; free -g | head -2; cat /sys/fs/cgroup/memory.max 2>/dev/null; cat /sys/fs/cgroup/cpu.max 2>/dev/null; which rocprofv3; rocm-smi --showclocks 2>/dev/null | head -20; } > $OUT/probe.txt 2>&1
  cat $OUT/probe.txt
}

# C4 / C5 at size: parity tests, then the two bench lines WITH the CPU port beside them
at_size() {
  probe
  timeout 1500 python -m pytest tests/test_gpu_c4_c5_at_size.py "tests/test_gpu_build_parity_production_batch.py::test_device_build_with_8192_row_batches_is_the_oracles_graph_edge_for_edge[c5_gaussian_64k_x_1536_l2sq]" -x -q -s --durations=0 > $OUT/at_size_tests.log 2>&1
  tail -25 $OUT/at_size_tests.log
  timeout 900 python bench.py --rows 10000000 --ef 128 --steps 5 --truth-queries 256 --build-quality-rows 0 > $OUT/r04_bench_line_10Mx768_ef128.json 2> $OUT/bench_10M.err
  tail -c 600 $OUT/r04_bench_line_10Mx768_ef128.json; tail -3 $OUT/bench_10M.err
  timeout 900 python bench.py --dim 1536 --steps 5 --truth-queries 1000 > $OUT/r04_bench_line_1Mx1536.json 2> $OUT/bench_1536.err
  tail -c 600 $OUT/r04_bench_line_1Mx1536.json; tail -3 $OUT/bench_1536.err
}

# counters in the bench's own run; the layout / ABI / mirror tests of this round; the 1536-d cases left over from at_size
traffic() {
  timeout 900 python -m pytest tests/test_c_abi.py tests/test_reference_index_sizes.py tests/test_gpu_scans_and_inserts.py tests/test_gpu_quantized_indexes.py tests/test_scan_server.py \
     "tests/test_gpu_build_parity_production_batch.py::test_device_build_with_8192_row_batches_is_the_oracles_graph_edge_for_edge[c5_gaussian_64k_x_1536_l2sq]" \
     tests/test_gpu_c4_c5_at_size.py::test_c5_batched_build_against_the_sequential_reference_build_at_100k_x_1536 -x -q -s --durations=8 > $OUT/traffic_tests.log 2>&1
  tail -15 $OUT/traffic_tests.log
  ( time python bench.py ) > $OUT/r04_bench_line.json 2> $OUT/bench_line.err
  tail -5 $OUT/bench_line.err
  python - <<'P'
import json
for l in open('gpurun_out/r04/r04_bench_line.json'):
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print(d['value'], d['ms_per_step'], {k:r[k] for k in ('achieved','frac','frac_algorithmic','traffic_measured_in_this_run','traffic','unique_rows_per_launch','frac_cold_miss_lower_bound','traffic_over_algorithmic')})
        print(r['pmc'])
P
  python bench.py --no-cpu --metric cos > $OUT/r04_bench_line_cos.json 2> $OUT/bench_cos.err
  python bench.py --no-cpu --data clustered > $OUT/r04_bench_line_clustered.json 2> $OUT/bench_clustered.err
  python - <<'P'
import json
for f in ('cos','clustered'):
    d=json.load(open(f'gpurun_out/r04/r04_bench_line_{f}.json')); r=d['roofline']
    print(f, d['value'], {k:r[k] for k in ('achieved','frac','frac_algorithmic','traffic_measured_in_this_run','unique_rows_per_launch','frac_cold_miss_lower_bound')})
P
}

"$@"
