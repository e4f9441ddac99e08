#!/bin/bash
# Round-5 GPU passes, one function per gpurun call of the round, in order (run through gpurun: `gpurun --timeout S -- 'bash scripts/gpu_r05.sh <step>'`).
# Outputs under gpurun_out/r05<step>/; the summaries the documents quote are copied to profiles/ by hand (profiles/README.md, Round 5).

# a: the new parity tests (quantised contract, 16384-row plans), the bench line with its `secondary` array
step_a() {
  cd "$GRAFT_REPO_ROOT" || exit 1
  mkdir -p gpurun_out/r05a
  export TMPDIR=/tmp
  nproc > gpurun_out/r05a/nproc.txt
  timeout 900 python -m pytest tests/test_quantized_contract.py -m gpu -x -q -p no:cacheprovider > gpurun_out/r05a/quant_contract.log 2>&1
  echo "quant rc $?" >> gpurun_out/r05a/quant_contract.log
  timeout 1500 python bench.py > gpurun_out/r05a/bench_line.json 2> gpurun_out/r05a/bench_err.log
  echo "bench rc $?" >> gpurun_out/r05a/bench_err.log
  timeout 1200 python -m pytest tests/test_gpu_build_parity_production_batch.py -m gpu -x -q -s -p no:cacheprovider -k "plan16384" > gpurun_out/r05a/build_parity_16384.log 2>&1
  echo "parity rc $?" >> gpurun_out/r05a/build_parity_16384.log
  timeout 900 python -m pytest tests/test_gpu_baseline_configs.py -m gpu -x -q -s -p no:cacheprovider -k "test_build_quality and batch16384" > gpurun_out/r05a/build_quality_16384.log 2>&1
  echo "quality rc $?" >> gpurun_out/r05a/build_quality_16384.log
}

# b: the one-wave walk: parity, lone-query figures against the 3 + 8 wave shape; the serving path at the headline shape
step_b() {
  cd "$GRAFT_REPO_ROOT" || exit 1
  mkdir -p gpurun_out/r05b
  export TMPDIR=/tmp
  timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -p no:cacheprovider -k "one_wave or lone_query_and_small or latency_bound_walk_is_the_oracle_walk" > gpurun_out/r05b/solo_parity.log 2>&1
  echo "rc $?" >> gpurun_out/r05b/solo_parity.log
  timeout 300 python scripts/bench_single_query.py > gpurun_out/r05b/single_query_solo.json 2> gpurun_out/r05b/single_query_solo.err
  LANTERN_GPU_SOLO=0 timeout 300 python scripts/bench_single_query.py --no-cpu > gpurun_out/r05b/single_query_spec2.json 2> gpurun_out/r05b/single_query_spec2.err
  timeout 600 python -m pytest tests/test_quantized_contract.py -m gpu -x -q -p no:cacheprovider > gpurun_out/r05b/quant_contract.log 2>&1
  echo "rc $?" >> gpurun_out/r05b/quant_contract.log
  timeout 900 python scripts/scan_load_headline.py > gpurun_out/r05b/scan_load_1Mx768.jsonl 2> gpurun_out/r05b/scan_load.err
  timeout 900 python -m pytest tests/test_gpu_baseline_configs.py -m gpu -x -q -s -p no:cacheprovider -k "test_build_quality and 400k" > gpurun_out/r05b/build_quality_400k.log 2>&1
  echo "rc $?" >> gpurun_out/r05b/build_quality_400k.log
}

# c: answers one by one (lane_notify) in the scan service, owned-only grouping of sharded batches, the one-wave hop sections, eight ranks on one GPU
step_c() {
  cd "$GRAFT_REPO_ROOT" || exit 1
  mkdir -p gpurun_out/r05c
  export TMPDIR=/tmp
  timeout 900 python -m pytest tests/test_scan_server.py tests/test_gpu_sharded_build.py -m gpu -x -q -p no:cacheprovider > gpurun_out/r05c/notify_and_sharded.log 2>&1
  echo "rc $?" >> gpurun_out/r05c/notify_and_sharded.log
  timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -p no:cacheprovider -k "one_wave" > gpurun_out/r05c/solo_parity.log 2>&1
  echo "rc $?" >> gpurun_out/r05c/solo_parity.log
  timeout 300 python scripts/profile_solo_hops.py > gpurun_out/r05c/one_wave_hop_phases.json 2> gpurun_out/r05c/one_wave_hop_phases.err
  timeout 900 python scripts/scan_load_headline.py --connections 1,16,64,256,1024 > gpurun_out/r05c/scan_load_1Mx768_notify.jsonl 2> gpurun_out/r05c/scan_load_notify.err
  LANTERN_SCAN_LANES=8 timeout 900 python scripts/scan_load_headline.py --connections 64,256,1024 > gpurun_out/r05c/scan_load_1Mx768_notify_8lanes.jsonl 2> gpurun_out/r05c/scan_load_notify_8lanes.err
  timeout 900 python bench.py --gpus 8 --dist-backend files --rows 200000 --no-secondary --build-quality-rows 0 --cpu-seconds 0 > gpurun_out/r05c/bench_8ranks_one_gpu_files.json 2> gpurun_out/r05c/bench_8ranks.err
  echo "rc $?" >> gpurun_out/r05c/bench_8ranks.err
  timeout 900 python bench.py --gpus 8 --dist-backend rccl --rows 200000 --no-secondary --build-quality-rows 0 --cpu-seconds 0 > gpurun_out/r05c/bench_8ranks_one_gpu_rccl_refused.json 2> gpurun_out/r05c/bench_8ranks_rccl.err
  echo "rc $?" >> gpurun_out/r05c/bench_8ranks_rccl.err
}

# d: the scan service without a batching window / with 8 lanes; row-sharded build with the shards searched before the batch joins them
step_d() {
  cd "$GRAFT_REPO_ROOT" || exit 1
  mkdir -p gpurun_out/r05d
  export TMPDIR=/tmp
  timeout 900 python -m pytest tests/test_gpu_sharded_build.py tests/test_scan_server.py -m gpu -x -q -p no:cacheprovider > gpurun_out/r05d/sharded_and_service.log 2>&1
  echo "rc $?" >> gpurun_out/r05d/sharded_and_service.log
  timeout 900 python scripts/scan_load_headline.py --connections 1,16,64,256,1024 > gpurun_out/r05d/scan_load_1Mx768_nowindow_4lanes.jsonl 2> gpurun_out/r05d/scan_load_4.err
  LANTERN_SCAN_LANES=8 timeout 900 python scripts/scan_load_headline.py --connections 16,64,256,1024 > gpurun_out/r05d/scan_load_1Mx768_nowindow_8lanes.jsonl 2> gpurun_out/r05d/scan_load_8.err
  timeout 600 python scripts/rowshard_sweep.py 200000 768 128 l2sq > gpurun_out/r05d/rowshard_sweep.log 2>&1
}

# e: k_dense_f32 plain-output variant with deferred stores: parity, kernel times under a trace
step_e() {
  cd "$GRAFT_REPO_ROOT" || exit 1
  OUT=gpurun_out/r05e; mkdir -p $OUT
  export TMPDIR=/tmp
  timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -p no:cacheprovider -k "mfma or exact_search or distance_matrix or assign_to_clusters" > $OUT/dense_tests.log 2>&1
  echo "rc $?" >> $OUT/dense_tests.log
  timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/plain_trace -o trace -- python scripts/bench_dense_plain.py l2sq > $OUT/plain_l2sq.json 2> $OUT/plain_trace.log
  timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/fused_trace -o trace -- python scripts/bench_dense.py cos > $OUT/fused_cos.json 2> $OUT/fused_trace.log
  python - <<'PY' > gpurun_out/r05e/dense_kernel_times.txt 2>&1
import glob, sqlite3
for which in ("plain_trace", "fused_trace"):
    for f in glob.glob(f"gpurun_out/r05e/{which}/**/*.db", recursive=True):
        cur = sqlite3.connect(f).cursor()
        tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
        print(which, [t for t in tabs if 'kernel' in t.lower()][:8])
        try:
            for r in cur.execute("select name,total_calls,total_duration,average from top_kernels limit 6"): print(which, r)
        except Exception as e: print("top_kernels:", e)
        try:
            kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
            ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
            q = f"select s.kernel_name, d.end - d.start from {kd} d join {ks} s on d.kernel_id = s.id where s.kernel_name like '%k_dense_f32%' order by d.start"
            rows = list(cur.execute(q))
            print(which, "k_dense_f32 launches (us):", [round(x[1] / 1000.0, 1) for x in rows][:40])
        except Exception as e: print("dispatch query:", e)
PY
  cat gpurun_out/r05e/dense_kernel_times.txt | cut -c1-600
}

# f: the whole -m gpu suite, the bench line, the kernel trace of the bench command, the service per-leg clock
step_f() {
  cd "$GRAFT_REPO_ROOT" || exit 1
  OUT=gpurun_out/r05f; mkdir -p $OUT
  export TMPDIR=/tmp
  ( time timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider --durations=15 ) > $OUT/gpu_suite.log 2>&1
  echo "rc $?" >> $OUT/gpu_suite.log
  ( time timeout 900 python bench.py ) > $OUT/bench_line.json 2> $OUT/bench.err
  echo "rc $?" >> $OUT/bench.err
  rm -rf gpurun_out/trace_only
  bash scripts/trace_only.sh --no-secondary --build-quality-rows 0 --no-pmc > $OUT/trace_summary.txt 2>&1
  cp -r gpurun_out/trace_only/summary $OUT/trace_summary 2>/dev/null
  cp gpurun_out/trace_only/bench_trace.json $OUT/bench_trace.json 2>/dev/null
  timeout 600 python scripts/scan_load_headline.py --connections 16,64,256,1024 > $OUT/scan_load_timing.jsonl 2> $OUT/scan_load.err
}

# g: service with queries read from the page-locked block; bench with the service leg only
step_g() {
  cd "$GRAFT_REPO_ROOT" || exit 1
  OUT=gpurun_out/r05g; mkdir -p $OUT
  export TMPDIR=/tmp
  timeout 600 python -m pytest tests/test_scan_server.py -m gpu -x -q -p no:cacheprovider > $OUT/service_tests.log 2>&1; echo "rc $?" >> $OUT/service_tests.log
  timeout 600 python scripts/scan_load_headline.py --connections 1,16,64,256,1024 > $OUT/scan_load.jsonl 2> $OUT/scan_load.err
  LANTERN_BENCH_SECONDARY=headline_scan_service timeout 600 python bench.py --no-pmc --no-cpu --build-quality-rows 0 > $OUT/bench_service_only.json 2> $OUT/bench_service_only.err
  LANTERN_BENCH_SECONDARY=headline_scan_service timeout 600 python bench.py --no-pmc --no-cpu --build-quality-rows 0 --add-batch 8192 > $OUT/bench_service_only_plan8192.json 2> $OUT/bench_service_only2.err
}

# h: server-side legs inside bench.py against the measuring script
step_h() {
  cd "$GRAFT_REPO_ROOT" || exit 1
  OUT=gpurun_out/r05h; mkdir -p $OUT
  export TMPDIR=/tmp
  LANTERN_BENCH_SECONDARY=headline_scan_service timeout 600 python bench.py --no-pmc --no-cpu --build-quality-rows 0 --steps 3 > $OUT/bench_service_only.json 2> $OUT/e1.err
  timeout 600 python scripts/scan_load_headline.py --connections 256 > $OUT/scan_load_256.jsonl 2> $OUT/e2.err
}

# i: kernel trace of the service inside bench.py and inside the script (identical kernel times: the hardware-queue clue)
step_i() {
  cd "$GRAFT_REPO_ROOT" || exit 1
  OUT=gpurun_out/r05i; mkdir -p $OUT
  export TMPDIR=/tmp
  LANTERN_BENCH_SECONDARY=headline_scan_service timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/t_bench -o trace -- python bench.py --no-pmc --no-cpu --build-quality-rows 0 --steps 3 > $OUT/bench_service_only.json 2> $OUT/e1.err
  timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/t_script -o trace -- python scripts/scan_load_headline.py --connections 256 > $OUT/scan_load_256.jsonl 2> $OUT/e2.err
  python - <<'PY' > gpurun_out/r05i/kernels.txt 2>&1
import glob, sqlite3
for which in ("t_bench", "t_script"):
    for f in glob.glob(f"gpurun_out/r05i/{which}/**/*.db", recursive=True):
        cur = sqlite3.connect(f).cursor()
        for r in cur.execute("select name,total_calls,total_duration,average from top_kernels where name like '%k_search%' limit 8"): print(which, r)
        for r in cur.execute("select name, count(*), avg(end-start), min(end-start), max(end-start), avg(grid_x), avg(workgroup_x), avg(lds_size) from kernels where name like '%k_search%' group by name"): print(which, 'K', r)
PY
  rm -rf $OUT/t_bench $OUT/t_script
  cat gpurun_out/r05i/kernels.txt | cut -c1-400
}

# j: GPU_MAX_HW_QUEUES=16: service at 4 and 8 lanes, bench service leg
step_j() {
  cd "$GRAFT_REPO_ROOT" || exit 1
  OUT=gpurun_out/r05j; mkdir -p $OUT
  export TMPDIR=/tmp
  GPU_MAX_HW_QUEUES=16 timeout 600 python scripts/scan_load_headline.py --connections 64,256,1024 > $OUT/scan_load_q16.jsonl 2> $OUT/e1.err
  GPU_MAX_HW_QUEUES=16 LANTERN_SCAN_LANES=8 timeout 600 python scripts/scan_load_headline.py --connections 64,256,1024 > $OUT/scan_load_q16_8lanes.jsonl 2> $OUT/e2.err
  GPU_MAX_HW_QUEUES=16 LANTERN_BENCH_SECONDARY=headline_scan_service timeout 600 python bench.py --no-pmc --no-cpu --build-quality-rows 0 --steps 3 > $OUT/bench_service_only_q16.json 2> $OUT/e3.err
}

# k: closing call: the whole -m gpu suite, smoke, the bench line, its kernel trace, eight ranks on the one GPU
step_k() {
  cd "$GRAFT_REPO_ROOT" || exit 1
  OUT=gpurun_out/r05k; mkdir -p $OUT
  export TMPDIR=/tmp
  ( time timeout 1800 python -m pytest tests -m gpu -x -q -p no:cacheprovider --durations=8 ) > $OUT/gpu_suite.log 2>&1
  echo "rc $?" >> $OUT/gpu_suite.log
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "rc $?" >> $OUT/smoke.log
  ( time timeout 900 python bench.py ) > $OUT/bench_line.json 2> $OUT/bench.err
  echo "rc $?" >> $OUT/bench.err
  rm -rf gpurun_out/trace_only
  bash scripts/trace_only.sh --no-secondary --build-quality-rows 0 --no-pmc > $OUT/trace_summary.txt 2>&1
  cp -r gpurun_out/trace_only/summary $OUT/trace_summary 2>/dev/null
  timeout 900 python bench.py --gpus 8 --dist-backend files --rows 200000 --no-secondary --build-quality-rows 0 --cpu-seconds 0 > $OUT/bench_8ranks_one_gpu_files.json 2> $OUT/bench_8ranks.err
  echo "rc $?" >> $OUT/bench_8ranks.err
}

# l: the records round 4 did not re-measure: B3 end to end at 1M x 1536, one insertion per call, BASELINE configs [3] and [4] on one GPU
step_l() {
  cd "$GRAFT_REPO_ROOT" || exit 1
  OUT=gpurun_out/r05l; mkdir -p $OUT
  export TMPDIR=/tmp
  timeout 300 lantern_amd/lib/lantern-index-load --rows 1000000 --dim 1536 > $OUT/index_load_1Mx1536.json 2> $OUT/indexload.err
  timeout 200 python scripts/bench_single_insert.py > $OUT/single_insert_100kx128.json 2> $OUT/single_insert.err
  timeout 900 python bench.py --rows 10000000 --ef 128 --steps 5 --truth-queries 256 --build-quality-rows 0 --no-secondary > $OUT/bench_line_10Mx768_ef128.json 2> $OUT/bench_10M.err
  timeout 900 python bench.py --dim 1536 --steps 5 --truth-queries 1000 --no-secondary > $OUT/bench_line_1Mx1536.json 2> $OUT/bench_1536.err
}

# m: host batches padded on a few threads: lantern_gpu_search_batch[_lane] at the headline shape, the tests that go through them
step_m() {
  cd "$GRAFT_REPO_ROOT" || exit 1
  OUT=gpurun_out/r05p; mkdir -p $OUT
  export TMPDIR=/tmp
  timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_scan_server.py tests/test_gpu_partitioned_search.py -m gpu -x -q -p no:cacheprovider > $OUT/tests.log 2>&1; echo "rc $?" >> $OUT/tests.log
  timeout 600 python scripts/scan_load_headline.py --connections 256 > $OUT/scan_load.jsonl 2> $OUT/scan_load.err
  LANTERN_BENCH_SECONDARY=headline_host_buffers timeout 600 python bench.py --no-pmc --no-cpu --build-quality-rows 0 --steps 5 > $OUT/bench_host_buffers.json 2> $OUT/bench.err
}

"step_${1:?usage: gpu_r05.sh <a..m>}"
