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

# the rest of the test list of `traffic` (it stops at the first failure), then the scan service with 1 / 2 / 3 / 4 lanes
lanes() {
  timeout 900 python -m pytest tests/test_c_abi.py tests/test_reference_index_sizes.py tests/test_gpu_scans_and_inserts.py tests/test_gpu_quantized_indexes.py tests/test_scan_server.py \
     "tests/test_gpu_build_parity_production_batch.py::test_device_build_with_8192_row_batches_is_the_oracles_graph_edge_for_edge[c5_gaussian_64k_x_1536_l2sq]" \
     tests/test_gpu_c4_c5_at_size.py::test_c5_batched_build_against_the_sequential_reference_build_at_100k_x_1536 -q -s --durations=8 > $OUT/lanes_tests.log 2>&1
  tail -15 $OUT/lanes_tests.log
  rm -f $OUT/r04_scan_load_lanes.jsonl
  for l in 1 2 3 4; do for c in 1 8 16 64 256; do
    echo "{\"lanes\": $l}" >> $OUT/r04_scan_load_lanes.jsonl
    LANTERN_SCAN_LANES=$l timeout 60 lantern_amd/lib/lantern-scan-load --connections $c --seconds 3 >> $OUT/r04_scan_load_lanes.jsonl 2>> $OUT/scanload.err
  done; done
  python - <<'P'
import json
lanes=None
for l in open('gpurun_out/r04/r04_scan_load_lanes.jsonl'):
    d=json.loads(l)
    if 'lanes' in d and len(d)==1: lanes=d['lanes']; continue
    print(lanes, {k:d.get(k) for k in ('connections','qps','latency_us_p50','latency_us_p99','mean_batch')} if 'qps' in d else list(d.items())[:8])
P
}

# the two-nodes-per-round walk: parity in every regime, then the lone query and the compact pq index with and without it
twin() {
  timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "latency_bound or lone_query" > $OUT/twin_tests.log 2>&1; tail -6 $OUT/twin_tests.log
  timeout 300 python -m pytest tests/test_gpu_quantized_indexes.py tests/test_c_abi.py "tests/test_gpu_scans_and_inserts.py" -q > $OUT/twin_tests2.log 2>&1; tail -4 $OUT/twin_tests2.log
  for t in 0 1; do
    LANTERN_GPU_TWIN=$t timeout 200 python scripts/bench_single_query.py > $OUT/single_twin$t.json 2> $OUT/single_twin$t.err
    python -c "
import json; d=json.load(open('$OUT/single_twin$t.json')); print('twin=$t', d['us_per_query_wall'], d['kernel_only']['latency_bound_shape'], d['identical_to_batch_search'], d.get('cpu_port_us_per_query_1_thread'))"
  done
  for t in 0 1; do
    LANTERN_GPU_TWIN=$t timeout 300 python bench.py --no-cpu --no-pmc --data clustered --pq-subvectors 96 --steps 5 > $OUT/pq96_twin$t.json 2> $OUT/pq96_twin$t.err
    python -c "
import json; d=json.load(open('$OUT/pq96_twin$t.json')); print('pq96 twin=$t', d['value'], d['recall_at_10'], d['ms_per_step'])"
  done
}

twinprof() {
  timeout 300 python scripts/profile_twin.py > $OUT/twin_profile.json 2> $OUT/twin_profile.err; cat $OUT/twin_profile.json; tail -3 $OUT/twin_profile.err
}

smallidx() {
  timeout 300 python scripts/profile_small_index.py > $OUT/small_index_profile.json 2> $OUT/small_index.err; python - <<'P'
import json
d=json.load(open('gpurun_out/r04/small_index_profile.json'))
for k,v in d.items():
    print(k, 'us/launch', round(v['us_per_launch'],1), 'hops', round(v['hops_per_query'],1), 'us/hop', round(v['us_per_hop'],3))
    for role,sec in v['cycles_per_hop'].items(): print('   ', role, {a:round(b) for a,b in sec.items()})
P
  tail -3 $OUT/small_index.err
}

# 3 + 4 waves (two per SIMD) against 3 + 8 for the lone query and the compact pq index
waves7() {
  timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -k "latency_bound or lone_query" > $OUT/w7_tests.log 2>&1; tail -3 $OUT/w7_tests.log
  for w in 11 7; do
    LANTERN_GPU_SPEC_WAVES=$w timeout 200 python scripts/bench_single_query.py --no-cpu > $OUT/single_w$w.json 2> $OUT/single_w$w.err
    python -c "
import json; d=json.load(open('$OUT/single_w$w.json')); print('waves=$w', d['us_per_query_wall'], d['kernel_only']['latency_bound_shape'], d['identical_to_batch_search'])"
  done
  for w in 11 7; do
    LANTERN_GPU_SPEC_WAVES=$w timeout 300 python bench.py --no-cpu --no-pmc --data clustered --pq-subvectors 96 --steps 5 > $OUT/pq96_w$w.json 2> $OUT/pq96_w$w.err
    python -c "
import json; d=json.load(open('$OUT/pq96_w$w.json')); print('pq96 waves=$w', d['value'], d['recall_at_10'], d['ms_per_step'])"
  done
}

# compact pq indexes: rows decoded on the fly (the default) against the table walk (LANTERN_GPU_PQ_ADC=1)
pqd() {
  timeout 600 python -m pytest tests/test_gpu_quantized_indexes.py -q -x > $OUT/pqd_tests.log 2>&1; tail -5 $OUT/pqd_tests.log
  for sv in 96 32; do for adc in 0 1; do
    LANTERN_GPU_PQ_ADC=$adc timeout 300 python bench.py --no-cpu --no-pmc --data clustered --pq-subvectors $sv --steps 5 > $OUT/pq${sv}_adc$adc.json 2> $OUT/pq${sv}_adc$adc.err
    python -c "
import json; d=json.load(open('$OUT/pq${sv}_adc$adc.json')); print('pq$sv adc=$adc', round(d['value']), d['recall_at_10'], round(d['ms_per_step'],3), d['roofline']['kernel'])" || tail -3 $OUT/pq${sv}_adc$adc.err
  done; done
  LANTERN_GPU_PQ_ADC=0 timeout 300 python bench.py --no-cpu --no-pmc --pq-subvectors 96 --steps 5 > $OUT/pq96_gaussian_pqd.json 2> $OUT/pq96_gaussian_pqd.err
  python -c "
import json; d=json.load(open('$OUT/pq96_gaussian_pqd.json')); print('pq96 gaussian pqd', round(d['value']), d['recall_at_10'])"
}

# k_dense_f32: one script, trace-only and counter runs back to back, the shader clock sampled alongside (rocm-smi every 100 ms)
dense() {
  D=$OUT/dense; rm -rf $D; mkdir -p $D
  sample() { while true; do rocm-smi --showclocks 2>/dev/null | grep -E "sclk|mclk" | tr -s ' \t' ' ' | tr '\n' ';'; echo; sleep 0.1; done; }
  for mode in plain trace pmc trace2; do
    sample > $D/clocks_$mode.txt & SP=$!
    case $mode in
      plain) timeout 200 python scripts/bench_dense.py > $D/bench_$mode.json 2> $D/$mode.log ;;
      trace|trace2) timeout 200 rocprofv3 --kernel-trace --stats -d $D/$mode -o t -- python scripts/bench_dense.py > $D/bench_$mode.json 2> $D/$mode.log ;;
      pmc) timeout 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 -d $D/$mode -o t -- python scripts/bench_dense.py > $D/bench_$mode.json 2> $D/$mode.log ;;
    esac
    kill $SP 2>/dev/null; wait $SP 2>/dev/null
  done
  python - <<'P'
import glob, sqlite3, json, re, collections
D='gpurun_out/r04/dense'
for mode in ('plain','trace','pmc','trace2'):
    line=[l for l in open(f'{D}/bench_{mode}.json') if l.startswith('{')]
    wall=json.loads(line[-1])['seconds_per_call_wall']*1e3 if line else None
    clk=collections.Counter(re.findall(r'sclk clock level: \S+ \((\d+)Mhz\)', open(f'{D}/clocks_{mode}.txt').read()))
    print(mode, 'exact k-NN wall ms', wall, 'sclk samples (MHz: count)', dict(clk.most_common(6)))
    for f in glob.glob(f'{D}/{mode}/**/*.db', recursive=True):
        cur=sqlite3.connect(f).cursor()
        try:
            for r in cur.execute("select name,total_calls,average from top_kernels where name like '%k_dense_f32%'"): print('   ', r[0][:40], r[1], round(r[2],1),'us')
        except Exception as e: pass
        try:
            rows=list(cur.execute("select (end-start) from kernels where name like '%k_dense_f32<1, true>%' order by start"))
            if rows:
                d=[r[0]/1e3 for r in rows]; print('    per-launch us: first 8', [round(x) for x in d[:8]], 'last 4', [round(x) for x in d[-4:]], 'median', round(sorted(d)[len(d)//2]))
        except Exception as e: print('   ', e)
        try:
            for r in cur.execute("select counter_name,count(*),avg(value) from counters_collection where kernel_name like '%k_dense_f32<1, true>%' group by counter_name"): print('    ', r)
        except Exception: pass
P
  rm -rf $D/trace $D/pmc $D/trace2
}

# the whole -m gpu suite + smoke, as the driver runs them at round end
suite() {
  ( time timeout 1500 python -m pytest tests/ -q -m gpu --durations=12 ) > $OUT/suite.log 2>&1; tail -25 $OUT/suite.log
  python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
}

# round-end records: build-parity tests (the small-batch grouping kernel is on their path), the kernel trace of the bench command, the
# secondary lines, one insertion / one query per call
final() {
  timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scans_and_inserts.py -q -x > $OUT/final_tests.log 2>&1; tail -3 $OUT/final_tests.log
  mkdir -p $OUT/ktrace
  timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/ktrace/trace -o trace -- python bench.py --no-cpu --no-pmc --steps 5 --warmup 2 > $OUT/ktrace/bench_trace.json 2> $OUT/ktrace/trace.log
  python scripts/summarize_prof.py $OUT/ktrace $OUT/r04_bench_1Mx768 > $OUT/ktrace_summary.txt 2>&1; head -12 $OUT/ktrace_summary.txt
  find $OUT/ktrace -name "*.db" -delete
  timeout 200 python scripts/bench_single_insert.py > $OUT/r04_single_insert_100kx128.json 2> $OUT/single_insert.err; python -c "
import json; d=json.load(open('$OUT/r04_single_insert_100kx128.json')); print({k:v for k,v in d.items() if 'us' in k or 'identical' in k})"
  timeout 200 python scripts/bench_single_query.py > $OUT/r04_single_query_100kx128.json 2> $OUT/single_query.err; python -c "
import json; d=json.load(open('$OUT/r04_single_query_100kx128.json')); print(d['us_per_query_wall'], d['kernel_only'], d.get('cpu_port_us_per_query_1_thread'))"
  python bench.py --no-cpu --metric cos --queries 1024 --steps 40 > $OUT/r04_bench_line_cos_q1024.json 2>/dev/null
  python bench.py --no-cpu --no-pmc --streams 2 > $OUT/r04_bench_line_2streams.json 2>/dev/null
  python bench.py --no-cpu --quant f16 > $OUT/r04_bench_line_f16.json 2>/dev/null
  python bench.py --no-cpu --quant i8 --data-scale 0.3 > $OUT/r04_bench_line_i8.json 2>/dev/null
  python bench.py --no-cpu --quant b1 > $OUT/r04_bench_line_b1.json 2>/dev/null
  python - <<'P'
import json
for f in ('cos_q1024','2streams','f16','i8','b1'):
    try:
        d=json.load(open(f'gpurun_out/r04/r04_bench_line_{f}.json')); r=d['roofline']
        print(f, round(d['value']), round(d['ms_per_step'],3), 'frac', round(r['frac'],3), 'alg', round(r['frac_algorithmic'],3), 'in-run', r['traffic_measured_in_this_run'], 'build', round(d['build_vectors_per_s']))
    except Exception as e: print(f, e)
P
}

# more lanes: 4 / 6 / 8 dispatchers at 16 / 64 / 256 connections
lanes8() {
  timeout 200 python -m pytest tests/test_gpu_scans_and_inserts.py tests/test_scan_server.py tests/test_gpu_quantized_indexes.py -q -x > $OUT/lanes8_tests.log 2>&1; tail -3 $OUT/lanes8_tests.log
  rm -f $OUT/r04_scan_load_lanes8.jsonl
  for l in 4 6 8; do for c in 16 64 256; do
    echo "{\"lanes\": $l}" >> $OUT/r04_scan_load_lanes8.jsonl
    LANTERN_SCAN_LANES=$l timeout 60 lantern_amd/lib/lantern-scan-load --connections $c --seconds 3 >> $OUT/r04_scan_load_lanes8.jsonl 2>> $OUT/scanload8.err
  done; done
  python - <<'P'
import json
lanes=None
for l in open('gpurun_out/r04/r04_scan_load_lanes8.jsonl'):
    d=json.loads(l)
    if len(d)==1: lanes=d['lanes']; continue
    print(lanes, d['connections'], round(d['queries_per_s']), d['latency_us']['p50'], d['latency_us']['p99'], d['service']['mean_batch'])
P
  timeout 600 python bench.py --gpus 2 --dist-backend files --no-cpu --rows 300000 --steps 10 > $OUT/bench_2ranks_one_gpu.json 2> $OUT/bench_2ranks.err; tail -c 700 $OUT/bench_2ranks_one_gpu.json; tail -3 $OUT/bench_2ranks.err
}

# the decode-on-the-fly walk by the counters: fabric bytes (the bench's own passes) and the L2 hit rate
pqdpmc() {
  timeout 400 python bench.py --no-cpu --data clustered --pq-subvectors 96 --steps 5 > $OUT/r04_bench_line_pq96_compact_clustered.json 2> $OUT/pq96_pmc.err
  timeout 400 python bench.py --no-cpu --data clustered --pq-subvectors 32 --steps 5 > $OUT/r04_bench_line_pq32_compact_clustered.json 2> $OUT/pq32_pmc.err
  mkdir -p $OUT/pqd_l2
  timeout 300 rocprofv3 --kernel-include-regex k_search --pmc TCC_HIT_sum TCC_MISS_sum -d $OUT/pqd_l2/p -o pmc -- python bench.py --no-cpu --no-pmc --data clustered --pq-subvectors 96 --steps 4 --warmup 1 > /dev/null 2> $OUT/pqd_l2.log
  timeout 300 rocprofv3 --kernel-include-regex k_search --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum -d $OUT/pqd_l2/q -o pmc -- python bench.py --no-cpu --no-pmc --data clustered --pq-subvectors 96 --steps 4 --warmup 1 > /dev/null 2> $OUT/pqd_l1.log
  python - <<'P'
import glob, sqlite3, json
for f in ('96','32'):
    d=json.load(open(f'gpurun_out/r04/r04_bench_line_pq{f}_compact_clustered.json')); r=d['roofline']
    print('pq'+f, round(d['value']), 'launch ms', round(r['avg_launch_ms'],3), 'traffic GB', (r['traffic'] or 0)/1e9, 'alg GB', r['algorithmic_bytes_per_launch']/1e9, 'frac', r['frac'], 'in-run', r['traffic_measured_in_this_run'])
for db in glob.glob('gpurun_out/r04/pqd_l2/**/*.db', recursive=True):
    cur=sqlite3.connect(db).cursor()
    for row in cur.execute("select counter_name,count(*),avg(value) from counters_collection where kernel_name like '%k_search%' group by counter_name"): print(db.split('/')[-3], row)
P
  rm -rf $OUT/pqd_l2
}

rowshard() {
  timeout 900 python -m pytest tests/test_gpu_sharded_build.py -q -s -k "row_sharded" > $OUT/rowshard.log 2>&1
  grep -a "recall@10\|passed\|failed\|Error" $OUT/rowshard.log | tail -40
}


rowsweep() {
  timeout 900 python scripts/rowshard_sweep.py 200000 768 128 l2sq > $OUT/rowshard_sweep_200k_768.log 2>&1; tail -9 $OUT/rowshard_sweep_200k_768.log
  timeout 900 python scripts/rowshard_sweep.py 100000 1536 128 cos > $OUT/rowshard_sweep_100k_1536.log 2>&1; tail -9 $OUT/rowshard_sweep_100k_1536.log
}

rowseeds() {
  timeout 600 python scripts/rowshard_seeds.py 20000 768 64 l2sq 2 2>&1 | tail -7
  timeout 600 python scripts/rowshard_seeds.py 20000 768 128 l2sq 3 2>&1 | tail -7
}

rowsweep2() {
  timeout 900 python scripts/rowshard_sweep.py 200000 768 128 l2sq 4:0:0,4:64:64,4:48:48,4:33:33,8:0:0,8:33:33 > $OUT/rowshard_sweep_ef_200k_768.log 2>&1; tail -9 $OUT/rowshard_sweep_ef_200k_768.log
}

instrace() {
  rm -rf $OUT/instrace; mkdir -p $OUT/instrace
  (cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/instrace -o t -- python $GRAFT_REPO_ROOT/scripts/bench_single_insert.py --no-cpu --inserts 200 > $GRAFT_REPO_ROOT/$OUT/instrace/line.json 2> $GRAFT_REPO_ROOT/$OUT/instrace/err.log)
  python - <<'P'
import csv, glob, collections
ks = []
for f in glob.glob('gpurun_out/r04/instrace/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        ks.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][:60]))
for f in glob.glob('gpurun_out/r04/instrace/**/*memory_copy_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        ks.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'COPY ' + r.get('Direction', r.get('Name', ''))[:40]))
ks.sort()
# the last 100 inserts: a cycle starts at each k_insert_spec
idx = [i for i, k in enumerate(ks) if 'k_insert_spec' in k[2]]
idx = idx[-101:]
dur = collections.defaultdict(list); gap = collections.defaultdict(list); cyc = []
for a, b in zip(idx[:-1], idx[1:]):
    # the cycle's ops: from the first op after the previous cycle's last kernel ... simpler: ops in [a - pre, b - pre) where pre = copies before the walk
    ops = ks[a:b]
    cyc.append(ops[-1][1] - ops[0][0])
    prev_end = None
    for s0, e0, n in ops:
        dur[n].append(e0 - s0)
        if prev_end is not None: gap[n].append(s0 - prev_end)
        prev_end = e0
print('ops per insert (from the walk kernel to the next walk kernel), mean over', len(cyc), 'inserts; ns')
tot = 0
for n in dur:
    d = sum(dur[n]) / len(cyc); g = sum(gap[n]) / len(cyc) if gap[n] else 0
    print(f'{n:62s} x{len(dur[n]) / len(cyc):.1f}  busy {d:9.0f}  gap before {g:9.0f}')
    tot += d
print('busy total', tot, 'walk-to-walk period', sum(b0 - a0 for a0, b0 in zip([ks[i][0] for i in idx[:-1]], [ks[i][0] for i in idx[1:]])) / len(cyc))
P
  cat $OUT/instrace/line.json | head -c 600
  rm -rf $OUT/instrace
}

# bench.py --gpus 2 on the one-GPU box (two ranks share the device, host transport): both collective builds, like for like
tworanks() {
  timeout 600 python bench.py --gpus 2 --dist-backend files --no-cpu --no-pmc --rows 500000 --steps 10 > $OUT/r04_bench_line_2ranks_one_gpu_work_sharded.json 2> $OUT/tworanks_work.err
  timeout 600 python bench.py --gpus 2 --dist-backend files --no-cpu --no-pmc --rows 500000 --steps 10 --build row-sharded > $OUT/r04_bench_line_2ranks_one_gpu_row_sharded.json 2> $OUT/tworanks_rows.err
  python - <<'P'
import json
for k in ('work', 'row'):
    try:
        d = json.load(open(f'gpurun_out/r04/r04_bench_line_2ranks_one_gpu_{k}_sharded.json'))
        c = d['config']; print(k, 'value', round(d['value']), 'recall', d.get('recall_at_k'), 'collective', json.dumps(c.get('collective_build'))[:400])
    except Exception as e:
        print(k, 'failed', e)
P
  tail -3 $OUT/tworanks_work.err $OUT/tworanks_rows.err
}

# the round's closing records: the benchmark line (CPU port beside it, counters in the run), the kernel trace of the bench command
final2() {
  ( time python bench.py ) > $OUT/r04_bench_line.json 2> $OUT/bench_line.err; tail -4 $OUT/bench_line.err
  python - <<'P'
import json
for l in open('gpurun_out/r04/r04_bench_line.json'):
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print(d['value'], d['ms_per_step'], 'build', d['build_vectors_per_s'], {k:r[k] for k in ('achieved','frac','frac_algorithmic','traffic_measured_in_this_run','traffic','unique_rows_per_launch')}, d['cpu_baseline'])
P
  mkdir -p $OUT/ktrace
  timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/ktrace/trace -o trace -- python bench.py --no-cpu --no-pmc --steps 5 --warmup 2 > $OUT/ktrace/bench_trace.json 2> $OUT/ktrace/trace.log
  python scripts/summarize_prof.py $OUT/ktrace $OUT/r04_bench_1Mx768 > $OUT/ktrace_summary.txt 2>&1; head -12 $OUT/ktrace_summary.txt
  find $OUT/ktrace -name "*.db" -delete
  timeout 200 python scripts/bench_single_query.py > $OUT/r04_single_query_100kx128.json 2> $OUT/single_query.err; python -c "
import json; d=json.load(open('$OUT/r04_single_query_100kx128.json')); print(d['us_per_query_wall'], d['kernel_only'], d.get('cpu_port_us_per_query_1_thread'))"
}

# BASELINE configs [3] and [4] again, in the line layout with the counters measured in the run
at_size2() {
  timeout 900 python bench.py --dim 1536 --steps 5 --truth-queries 1000 > $OUT/r04_bench_line_1Mx1536.json 2> $OUT/bench_1536.err
  tail -2 $OUT/bench_1536.err
  timeout 1500 python bench.py --rows 10000000 --ef 128 --steps 5 --truth-queries 256 --build-quality-rows 0 --pmc-steps 2 > $OUT/r04_bench_line_10Mx768_ef128.json 2> $OUT/bench_10M.err
  tail -2 $OUT/bench_10M.err
  python - <<'P'
import json
for f in ('1Mx1536','10Mx768_ef128'):
    try:
        d=json.load(open(f'gpurun_out/r04/r04_bench_line_{f}.json')); r=d['roofline']
        print(f, round(d['value']), d['ms_per_step'], 'recall', d['recall_at_10'], 'frac', r['frac'], 'alg', r['frac_algorithmic'], 'in-run', r['traffic_measured_in_this_run'], 'build', round(d['build_vectors_per_s']), 'cpu', d['cpu_baseline'].get('value'))
    except Exception as e: print(f, 'failed', e)
P
}

"$@"
