#!/bin/bash
# round 5, fifth GPU call: k_dense_f32's plain-output variant with the deferred stores (parity, then the kernel's time under a trace)
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
