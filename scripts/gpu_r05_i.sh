#!/bin/bash
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
