#!/bin/bash
# round 5, closing GPU call: the whole -m gpu suite, smoke, the bench line (default command), the kernel trace of the same command,
# eight ranks on the one GPU
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
