cd $GRAFT_REPO_ROOT
bash scripts/gpu.sh bench-10m | tail -2 | cut -c1-200
mkdir -p gpurun_out/c5; timeout 700 python bench.py --dim 1536 --no-secondary --build-quality-rows 0 > gpurun_out/c5/line.json 2> gpurun_out/c5/stderr.log; echo "c5 rc=$?"
