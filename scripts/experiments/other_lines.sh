# the lines beside the headline on one box: config[3] (10M x 768, ef = 128), config[4] (1M x 1536), the storage kinds on the clustered set
cd $GRAFT_REPO_ROOT
bash scripts/gpu.sh bench-10m | tail -3
mkdir -p gpurun_out/c5; timeout 700 python bench.py --dim 1536 --no-secondary --build-quality-rows 0 > gpurun_out/c5/line.json 2> gpurun_out/c5/stderr.log; echo "c5 rc=$?"
bash scripts/experiments/quant_lines.sh
