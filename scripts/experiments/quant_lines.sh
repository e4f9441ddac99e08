cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/quant
for q in f16 i8 b1; do
  extra=""; [ "$q" = "i8" ] && extra="--data-scale 0.3"
  timeout 500 python bench.py --quant $q $extra --no-secondary --build-quality-rows 0 --data clustered > gpurun_out/quant/line_$q.json 2> gpurun_out/quant/stderr_$q.log; echo "$q rc=$?"
done
timeout 500 python bench.py --data clustered --pq-subvectors 96 --no-cpu --no-secondary --build-quality-rows 0 > gpurun_out/quant/line_pq96.json 2> gpurun_out/quant/stderr_pq96.log; echo "pq96 rc=$?"
timeout 500 python bench.py --data clustered --pq-subvectors 32 --no-cpu --no-secondary --build-quality-rows 0 > gpurun_out/quant/line_pq32.json 2> gpurun_out/quant/stderr_pq32.log; echo "pq32 rc=$?"
