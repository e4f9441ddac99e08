# cosine search at six workgroups per CU (80 VGPRs) against five (96 VGPRs, with two / three steps of row loads in flight): same box
cd $GRAFT_REPO_ROOT; LIB=lantern_amd/lib/liblantern_gpu.so; cp $LIB /tmp/lib_tree.so
for round in 1 2; do for which in tree c5b2 c5b3; do
  [ $which = tree ] && cp /tmp/lib_tree.so $LIB || cp gpurun_ab_$which.so $LIB
  for wpc in 24 20; do for q in 8192 1024; do
    LANTERN_GPU_WAVES_PER_CU=$wpc python bench.py --data clustered --metric cos --queries $q --no-pmc --no-secondary --no-cpu --no-dram-model --no-gather-ceiling --build-quality-rows 0 2>/dev/null | python -c "
import json,sys; l=json.loads(sys.stdin.read()); print(json.dumps({'lib': '$which', 'waves_per_cu': $wpc, 'queries': $q, 'kernel_ms': round(l['roofline']['avg_launch_ms'],4), 'build': round(l['build_vectors_per_s'])}))"
  done; done
done; done
cp /tmp/lib_tree.so $LIB
