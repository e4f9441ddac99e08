#!/bin/bash
# batches between one and two walks per CU in the lone-query shape: two workgroups per CU side by side vs one after the other
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r03step12; mkdir -p $OUT; rm -f $OUT/q*.json
timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -p no:cacheprovider -k "shape or spec or lone" > $OUT/t.log 2>&1; echo "rc=$?" >> $OUT/t.log; tail -2 $OUT/t.log
for q in 384 512; do
timeout 300 python bench.py --no-cpu --metric cos --queries $q --steps 40 > $OUT/q${q}_cos_two_per_cu.json 2>> $OUT/err.log
LANTERN_GPU_SPEC2_ONE_PER_CU=1 timeout 300 python bench.py --no-cpu --metric cos --queries $q --steps 40 > $OUT/q${q}_cos_one_per_cu.json 2>> $OUT/err.log
done
timeout 300 python bench.py --no-cpu --queries 512 --steps 40 > $OUT/q512_l2_default.json 2>> $OUT/err.log
LANTERN_GPU_SPEC=0 timeout 300 python bench.py --no-cpu --queries 512 --steps 40 > $OUT/q512_l2_classic.json 2>> $OUT/err.log
for f in $OUT/q*.json; do python - $f <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print(sys.argv[1].split('/')[-1], round(d['value']), round(d['ms_per_step'],3), round(d['roofline']['frac'],3))
PY
done; tail -2 $OUT/err.log
