#!/bin/bash
# round 5, first GPU call: the new parity tests, the bench line with its `secondary` array, the scan service at the headline shape
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
