#!/bin/bash
# round 5, second GPU call: the one-wave walk (parity, then the lone-query figures against the 3 + 8 wave shape), the fixed tests,
# the serving path at the headline shape
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05b
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -p no:cacheprovider -k "one_wave or lone_query_and_small or latency_bound_walk_is_the_oracle_walk" > gpurun_out/r05b/solo_parity.log 2>&1
echo "rc $?" >> gpurun_out/r05b/solo_parity.log
timeout 300 python scripts/bench_single_query.py > gpurun_out/r05b/single_query_solo.json 2> gpurun_out/r05b/single_query_solo.err
LANTERN_GPU_SOLO=0 timeout 300 python scripts/bench_single_query.py --no-cpu > gpurun_out/r05b/single_query_spec2.json 2> gpurun_out/r05b/single_query_spec2.err
timeout 600 python -m pytest tests/test_quantized_contract.py -m gpu -x -q -p no:cacheprovider > gpurun_out/r05b/quant_contract.log 2>&1
echo "rc $?" >> gpurun_out/r05b/quant_contract.log
timeout 900 python scripts/scan_load_headline.py > gpurun_out/r05b/scan_load_1Mx768.jsonl 2> gpurun_out/r05b/scan_load.err
timeout 900 python -m pytest tests/test_gpu_baseline_configs.py -m gpu -x -q -s -p no:cacheprovider -k "test_build_quality and 400k" > gpurun_out/r05b/build_quality_400k.log 2>&1
echo "rc $?" >> gpurun_out/r05b/build_quality_400k.log
