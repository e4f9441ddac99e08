#!/bin/bash
# round 5, third GPU call: answers one by one (lane_notify) in the scan service, owned-only grouping of sharded batches,
# the one-wave walk's hop sections, eight ranks on one GPU
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05c
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_scan_server.py tests/test_gpu_sharded_build.py -m gpu -x -q -p no:cacheprovider > gpurun_out/r05c/notify_and_sharded.log 2>&1
echo "rc $?" >> gpurun_out/r05c/notify_and_sharded.log
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -p no:cacheprovider -k "one_wave" > gpurun_out/r05c/solo_parity.log 2>&1
echo "rc $?" >> gpurun_out/r05c/solo_parity.log
timeout 300 python scripts/profile_solo_hops.py > gpurun_out/r05c/one_wave_hop_phases.json 2> gpurun_out/r05c/one_wave_hop_phases.err
timeout 900 python scripts/scan_load_headline.py --connections 1,16,64,256,1024 > gpurun_out/r05c/scan_load_1Mx768_notify.jsonl 2> gpurun_out/r05c/scan_load_notify.err
LANTERN_SCAN_LANES=8 timeout 900 python scripts/scan_load_headline.py --connections 64,256,1024 > gpurun_out/r05c/scan_load_1Mx768_notify_8lanes.jsonl 2> gpurun_out/r05c/scan_load_notify_8lanes.err
timeout 900 python bench.py --gpus 8 --dist-backend files --rows 200000 --no-secondary --build-quality-rows 0 --cpu-seconds 0 > gpurun_out/r05c/bench_8ranks_one_gpu_files.json 2> gpurun_out/r05c/bench_8ranks.err
echo "rc $?" >> gpurun_out/r05c/bench_8ranks.err
timeout 900 python bench.py --gpus 8 --dist-backend rccl --rows 200000 --no-secondary --build-quality-rows 0 --cpu-seconds 0 > gpurun_out/r05c/bench_8ranks_one_gpu_rccl_refused.json 2> gpurun_out/r05c/bench_8ranks_rccl.err
echo "rc $?" >> gpurun_out/r05c/bench_8ranks_rccl.err
