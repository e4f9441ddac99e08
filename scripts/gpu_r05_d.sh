#!/bin/bash
# round 5, fourth GPU call: the scan service without a batching window (answers one by one), 4 and 8 lanes; row-sharded build with the
# shard graphs searched BEFORE the batch joins them
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05d
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_sharded_build.py tests/test_scan_server.py -m gpu -x -q -p no:cacheprovider > gpurun_out/r05d/sharded_and_service.log 2>&1
echo "rc $?" >> gpurun_out/r05d/sharded_and_service.log
timeout 900 python scripts/scan_load_headline.py --connections 1,16,64,256,1024 > gpurun_out/r05d/scan_load_1Mx768_nowindow_4lanes.jsonl 2> gpurun_out/r05d/scan_load_4.err
LANTERN_SCAN_LANES=8 timeout 900 python scripts/scan_load_headline.py --connections 16,64,256,1024 > gpurun_out/r05d/scan_load_1Mx768_nowindow_8lanes.jsonl 2> gpurun_out/r05d/scan_load_8.err
timeout 600 python scripts/rowshard_sweep.py 200000 768 128 l2sq > gpurun_out/r05d/rowshard_sweep.log 2>&1
