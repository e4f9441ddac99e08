#!/bin/bash
# streaming serialiser: file round trip + indexing-server tests, then the 1M x 1536 end-to-end load
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r03step8; mkdir -p $OUT
timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_index_server.py tests/test_gpu_quantized_indexes.py -q -x -p no:cacheprovider -k "file_round_trip or server or header or pq" > $OUT/t.log 2>&1; echo "rc=$?" >> $OUT/t.log
LANTERN_INDEX_SERVER_TRACE=1 timeout 200 lantern_amd/lib/lantern-index-load --rows 1000000 --dim 1536 > $OUT/r03_index_load_1Mx1536.json 2> $OUT/indexload.err
tail -3 $OUT/t.log; cat $OUT/r03_index_load_1Mx1536.json; tail -3 $OUT/indexload.err
