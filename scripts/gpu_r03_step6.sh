#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r03s6
rm -rf "$OUT"; mkdir -p "$OUT"
timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -p no:cacheprovider -k "latency_bound or lone_query" > "$OUT/t_spec.log" 2>&1; echo "rc=$?" >> "$OUT/t_spec.log"
timeout 120 python scripts/bench_single_query.py > "$OUT/r03_single_query_100kx128.json" 2> "$OUT/single.err"
timeout 200 python scripts/profile_spec_hops.py > "$OUT/r03_spec_hop_phases.json" 2> "$OUT/spec_hops.err"
tail -n 3 "$OUT/t_spec.log"; cat "$OUT/r03_single_query_100kx128.json"
