#!/bin/bash
# on the GPU box: the harness built by scripts/experiments/build_dense_variants.sh, one JSON line per (fused) case
mkdir -p gpurun_out/dense_variants
for b in scripts/experiments/bin/dv_*; do timeout 120 $b; done | tee gpurun_out/dense_variants/results.jsonl
