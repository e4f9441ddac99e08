#!/bin/bash
# build scripts/experiments/dense_variants.hip (here, hipcc cross-compiles) into scripts/experiments/bin/, which travels with gpurun
cd "$(dirname "$0")/../.."
mkdir -p scripts/experiments/bin
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iinclude -Ilantern_amd/csrc "$@" scripts/experiments/dense_variants.hip -o scripts/experiments/bin/dv_main
