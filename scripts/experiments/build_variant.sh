#!/bin/bash
# build_variant.sh NAME "-DFOO=1 -DBAR=2" -- a library variant for scripts/experiments/ab_lib.sh: the kernel sources recompiled with the
# given defines, linked with the host objects of the library in the tree -> gpurun_ab_NAME.so at the repo root (git-ignored, travels).
set -e
cd "$(dirname "$0")/../.."
NAME="$1"; DEFS="$2"
OBJ=lantern_amd/lib/obj; VAR=/tmp/lgpu_variant_$NAME; mkdir -p "$VAR"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -fvisibility=hidden -Wall -Wno-unused-function -x hip -Ilantern_amd/csrc"
for f in lantern_amd/csrc/*.hip; do
  b=$(basename "$f" .hip)
  /opt/rocm/bin/hipcc $FLAGS $DEFS -c "$f" -o "$VAR/$b.o" &
done
wait
HOST=$(ls $OBJ/*.o | grep -v -E "/(search_kernel|search_spec_kernel|search_adc_kernel|insert_kernel|insert_spec_kernel|kernels|bruteforce|grouping)\.o$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "gpurun_ab_$NAME.so" $VAR/*.o $HOST -lpthread -ldl
ls -la "gpurun_ab_$NAME.so"
