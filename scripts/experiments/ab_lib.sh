#!/bin/bash
# ab_lib.sh "OTHER1.so OTHER2.so ..." [bench flags] -- the search / build legs of bench.py with the library in the tree ("tree") and with
# each OTHER.so swapped in, alternating twice on the same box (box-to-box differences are 1 - 2 %, larger than most kernel changes):
# one JSON summary line per run.  LEGS="lone gaussian clustered cos" selects the legs (default: all).
# (scripts/experiments/build_variant.sh makes the OTHER libraries.)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
OTHERS="$1"; shift
LEGS="${LEGS:-lone gaussian clustered cos}"
OUT=gpurun_out/ab_lib; mkdir -p "$OUT"
LIB=lantern_amd/lib/liblantern_gpu.so
cp "$LIB" /tmp/lib_tree.so
NAMES="tree"
for o in $OTHERS; do n=$(basename "$o" .so); n=${n#gpurun_ab_}; cp "$o" /tmp/lib_$n.so; NAMES="$NAMES $n"; done
for round in 1 2; do
  for which in $NAMES; do
    cp /tmp/lib_$which.so "$LIB"
    for leg in $LEGS; do
      case $leg in
        lone) python scripts/experiments/lone_query.py 2> "$OUT/${which}_lone_$round.err" | sed "s/^/{\"lib\": \"$which\", \"round\": $round, \"lone\": /; s/$/}/"; continue ;;
        gaussian) flags="--data gaussian" ;;
        clustered) flags="--data clustered" ;;
        cos) flags="--data clustered --metric cos" ;;
        cos1024) flags="--data clustered --metric cos --queries 1024" ;;
        f16) flags="--data clustered --quant f16" ;;
        i8) flags="--data clustered --quant i8 --data-scale 0.3" ;;
        pq96) flags="--data clustered --pq-subvectors 96" ;;
        b1) flags="--data clustered --quant b1" ;;
        d1536) flags="--data clustered --rows 500000 --dim 1536" ;;
        d128) flags="--data clustered --rows 1000000 --dim 128" ;;
        *) flags="$leg" ;;
      esac
      timeout 600 python bench.py $flags --no-pmc --no-secondary --no-cpu --no-dram-model --build-quality-rows 0 "$@" > "$OUT/${which}_${leg}_$round.json" 2> "$OUT/${which}_${leg}_$round.err"
      python - "$OUT/${which}_${leg}_$round.json" $which $round "$leg" <<'PY'
import json, sys
l = json.load(open(sys.argv[1]))
ph = (l.get("build_roofline") or {}).get("phases_ms", {})
print(json.dumps({"lib": sys.argv[2], "round": int(sys.argv[3]), "leg": sys.argv[4], "qps_wall": round(l["value"]), "kernel_ms": round(l["roofline"]["avg_launch_ms"], 4),
                  "build": round(l["build_vectors_per_s"]), "walk_ms": round(ph.get("walk_ms", 0)), "recall": l.get("recall_at_10")}))
PY
    done
  done
done | tee "$OUT/summary.jsonl"
cp /tmp/lib_tree.so "$LIB"
