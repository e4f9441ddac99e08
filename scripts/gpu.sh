#!/bin/bash
# scripts/gpu.sh -- the GPU-box jobs of this repo, one entry per job (run under gpurun: `gpurun --timeout N -- bash scripts/gpu.sh JOB`).
# Everything a job writes goes under gpurun_out/<job>/ ; summaries worth keeping are copied into profiles/ by hand.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
JOB="${1:-help}"; shift || true
OUT="gpurun_out/$JOB"; mkdir -p "$OUT"
case "$JOB" in
  tests)        # the default -m gpu suite (what the driver runs)
    timeout 1500 python -m pytest tests -m gpu -x -q --durations=40 2>&1 | tail -60 > "$OUT/pytest.log"; tail -40 "$OUT/pytest.log" ;;
  tests-new)    # a named subset: bash scripts/gpu.sh tests-new "expr for -k"
    timeout 900 python -m pytest tests -m gpu -x -q -k "$1" 2>&1 | tail -40 | tee "$OUT/pytest.log" ;;
  bench)        # the driver's command; extra flags pass through
    LANTERN_BENCH_PMC_LOG="$OUT" timeout ${BENCH_TIMEOUT:-1200} python bench.py "$@" > "$OUT/line.json" 2> "$OUT/stderr.log"; echo "rc=$?"; tail -5 "$OUT/stderr.log"; head -c 600 "$OUT/line.json" ;;
  bench-trace)  # rocprofv3 --kernel-trace --stats of the search leg only (no counters, no secondary legs)
    timeout 900 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o trace -- python bench.py --no-pmc --no-secondary --no-cpu --no-dram-model --build-quality-rows 0 "$@" > "$OUT/line.json" 2> "$OUT/trace.log"
    python scripts/prof_dump.py "$OUT/trace" > "$OUT/kernel_stats.md"; head -30 "$OUT/kernel_stats.md" ;;
  tests-slow)   # the at-size comparisons marked slow (10M x 768, sequential-build comparisons)
    LANTERN_TEST_SLOW=1 timeout 1700 python -m pytest tests -m "gpu and slow" -x -q --durations=15 2>&1 | tail -40 > "$OUT/pytest.log"; tail -30 "$OUT/pytest.log" ;;
  rccl-double)  # the RCCL transport at worlds 2 / 3 / 8 through tests/fake_rccl (one JSON line per world)
    hipcc -O2 -shared -fPIC -o tests/fake_rccl/librccl_fake.so tests/fake_rccl/fake_rccl.cpp
    for w in 2 3 8; do LANTERN_GPU_RCCL_LIB="$PWD/tests/fake_rccl/librccl_fake.so" timeout 300 python tests/fake_rccl/run_world.py $w; done > "$OUT/worlds.jsonl" 2> "$OUT/stderr.log"
    cut -c1-400 "$OUT/worlds.jsonl" ;;
  bench-10m)    # BASELINE config[3] on one GPU with this run's counters (8192-row batches: rocprofv3 --pmc survives them at 10M rows)
    LANTERN_BENCH_PMC_LOG="$OUT" timeout 1500 python bench.py --rows 10000000 --ef 128 --steps 5 --truth-queries 256 --no-secondary --build-quality-rows 0 --add-batch 8192 "$@" > "$OUT/line.json" 2> "$OUT/stderr.log"
    echo "rc=$?"; tail -12 "$OUT/stderr.log"; head -c 400 "$OUT/line.json" ;;
  calibrate)    # the cache model against the counters on launches with known traffic (a stream, a uniformly random gather)
    for pass in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
      d="$OUT/$(echo $pass | tr ' ' '_')"; rm -rf "$d"
      timeout 400 rocprofv3 --kernel-include-regex k_gather --pmc $pass -d "$d" -o pmc -- python scripts/calibrate_cache_model.py > "$OUT/model.json" 2> "$OUT/stderr.log"
    done
    timeout 400 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o trace -- python scripts/calibrate_cache_model.py > /dev/null 2>> "$OUT/stderr.log"
    python scripts/prof_dump.py "$OUT/trace" > "$OUT/kernel_stats.md"; head -8 "$OUT/kernel_stats.md"
    cat "$OUT/model.json"
    python - <<'PY'
import glob, sqlite3, json
out = {}
for db in glob.glob("gpurun_out/calibrate/trace/**/*.db", recursive=True):
    cur = sqlite3.connect(db).cursor()
    try:
        rows = [r for r in cur.execute("select name, start, end from kernels where name like '%k_gather%' order by start")]
        out["k_gather_durations_us"] = [(e - s) / 1e3 for _, s, e in rows]
    except Exception as ex:
        out["k_gather_durations_error"] = repr(ex)
for db in glob.glob("gpurun_out/calibrate/**/*.db", recursive=True):
    cur = sqlite3.connect(db).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
    order = "dispatch_id" if "dispatch_id" in cols else None
    for name, v in cur.execute("select counter_name, value from counters_collection where kernel_name like '%k_gather%'" + (f" order by {order}" if order else "")):
        out.setdefault(name, []).append(v)
print(json.dumps(out))
open("gpurun_out/calibrate/counters.json", "w").write(json.dumps(out))
PY
    ;;
  *) echo "jobs: tests | tests-slow | tests-new EXPR | bench [flags] | bench-trace [flags] | bench-10m | rccl-double" ;;
esac
