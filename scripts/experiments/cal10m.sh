cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/cal10m
CAL_ROWS=10000000 timeout 900 rocprofv3 --kernel-include-regex k_gather --pmc FETCH_SIZE -d gpurun_out/cal10m/pmc -o pmc -- python scripts/calibrate_cache_model.py > gpurun_out/cal10m/model.json 2> gpurun_out/cal10m/stderr.log
echo rc=$?
python - <<'PY'
import glob, sqlite3
for db in glob.glob("gpurun_out/cal10m/pmc/**/*.db", recursive=True):
    cur = sqlite3.connect(db).cursor()
    print([v for (v,) in cur.execute("select value from counters_collection where kernel_name like '%k_gather%' and counter_name='FETCH_SIZE'")])
PY
cat gpurun_out/cal10m/model.json
