cd "${GRAFT_REPO_ROOT:-/root/repo}"
for sp in 100 400 100000000; do
  echo "== LANTERN_GPU_NOTIFY_SPIN_US=$sp"
  LANTERN_GPU_NOTIFY_SPIN_US=$sp LANTERN_BENCH_SECONDARY=headline_scan_service timeout 300 python bench.py --no-pmc --no-cpu --no-dram-model --build-quality-rows 0 --steps 5 2>/dev/null | python -c "
import sys,json
l=json.loads(sys.stdin.read().strip().splitlines()[-1])
e=[x for x in l['secondary'] if x['name']=='headline_scan_service'][0]
print(e['value'], e['over_device_resident'], {k:(v['value'], v['latency_us']['p50'], v['server_side_us']['batch_closed_to_answer_us']) for k,v in e['modes'].items()})"
done
