cd $GRAFT_REPO_ROOT
for data in gaussian clustered; do for s in 1 2; do
python bench.py --data $data --streams $s --steps 40 --no-pmc --no-secondary --no-cpu --no-dram-model --no-gather-ceiling --build-quality-rows 0 2>/dev/null | python -c "
import json,sys; l=json.loads(sys.stdin.read()); print(json.dumps({'data': '$data', 'launches_in_flight': $s, 'qps': round(l['value']), 'ms_per_step': round(l['ms_per_step'],4), 'frac_algorithmic': round(l['roofline']['frac'],4), 'recall': l['recall_at_10']}))"
done; done
