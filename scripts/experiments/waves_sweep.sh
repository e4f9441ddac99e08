cd $GRAFT_REPO_ROOT
for data in gaussian clustered; do for w in 0 4 8 2; do
python bench.py --data $data --waves $w --no-pmc --no-secondary --no-cpu --no-dram-model --no-gather-ceiling --build-quality-rows 0 2>/dev/null | python -c "
import json,sys; l=json.loads(sys.stdin.read()); print('$data', 'waves', $w, round(l['value']), round(l['roofline']['avg_launch_ms'],4))"
done; done
