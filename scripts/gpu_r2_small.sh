# cursor engine on small and large problems after the no-descent shortcut: digests, 1M headline, the 512 x 20k batched leg
cd $GRAFT_REPO_ROOT
python tests/tools/gpu_engines_cmp.py 200000 500 2>&1 | tail -3
python bench.py --steps 3 --warmup 1 --topology-pods 0 --components-pods 0 --no-host-engine-baseline --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('value', round(d['value']), 'pack ms', d['phases_ms']['pack_kernel_ms'], 'slow sorts', d['counters']['slowSorts'], 'digest ok', d['parity']['oracle_pin']['digest_matches_oracle']); print('batched', d['batched'])"
