# round 3: the classing kernel after row_diff_far went branch-free (clamped loads); headline leg only, twice
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3j
mkdir -p $O
HEAD="python bench.py --steps 6 --warmup 2 --topology-pods 0 --batch-problems 0 --components-pods 0 --sweep-nodes 0 --no-host-engine-baseline --no-cpu-baseline"
for i in 1 2; do
timeout 600 $HEAD > $O/bench_$i.json 2> $O/bench_$i.err; python -c "
import json; d=json.load(open('$O/bench_$i.json')); r=d['roofline']; print('rowhash', r['avg_kernel_ms'], r['frac'], d['value'], d['parity']['oracle_pin']['digest_matches_oracle'])"; tail -2 $O/bench_$i.err
done
