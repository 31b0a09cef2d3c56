# round 3: (1) the classing kernel at three against four wavefronts per SIMD (KSOLVE_ROWHASH_W4), (2) BASELINE configs[3] as ONE
# exact Solve() of the whole batch at 1M (digest-pinned) and 10M pods beside the component split (tests/tools/whole_batch_c3.py)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3h
mkdir -p $O
HEAD="python bench.py --steps 6 --warmup 2 --topology-pods 0 --batch-problems 0 --components-pods 0 --sweep-nodes 0 --no-host-engine-baseline --no-cpu-baseline"
for W in 0 1 0 1; do
if [ $W = 1 ]; then export KSOLVE_ROWHASH_W4=1; else unset KSOLVE_ROWHASH_W4; fi
timeout 600 $HEAD > $O/bench_w4_$W.json 2> $O/bench_w4_$W.err; python -c "
import json; d=json.load(open('$O/bench_w4_$W.json')); r=d['roofline']; print('rowhash w4=$W', r['avg_kernel_ms'], r['frac'], r.get('traffic'), d['value'], d['parity']['oracle_pin']['digest_matches_oracle'])"; tail -2 $O/bench_w4_$W.err
done
unset KSOLVE_ROWHASH_W4
timeout 900 python tests/tools/whole_batch_c3.py --pods 1000000 10000000 --out $O/whole_batch_c3.json > $O/whole_batch_c3.log 2>&1; tail -3 $O/whole_batch_c3.log | cut -c1-1200
