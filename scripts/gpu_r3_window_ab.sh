# round 3: the cursor engine's frontier window off / on / hopping on configs[1] (1M pods), digest-checked by bench.py
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3e
mkdir -p $O
HEAD="python bench.py --steps 4 --warmup 1 --topology-pods 0 --batch-problems 0 --components-pods 0 --sweep-nodes 0 --no-host-engine-baseline --no-cpu-baseline"
for W in 0 1 2; do
KSOLVE_FAST_WINDOW=$W timeout 600 $HEAD > $O/bench_w$W.json 2> $O/bench_w$W.err; python -c "
import json; d=json.load(open('$O/bench_w$W.json')); print('window $W', d['value'], d['ms_per_step'], d['pack_kernel']['avg_kernel_ms'], d['parity']['oracle_pin']['digest_matches_oracle'])"; tail -2 $O/bench_w$W.err
done
