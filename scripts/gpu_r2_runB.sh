# round 2, run B: new GPU tests, the 10k-node resident consolidation sweep, classing timings via the default bench line
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2b
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r2b
timeout 1200 python -m pytest tests -m gpu -x -q -k "resident or override or host_ports or example or config1 or edge or existing" > $O/pytest_new.log 2>&1
tail -4 $O/pytest_new.log
timeout 900 python tests/tools/consolidation_sweep.py 10000 256 6 --types 500 --out $O/consolidation_sweep_10k_256.json > $O/sweep.log 2>&1
tail -c 1800 $O/sweep.log
timeout 900 python tests/tools/consolidation_sweep.py 20000 1024 4 --types 500 --out $O/consolidation_sweep_20k_1024.json > $O/sweep2.log 2>&1
tail -c 900 $O/sweep2.log
timeout 900 python bench.py --topology-pods 0 --batch-problems 0 --components-pods 0 --no-host-engine-baseline --no-cpu-baseline > $O/bench_short.json 2> $O/bench_short.err
python - $O <<'PY'
import json, sys
d = json.loads(open(sys.argv[1] + "/bench_short.json").read().strip().splitlines()[-1])
print(d["value"], d["roofline_stream"], d["phases_ms"])
PY
BENCH="python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --topology-pods 0 --batch-problems 0 --components-pods 0 --no-host-engine-baseline --no-cpu-baseline"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- $BENCH > $O/stats.log 2>&1)
find $O/stats -name "*kernel_stats*.csv" | head -1 | xargs -r grep -E "ksolve_(row|class|it_index|sort_key)" | cut -c1-160
