# NOTE: as run at commit 361dd55, when the cursor engine still had its frontier window (KSOLVE_FAST_WINDOW; removed after these measurements, profiles/README.md)
# round 3, third GPU run: the frontier window of the cursor engine (A/B against the group path), the shared strict table of the
# classing kernel, the new parity tests (volume limits, topology probes, the 10k-node sweep with topology pods)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3c
mkdir -p $O
HEAD="python bench.py --steps 5 --warmup 2 --topology-pods 0 --batch-problems 0 --components-pods 0 --sweep-nodes 0 --no-host-engine-baseline --no-cpu-baseline"
timeout 600 $HEAD > $O/bench_window.json 2> $O/bench_window.err; python -c "
import json; d=json.load(open('$O/bench_window.json')); print('window ON ', d['value'], d['ms_per_step'], d['roofline']['avg_kernel_ms'], d['roofline']['frac'], d['parity']['oracle_pin'])"; tail -2 $O/bench_window.err
KSOLVE_FAST_WINDOW=2 timeout 600 $HEAD > $O/bench_hop.json 2> $O/bench_hop.err; python -c "
import json; d=json.load(open('$O/bench_hop.json')); print('window HOP', d['value'], d['ms_per_step'], d['parity']['oracle_pin'])"; tail -2 $O/bench_hop.err
KSOLVE_FAST_WINDOW=0 timeout 600 $HEAD > $O/bench_nowindow.json 2> $O/bench_nowindow.err; python -c "
import json; d=json.load(open('$O/bench_nowindow.json')); print('window OFF', d['value'], d['ms_per_step'], d['roofline']['avg_kernel_ms'], d['roofline']['frac'])"; tail -2 $O/bench_nowindow.err
KSOLVE_TEST_NO_SHARED_STRICT=1 timeout 600 $HEAD > $O/bench_twotables.json 2> $O/bench_twotables.err; python -c "
import json; d=json.load(open('$O/bench_twotables.json')); print('two staged tables', d['roofline']['avg_kernel_ms'])"
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log
