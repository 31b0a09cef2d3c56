# round 4, third GPU pass (final build of the round unless a later pass says otherwise): parity tests, smoke, bench line, PMC /
# kernel stats of THIS build (-> profiles/round4/pmc_traffic.json), a rocprofv3 --marker-trace run showing the ROCTx phase ranges,
# and the host phases of NewScheduler at 1M pods (KSCHED_TRACE)
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -8 | tee $O/pytest_gpu.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4 | tee $O/smoke.log
KSCHED_TRACE=1 timeout 300 python -c "
import time
from karpenter_amd import fixtures as fx
from karpenter_amd.scheduling import NewScheduler
p = fx.config2(pods=1000000, n_types=500, seed=42)
for i in range(2):
    t = time.time(); s = NewScheduler(p); print('NewScheduler', time.time() - t); s.close()
" 2>&1 | tail -30 | tee $O/new_scheduler_trace.log
bash scripts/gpu_r4_pmc.sh 2>&1 | tail -30
cp gpurun_out/r4pmc/pmc_traffic.json profiles/round4/pmc_traffic.json
timeout 1800 python bench.py 2>$O/bench.err | tail -1 | tee $O/bench.json
tail -5 $O/bench.err
(cd /tmp && timeout 300 rocprofv3 --marker-trace --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/marker -o m -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --pods 200000 --no-parity-pin --topology-pods 0 --batch-problems 0 --components-pods 0 --sweep-nodes 20000 --sweep-candidates 2000 --sweep-sample 0 --sweep-topology-sample 0 --sweep-windows 0 --no-host-engine-baseline --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/marker.log 2>&1)
find $O/marker -name "*marker*stats*.csv" -o -name "*marker_api_trace*.csv" | head -3
for f in $(find $O/marker -name "*marker*stats*.csv" | head -1); do cut -c1-160 $f | head -14; cp $f $O/roctx_marker_stats.csv; done
rm -rf $O/marker
