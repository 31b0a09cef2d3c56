# round 3: rocprofv3 kernel stats of the configs[4] legs (single-node sweep of 10k probes + multi-node windows) — bench.py with the other legs off
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3n
mkdir -p $O
BENCH="python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --pods 20000 --topology-pods 0 --batch-problems 0 --components-pods 0 --no-host-engine-baseline --no-cpu-baseline --sweep-sample 0 --no-parity-pin"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o sweep -- $BENCH > $O/stats.log 2>&1)
find $O/stats -name "*kernel_stats*.csv" | head -1 | xargs -r -I{} cp {} $O/rocprofv3_kernel_stats_sweep.csv
grep -E "Name|ksolve" $O/rocprofv3_kernel_stats_sweep.csv | cut -c1-200 | head -20
rm -rf $O/stats
