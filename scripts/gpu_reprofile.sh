# Bench line + rocprofv3 kernel stats of the same command for the build in the tree (no PMC passes, no tests).
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf gpurun_out/prof_1m
(cd /tmp && timeout 95 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_1m -o r1 -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --batch-problems 64 > $GRAFT_REPO_ROOT/gpurun_out/prof_bench_1m.log 2>&1)
for f in $(find gpurun_out/prof_1m -name "*kernel_stats*.csv" | head -1); do cut -c1-160 $f | head -6; done
timeout 110 python bench.py --steps 2 --warmup 1 2>gpurun_out/bench_1m.err | tail -1 | tee gpurun_out/bench_1m.json
