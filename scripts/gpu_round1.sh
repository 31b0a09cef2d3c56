set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 | tee gpurun_out/pytest_gpu.log
timeout 900 python bench.py --pods 200000 --steps 2 --warmup 1 --cpu-sample 20000 2>&1 | tail -3 | tee gpurun_out/bench_200k.json
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_200k -o r1 -- python $GRAFT_REPO_ROOT/bench.py --pods 200000 --steps 2 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_bench.log 2>&1)
find gpurun_out/prof_200k -type f | head; for f in $(find gpurun_out/prof_200k -name "*kernel_stats*.csv" | head -1); do head -20 $f; done
