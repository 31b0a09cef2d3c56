# Round-1 GPU evidence: parity tests, the BASELINE configs[1] bench line, and a rocprofv3 kernel summary of the same command.
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 | tee gpurun_out/pytest_gpu.log
timeout 1200 python bench.py --steps 2 --warmup 1 2>gpurun_out/bench_1m.err | tail -1 | tee gpurun_out/bench_1m.json
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_1m -o r1 -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_bench_1m.log 2>&1)
find gpurun_out/prof_1m -type f | head
for f in $(find gpurun_out/prof_1m -name "*kernel_stats*.csv" | head -1); do cut -c1-160 $f | head -12; done
bash scripts/gpu_quick.sh
