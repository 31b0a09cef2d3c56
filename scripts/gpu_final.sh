# Round evidence: GPU parity tests, smoke, the BASELINE configs[1] bench line, rocprofv3 kernel stats of the same command,
# TCC counter passes for the pack kernel, phase breakdown with the profiling build.
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 | tee gpurun_out/pytest_gpu.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee gpurun_out/smoke.log
timeout 1500 python bench.py --steps 2 --warmup 1 2>gpurun_out/bench_1m.err | tail -1 | tee gpurun_out/bench_1m.json
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_1m -o r1 -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --batch-problems 64 > $GRAFT_REPO_ROOT/gpurun_out/prof_bench_1m.log 2>&1)
for f in $(find gpurun_out/prof_1m -name "*kernel_stats*.csv" | head -1); do cut -c1-160 $f | head -8; done
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$c -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --batch-problems 0 > $GRAFT_REPO_ROOT/gpurun_out/pmc_$c.log 2>&1)
  f=$(find gpurun_out/pmc_$c -name "*counter_collection*.csv" | head -1)
  grep "ksolve_pack" $f | head -2
done
bash scripts/gpu_quick.sh
bash scripts/gpu_c3big.sh
