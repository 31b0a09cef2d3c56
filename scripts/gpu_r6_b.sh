# round 6, GPU pass B: the device library linked from several translation units for the first time (every GPU test), the spread engine
# with the step's move made at once + the one-read window: pins and SQ counters.   usage (GPU box): bash scripts/gpu_r6_b.sh
set -x
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r6b; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 | tee $O/pytest_gpu.log
for pin in config3_p200000_t500_s42 config3_p1000000_t500_s42; do
  timeout 300 python tests/tools/gpu_check_pin.py tests/golden/fullsize/$pin.json spread 2>&1 | tail -1 | tee -a $O/pins.log
done
bash scripts/gpu_r6_prof.sh r6b 2>&1 | tail -8
