# round 6, GPU pass C: the spread engine's loop in registers (global-address-space pointers, no engine object on its path): GPU tests of the
# engine, the pins, the phase timers of a measurement build, SQ counters.   usage (GPU box): bash scripts/gpu_r6_c.sh [tag]
set -x
cd $GRAFT_REPO_ROOT
T=${1:-r6c}
O=$GRAFT_REPO_ROOT/gpurun_out/$T; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_spread_engine.py -m gpu -x -q 2>&1 | tail -3 | tee $O/pytest_spread.log
for pin in config3_p200000_t500_s42 config3_p500000_t500_s42 config3_p1000000_t500_s42; do
  timeout 300 python tests/tools/gpu_check_pin.py tests/golden/fullsize/$pin.json spread 2>&1 | tail -1 | tee -a $O/pins.log
done
KSOLVE_TEST_SOLVER_LIB=1 KSOLVE_LIB=$GRAFT_REPO_ROOT/karpenter_amd/variants/libksolve_timers.so timeout 300 python tests/tools/gpu_check_pin.py tests/golden/fullsize/config3_p1000000_t500_s42.json spread 2>&1 | tail -1 | tee $O/timers.log
bash scripts/gpu_r6_prof.sh $T 2>&1 | tail -8
