# round 6, pass m: the sweep with device-built workspace records (sweep GPU tests, the three configs[4] legs with their pins), what the
# spills of ksolve_pack_sweep4 cost (A/B against the no-spill measurement build), and the spread engine's pins after the first-group
# specialisation of apply_choice.   usage (GPU box): bash scripts/gpu_r6_m.sh [tag]
cd $GRAFT_REPO_ROOT
T=${1:-r6m}
O=$GRAFT_REPO_ROOT/gpurun_out/$T; mkdir -p $O
export TMPDIR=/tmp
bash scripts/gpu_r6_sweep.sh $T 2>&1 | tail -6
bash scripts/gpu_r6_sweep_ab.sh $T 2>&1 | tail -2
timeout 600 python -m pytest tests/test_spread_engine.py -m gpu -x -q 2>&1 | tail -2 | tee $O/pytest_spread.log
for pin in config3_p200000_t500_s42 config3_p1000000_t500_s42; do
  timeout 300 python tests/tools/gpu_check_pin.py tests/golden/fullsize/$pin.json spread 2>&1 | tail -1 | tee -a $O/pins.log
done
