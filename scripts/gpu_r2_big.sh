# BIG engine with the per-count rings (run_order.h): parity tests that reach it, phase counters, the configs[2] shape at 200k and 1M pods
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2
timeout 900 python -m pytest tests -m gpu -x -q -k "big_engine or full_size_digest or topology_mix or config4" > gpurun_out/r2/pytest_big.log 2>&1
tail -3 gpurun_out/r2/pytest_big.log
bash scripts/gpu_c3_phases.sh | grep -E "config3|^sort|^scan|^commit|^can_add|^new_claim|^total"
bash scripts/gpu_c3big.sh 2>&1 | tee gpurun_out/r2/c3big.log
