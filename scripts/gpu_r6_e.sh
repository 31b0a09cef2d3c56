# round 6, GPU pass E: ksolve_node_dead0 on the 2D grid + the cursor engine's plan chosen from the rows (exact configs[3] batch: one attempt) +
# the spread engine after the scalar-register diet: the GPU tests that touch them, the 4M-pod pin of the configs[3] shape BY DIGEST on the
# device, the exact 10M leg of bench.py.   usage (GPU box): bash scripts/gpu_r6_e.sh
set -x
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r6e; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_spread_engine.py tests/test_gpu_parity.py -m gpu -x -q -k "spread or sweep or resident or consolidation or window or balanced or config3_p200000 or config4_p1000000 or config2_p2000000" 2>&1 | tail -4 | tee $O/pytest_subset.log
timeout 300 python tests/tools/gpu_check_pin.py tests/golden/fullsize/config3_p1000000_t500_s42.json spread 2>&1 | tail -1 | tee -a $O/pins.log
KSOLVE_TEST_HUGE_PINS=1 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "config4_p4000000" 2>&1 | tail -3 | tee $O/pin_4m_by_digest.log
timeout 900 python bench.py --steps 2 --warmup 1 --topology-pods 0 --batch-problems 0 --components-pods 10000000 --beyond-lds-pods 2000000 --whole-batch-pods 1000000 --sweep-nodes 0 --no-host-engine-baseline --no-cpu-baseline 2>$O/bench_exact.err | tail -1 > $O/bench_exact.json
tail -2 $O/bench_exact.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r6e/bench_exact.json"))
c = d.get("config3_components", {})
print("components", c.get("seconds"), "whole_batch", {k: c.get("whole_batch", {}).get(k) for k in ("seconds", "pack_kernel_ms", "cursor_attempts", "cursor_plan")})
print("exact", {k: c.get("whole_batch_exact", {}).get(k) for k in ("seconds", "pack_kernel_ms", "cursor_attempts", "cursor_plan", "node_claims")})
print("beyond", {k: d.get("config1_beyond_lds", {}).get(k) for k in ("seconds", "first_solve_s", "cursor_attempts", "oracle_pin")})
PY
