# round 6, GPU pass H: the view / workspace records in LDS for the batched and sweep kernels: every GPU test, the sweep legs and the
# batched leg of bench.py.   usage (GPU box): bash scripts/gpu_r6_h.sh
set -x
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r6h; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee $O/pytest_gpu.log
bash scripts/gpu_r6_sweep.sh r6h 2>&1 | tail -4
timeout 600 python bench.py --steps 1 --warmup 0 --pods 20000 --no-parity-pin --topology-pods 0 --components-pods 10000000 --beyond-lds-pods 0 --whole-batch-exact-pods 0 --whole-batch-pods 0 --sweep-nodes 0 --no-host-engine-baseline --no-cpu-baseline 2>$O/bench_batched.err | tail -1 > $O/bench_batched.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r6h/bench_batched.json"))
print("batched", d.get("batched"), "components", d.get("config3_components", {}).get("seconds"))
PY
