# round 5, GPU pass K: four-row problems with class slots clustered by compatibility (the refresh skips the rows a claim cannot
# accept any class of) — the x16 pin on two plans, the configs[3] legs of the bench line (components, whole batch 1M, exact 10M)
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5k; mkdir -p $O
export TMPDIR=/tmp
for eng in auto cursor-hbm; do timeout 300 python tests/tools/gpu_check_pin.py tests/golden/fullsize/config4_p1000000_t1000_s42_x16.json $eng 2>&1 | tail -1 | tee -a $O/pins.log; done
timeout 300 python tests/tools/gpu_check_pin.py tests/golden/fullsize/config2_p1000000_t500_s42.json auto 2>&1 | tail -1 | tee -a $O/pins.log
timeout 900 python bench.py --steps 3 --topology-pods 0 --beyond-lds-pods 0 --batch-problems 0 --sweep-nodes 0 --no-cpu-baseline --no-host-engine-baseline 2>$O/bench_c3.err | tail -1 > $O/bench_c3.json
tail -3 $O/bench_c3.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r5k/bench_c3.json"))
print("value", d["value"], "pack", d["pack_kernel"]["avg_kernel_ms"])
c = d.get("config3_components", {}); print("components", {k: c.get(k) for k in ("seconds", "value", "pack_kernel_ms")}); print("exact", {k: (c.get("whole_batch_exact") or {}).get(k) for k in ("seconds", "pack_kernel_ms", "node_claims", "cursor_memory_plan", "cursor_attempts")}); print("whole 1M", {k: (c.get("whole_batch") or {}).get(k) for k in ("seconds", "pack_kernel_ms", "oracle_pin")})
PY
