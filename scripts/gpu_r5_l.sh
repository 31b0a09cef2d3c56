# round 5, GPU pass L: claim records in HBM (plan 1) — the next pod's claims gathered before this pod's predicates, the claim just
# changed patched from registers, the step's one fence in front of its stores: the x16 pin, the 2M pin, the configs[3] legs and the
# beyond-LDS leg of the bench line; the same with the order in HBM too (plan 2) on a variant library (KS_FAST_EARLY_GATHER_MASK=6)
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5l; mkdir -p $O
export TMPDIR=/tmp
for eng in auto cursor-hbm; do timeout 300 python tests/tools/gpu_check_pin.py tests/golden/fullsize/config4_p1000000_t1000_s42_x16.json $eng 2>&1 | tail -1 | tee -a $O/pins.log; done
for eng in auto cursor-wide cursor-hbm; do timeout 300 python tests/tools/gpu_check_pin.py tests/golden/fullsize/config2_p1000000_t500_s42.json $eng 2>&1 | tail -1 | tee -a $O/pins.log; done
timeout 300 python tests/tools/gpu_check_pin.py tests/golden/fullsize/config2_p2000000_t500_s42.json auto 2>&1 | tail -1 | tee -a $O/pins.log
timeout 900 python bench.py --steps 3 --topology-pods 0 --batch-problems 0 --sweep-nodes 0 --no-cpu-baseline --no-host-engine-baseline 2>$O/bench_c3.err | tail -1 > $O/bench_c3.json
tail -3 $O/bench_c3.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r5l/bench_c3.json"))
print("value", d["value"], "pack", d["pack_kernel"]["avg_kernel_ms"])
b = d.get("config1_beyond_lds", {}); print("beyond", {k: b.get(k) for k in ("seconds", "value", "pack_kernel_ms", "us_per_pod")}, (b.get("oracle_pin") or {}).get("digest_matches_oracle"))
c = d.get("config3_components", {}); print("components", {k: c.get(k) for k in ("seconds", "value", "pack_kernel_ms")}); print("exact", {k: (c.get("whole_batch_exact") or {}).get(k) for k in ("seconds", "pack_kernel_ms", "node_claims", "cursor_memory_plan", "cursor_attempts")}); print("whole 1M", {k: (c.get("whole_batch") or {}).get(k) for k in ("seconds", "pack_kernel_ms", "oracle_pin")})
PY
# the variant: plan 2 with the early gather too
[ -f karpenter_amd/variants/libksolve_eg6.so ] && KSOLVE_TEST_SOLVER_LIB=1 timeout 600 python - <<'PY' 2>&1 | tee $O/variant_eg6.log
import json, os, sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests"); sys.path.insert(0, "tests/golden")
import parity
from make_fullsize_digests import build_problem
from karpenter_amd.scheduling import NewScheduler
for pin, eng in (("config4_p1000000_t1000_s42_x16.json", "cursor-hbm"), ("config2_p1000000_t500_s42.json", "cursor-hbm"), ("config2_p2000000_t500_s42.json", "cursor-hbm")):
    g = json.load(open("tests/golden/fullsize/" + pin))
    prob = build_problem(g["config"], g["pods"], g["types"], g["seed"], g["extra"])
    prob = dict(prob, options=dict(prob["options"], engine=eng))
    for label, lib in (("product", None), ("early-gather on plan 2", os.path.abspath("karpenter_amd/variants/libksolve_eg6.so"))):
        s = NewScheduler(prob, solver_lib=lib); r = s.Solve(repeat=2); s.close()
        digest, _ = parity.results_digest(r)
        print(json.dumps({"pin": pin, "lib": label, "plan": r["counters"].get("cursorMemoryPlan"), "pack_ms": [round(t["pack_kernel_ms"], 1) for t in r["timings"]], "digest_matches": digest == g["digest"]}), flush=True)
PY
