# round 5, GPU pass F: plan 1 with the order's snapshot in HBM (the exact configs[3] batch at 10M pods fits its LDS order), the slow
# sort's snapshot in 16-byte pieces; the measurement build without the fence behind HBM stores (pins on plans 1 / 2, the 10M batch)
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5f; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python bench.py --steps 5 --topology-pods 0 --components-pods 0 --beyond-lds-pods 0 --whole-batch-exact-pods 0 --whole-batch-pods 0 --batch-problems 0 --sweep-nodes 0 --no-cpu-baseline --no-host-engine-baseline 2>$O/bench_reduced.err | tail -1 > $O/bench_reduced.json
tail -3 $O/bench_reduced.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r5f/bench_reduced.json"))
print("value", d["value"], "ms_per_step", d["ms_per_step"], "pack ms", d["pack_kernel"]["avg_kernel_ms"])
PY
for eng in auto cursor-wide cursor-hbm; do timeout 300 python tests/tools/gpu_check_pin.py tests/golden/fullsize/config2_p1000000_t500_s42.json $eng 2>&1 | tail -1 | tee -a $O/pins.log; done
for eng in auto cursor-hbm; do timeout 300 python tests/tools/gpu_check_pin.py tests/golden/fullsize/config4_p1000000_t1000_s42_x16.json $eng 2>&1 | tail -1 | tee -a $O/pins.log; done
timeout 300 python tests/tools/gpu_check_pin.py tests/golden/fullsize/config2_p2000000_t500_s42.json auto 2>&1 | tail -1 | tee -a $O/pins.log
timeout 600 python tests/tools/whole_batch_c3.py --pods 10000000 --no-components --out $O/whole_batch_c3_10m.json 2>&1 | tail -2
# the measurement build: no fence behind HBM stores in the fast loop
python - <<'PY' 2>&1 | tee gpurun_out/r5f/nofence.log
import os, json, sys, time
os.environ["KSOLVE_TEST_SOLVER_LIB"] = "1"
sys.path.insert(0, "tests"); sys.path.insert(0, "tests/golden")
import parity
from make_fullsize_digests import build_problem
from karpenter_amd.scheduling import NewScheduler
lib = os.path.abspath("karpenter_amd/variants/libksolve_nofence.so")
for pin, engines, reps in (("config2_p1000000_t500_s42", ("cursor-wide", "cursor-hbm"), 10), ("config4_p1000000_t1000_s42_x16", ("auto", "cursor-hbm"), 5), ("config2_p2000000_t500_s42", ("auto",), 3)):
    g = json.load(open(f"tests/golden/fullsize/{pin}.json"))
    prob = build_problem(g["config"], g["pods"], g["types"], g["seed"], g["extra"])
    for eng in engines:
        s = NewScheduler(dict(prob, options=dict(prob["options"], engine=eng)), solver_lib=lib)
        oks, ms = [], []
        for _ in range(reps):
            r = s.Solve()
            d, _ = parity.results_digest(r)
            oks.append(d == g["digest"] and r["counters"]["referenceBinEvaluations"] == g["binEvaluations"]); ms.append(round(r["timings"][0]["pack_kernel_ms"], 1))
        print(json.dumps({"pin": pin, "engine": eng, "plan": r["counters"].get("cursorMemoryPlan"), "no_fence_build": True, "runs": reps, "all_digests_match": all(oks), "pack_ms": ms}))
        s.close()
PY
