# round 5, GPU pass I: the two-wavefront cursor kernel (ksolve_pack_fast2) against the one-wavefront form — phase timers, the 1M pin on
# both, the reduced bench line, the cursor / batch / edge GPU tests
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5i; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python - <<'PY' 2>&1 | tee $O/fast_phases.log
import os
os.environ["KSOLVE_TEST_SOLVER_LIB"] = "1"
from karpenter_amd import fixtures as fx
from karpenter_amd.scheduling import NewScheduler
names = ["slow loop (inside)", "ev:refresh", "ev:slot", "ev:slowsort", "ev:place", "ev:newclaim", "total", "counts(packed)", "fast loop", "n fast calls", "slow loop (calls)", "n slow calls", "n refresh", "n slot", "n slowsort | n place<<32", "n newclaim"]
for label, eng in (("config2 1M two wavefronts", "cursor"), ("config2 1M one wavefront", "cursor-solo")):
    prob = fx.config2(pods=1000000)
    prob = dict(prob, options=dict(prob["options"], engine=eng))
    s = NewScheduler(prob, solver_lib=os.path.abspath("karpenter_amd/variants/libksolve_timers.so"))
    r = s.Solve(repeat=2, want_results=False)
    c = r["counters"]; pc = c["phaseCycles"]
    print(label, c["engine"], "plan", c.get("cursorMemoryPlan"), "fallback", c.get("engineFallbackReason"), "pack ms", [round(t["pack_kernel_ms"], 1) for t in r["timings"]], "pods", c["pods"], "claims", c["claims"])
    for n, v in zip(names, pc):
        print("%-28s %14d  %9.1f /pod" % (n, v, v / c["pods"]))
    s.close()
PY
for eng in auto cursor-solo; do timeout 300 python tests/tools/gpu_check_pin.py tests/golden/fullsize/config2_p1000000_t500_s42.json $eng 2>&1 | tail -1 | tee -a $O/pins.log; done
timeout 300 python tests/tools/gpu_check_pin.py tests/golden/fullsize/config1_p5000_t50_s42.json auto 2>&1 | tail -1 | tee -a $O/pins.log
timeout 300 python tests/tools/gpu_check_pin.py tests/golden/fullsize/config2_p200000_t500_s42.json auto 2>&1 | tail -1 | tee -a $O/pins.log
timeout 900 python bench.py --steps 5 --topology-pods 0 --components-pods 0 --beyond-lds-pods 0 --whole-batch-exact-pods 0 --whole-batch-pods 0 --batch-problems 0 --sweep-nodes 0 --no-cpu-baseline --no-host-engine-baseline 2>$O/bench_reduced.err | tail -1 > $O/bench_reduced.json
tail -3 $O/bench_reduced.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r5i/bench_reduced.json"))
print("value", d["value"], "ms_per_step", d["ms_per_step"], "pack ms", d["pack_kernel"]["avg_kernel_ms"])
PY
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "edge or cursor or batch or hundred or ragged or empty" 2>&1 | tail -5 | tee $O/pytest_subset.log
