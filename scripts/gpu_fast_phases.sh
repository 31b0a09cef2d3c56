# phase counters of the cursor engine (profiling build, -DKSOLVE_PHASE_TIMERS) on configs[1] at 200k and 1M pods
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2
timeout 900 python - <<'PY' 2>&1 | tee gpurun_out/r2/fast_phases.log
import sys, time, json, os
from karpenter_amd import fixtures as fx
from karpenter_amd.scheduling import NewScheduler
names = ["hot loop","ev:entry","ev:slot","ev:slowsort","ev:place","ev:newclaim","total","counts","top: block+pend","group tests","group placements","window placements","-","n group tests","n group placements","n window placements"]
for label, prob in (("config2 1M", fx.config2(pods=1000000)),):
    s = NewScheduler(prob, solver_lib=os.path.abspath("karpenter_amd/variants/libksolve_timers.so"))
    r = s.Solve(repeat=2, want_results=False)
    c = r["counters"]; pc = c["phaseCycles"]
    print(label, c["engine"], "pack ms", [t["pack_kernel_ms"] for t in r["timings"]], "pods", c["pods"], "claims", c["claims"], "steps", pc[21], "V", c["referenceBinEvaluations"], r["timings"])
    for n, v in zip(names, pc): print("%-12s %14d cycles  %9.0f /pod" % (n, v, v / c["pods"]))
PY
