# round 3: per-path shader-clock counters of the cursor engine (profiling build, -DKSOLVE_PHASE_TIMERS) on configs[1] at 1M pods,
# frontier window off / on / hopping
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r3d
mkdir -p $O
for W in 0 1 2; do
KSOLVE_FAST_WINDOW=$W timeout 600 python - <<'PY' 2>&1 | tee $O/fast_phases_window$W.log
import sys, time, json, os
from karpenter_amd import fixtures as fx
from karpenter_amd.scheduling import NewScheduler
w = os.environ["KSOLVE_FAST_WINDOW"]
if w == "0":
    names = ["hot loop","ev:entry","ev:slot","ev:slowsort","ev:place","ev:newclaim","total","counts","top: block+pend","group tests","group placements","scan placements","-","n group tests","-","n scan placements"]
else:
    names = ["hot loop","ev:entry","ev:slot","ev:slowsort","ev:place","ev:newclaim","total","counts","top: block+pend","window loads","window tests","scan placements (beyond the window, after a flush)","window commits","n window loads","window flushes","n scan placements"]
prob = fx.config2(pods=1000000)
s = NewScheduler(prob, solver_lib=os.path.abspath("karpenter_amd/variants/libksolve_timers.so"))
r = s.Solve(repeat=2, want_results=False)
c = r["counters"]; pc = c["phaseCycles"]
print("window mode", w, c["engine"], "pack ms", [t["pack_kernel_ms"] for t in r["timings"]], "pods", c["pods"], "claims", c["claims"], "V", c["referenceBinEvaluations"])
for n, v in zip(names, pc): print("%-60s %14d   %9.0f /pod" % (n, v, v / c["pods"]))
PY
done
