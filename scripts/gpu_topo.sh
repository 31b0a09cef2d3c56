cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/pytest_gpu.log
cat gpurun_out/pytest_gpu.log
timeout 600 python - <<'PY' 2>&1 | tee gpurun_out/c3.log
import sys, time, json
from karpenter_amd import fixtures as fx
from karpenter_amd.scheduling import NewScheduler
for pods, anti in ((20000, 1000), (100000, 3000)):
    prob = fx.config3(pods=pods, n_types=500, seed=42, anti_affinity_pods=anti)
    s = NewScheduler(prob)
    r = s.Solve(repeat=1, want_results=False)
    c = r["counters"]; pc = c["phaseCycles"]
    names = ["queue","class_fetch","sort","scan","rec_load","can_add","commit","new_claim","dead_mark","try_sched","total","ca_pre","ca_merge","ca_total","ca_filter"]
    print("config3", pods, "pack ms", [t["pack_kernel_ms"] for t in r["timings"]], "claims", c["claims"], "scheduled", r["scheduledPods"], "evals", c["binEvaluations"], "V", c["referenceBinEvaluations"])
    for n, v in zip(names, pc): print("%-12s %14d cycles  %10.0f /pod" % (n, v, v / c["pods"]))
PY
