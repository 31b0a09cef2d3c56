cd $GRAFT_REPO_ROOT
timeout 600 python - <<'PY'
import sys, time, json
from karpenter_amd import fixtures as fx
from karpenter_amd.scheduling import NewScheduler
prob = fx.config2(pods=200000)
s = NewScheduler(prob)
r = s.Solve(repeat=2, want_results=False)
c = r["counters"]; pc = c["phaseCycles"]
names = ["queue","class_fetch","sort","scan","rec_load","can_add","commit","new_claim","dead_mark","try_sched","total","ca_pre","ca_merge","ca_total","ca_filter"]
print("pack ms", [t["pack_kernel_ms"] for t in r["timings"]], "pods", c["pods"], "evals", c["binEvaluations"])
for n, v in zip(names, pc): print("%-12s %12d cycles  %8.0f /pod" % (n, v, v / c["pods"]))
PY
