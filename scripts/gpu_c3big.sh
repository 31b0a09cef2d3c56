cd $GRAFT_REPO_ROOT
timeout 1200 python - <<'PY'
import time
from karpenter_amd import fixtures as fx
from karpenter_amd.scheduling import NewScheduler
for pods in (200000, 1000000):
    prob = fx.config3(pods=pods, n_types=500, seed=42)
    t = time.time(); s = NewScheduler(prob); tf = time.time() - t
    t = time.time(); r = s.Solve(want_results=False); ts = time.time() - t
    c = r["counters"]
    print("config3 full shape", pods, "pods: flatten", round(tf, 1), "s, solve", round(ts, 2), "s, pack kernel ms", [round(x["pack_kernel_ms"]) for x in r["timings"]],
          "claims", c["claims"], "scheduled", r["scheduledPods"], "evals", c["binEvaluations"], "V", c["referenceBinEvaluations"], flush=True)
    s.close()
PY
