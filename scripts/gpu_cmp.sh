cd $GRAFT_REPO_ROOT
timeout 900 python - <<'PY'
from karpenter_amd import fixtures as fx
from karpenter_amd.scheduling import NewScheduler, SolveBatch
prob = fx.config2(pods=200000)
s = NewScheduler(prob)
r = s.Solve(repeat=2, want_results=False)
print("kernarg kernel pack ms", [t["pack_kernel_ms"] for t in r["timings"]])
for _ in range(2):
    rb = SolveBatch([s], want_results=False)
    print("batch kernel (n=1) pack ms", rb[0]["timings"][0]["pack_kernel_ms"], "total cycles/pod", rb[0]["counters"]["phaseCycles"][10] / 200000)
PY
