set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rocminfo | grep -E "gfx|Marketing" | head -4
timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -5
timeout 900 python - <<'PY' 2>&1 | tail -30
import sys, time, json
sys.path.insert(0, 'tests')
import oracle, parity
from karpenter_amd import fixtures as fx
from karpenter_amd.scheduling import NewScheduler
for name, prob in [("c1_5000", fx.config1()), ("c2_20000", fx.config2(pods=20000))]:
    want = oracle.solve(prob)
    t=time.time(); got = NewScheduler(prob).Solve(); tg=time.time()-t
    parity.assert_same_results(got, want)
    print(name, "PARITY OK claims", len(got["newNodeClaims"]), "wall %.2fs"%tg, got["timings"], got["counters"], flush=True)
for n in (100000, 1000000):
    prob = fx.config2(pods=n)
    t=time.time(); got = NewScheduler(prob).Solve(repeat=2, want_results=False); tg=time.time()-t
    print("c2", n, "wall %.2fs"%tg, got["timings"], got["counters"], got["scheduledPods"], got["packingCost"], flush=True)
PY
