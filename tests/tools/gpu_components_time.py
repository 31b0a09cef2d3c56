import sys, time, os
sys.path.insert(0, os.getcwd())
from karpenter_amd import fixtures as fx
from karpenter_amd.components import split_by_nodepool
from karpenter_amd.scheduling import NewScheduler, SolveBatch
pods = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
whole = fx.config4(pods=pods, n_types=1000, n_pools=16, seed=42)
parts = split_by_nodepool(whole)
scheds = [NewScheduler(sub) for _, sub in parts]
for rep in range(3):
    t = time.perf_counter(); rs = SolveBatch(scheds, want_results="claims"); dt = time.perf_counter() - t
    print("rep", rep, "seconds", round(dt, 3), "timings[0]", rs[0]["timings"][0], flush=True)
