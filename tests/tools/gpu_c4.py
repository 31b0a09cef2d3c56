"""BASELINE configs[3] / configs[4] shapes on the device against the oracle (small sizes, a few seconds)."""
import os as _os
_os.environ.setdefault("KSOLVE_TEST_SOLVER_LIB", "1")   # a test tool: may hand a test build of the solver library to NewScheduler(solver_lib=)
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
t0 = time.time()
import oracle, parity
from karpenter_amd import fixtures as fx, disruption as dz
from karpenter_amd.components import split_by_nodepool
from karpenter_amd.scheduling import NewScheduler, SolveBatch
lib = sys.argv[1] if len(sys.argv) > 1 else None
prob = fx.config4(pods=20000, n_types=144, n_pools=16, seed=2)
parity.assert_same_results(NewScheduler(prob, solver_lib=lib).Solve(), oracle.solve(prob))
parts = split_by_nodepool(prob)
for g, (_, sub) in zip(SolveBatch([NewScheduler(sub, solver_lib=lib) for _, sub in parts]), parts):
    parity.assert_same_results(g, oracle.solve(sub))
print("config4: whole batch and 16 components exact", round(time.time() - t0, 1), "s")
cluster = dz.make_cluster(n_nodes=1000, pods_per_node=6, seed=7)
cands = dz.sort_candidates(cluster, cluster["nodes"])[:4]
got = dz.sweep_batched(cluster, cands, lambda ps: SolveBatch([NewScheduler(p, solver_lib=lib) for p in ps]))
want = dz.sweep(cluster, cands, oracle.solve)
for g, w in zip(got, want):
    assert g["decision"] == w["decision"]
    parity.assert_same_results(g["results"], w["results"])
print("sweep over a 1000-node cluster:", [c["decision"] for c in got], round(time.time() - t0, 1), "s")
