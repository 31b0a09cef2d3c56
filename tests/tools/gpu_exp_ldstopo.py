"""EXPERIMENT (round 4, measurement only — not in the product): the general engine with the topology groups' descriptors and small
mutable state (registered domains, per-domain counts) in LDS instead of HBM, built from a patched copy of csrc/
(profiles/round4/experiments/lds_topology_state.patch) into karpenter_amd/variants/libksolve_ldstopo.so. Solves the configs[2]
shape with the product library and with the variant: same digest, pack kernel time.  usage: gpu_exp_ldstopo.py [pods ...]"""
import os as _os
_os.environ.setdefault("KSOLVE_TEST_SOLVER_LIB", "1")
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import parity
from karpenter_amd import fixtures as fx
from karpenter_amd.scheduling import NewScheduler

variant = os.path.join(ROOT, "karpenter_amd", "variants", "libksolve_ldstopo.so")
for pods in [int(a) for a in sys.argv[1:]] or [60_000, 200_000]:
    p = fx.config3(pods=pods, n_types=500, seed=42)
    row = {"pods": pods}
    for name, lib in (("product", None), ("lds_topology_state", variant)):
        s = NewScheduler(p, solver_lib=lib)
        r = s.Solve(want_results=False); r = s.Solve(want_results=False)
        full = s.Solve(want_results=True); s.close()
        row[name] = {"pack_kernel_ms": round(r["timings"][0]["pack_kernel_ms"], 1), "digest": parity.results_digest(full)[0][:16], "claims": r["counters"]["claims"],
                     "ref_evals": full["counters"]["referenceBinEvaluations"], "engine": r["counters"]["engine"]}
    row["same_results"] = row["product"]["digest"] == row["lds_topology_state"]["digest"] and row["product"]["ref_evals"] == row["lds_topology_state"]["ref_evals"]
    row["speedup"] = round(row["product"]["pack_kernel_ms"] / row["lds_topology_state"]["pack_kernel_ms"], 3)
    print(json.dumps(row))
