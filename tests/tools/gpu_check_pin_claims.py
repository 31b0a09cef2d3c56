"""A full-size pin against the device WITHOUT the digest of every pod's placement (TEST TOOL, GPU): NodeClaim count, pods per NodeClaim in
the reference's order, reference evaluation count and packing cost (bit for bit) from the claims-only Results — seconds instead of a
minute of host time for a 4M-pod Results document; what the last GPU seconds of round 5 could hold.
usage: gpu_check_pin_claims.py tests/golden/fullsize/<pin>.json [engine] [solver_lib (test switch)]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from make_fullsize_digests import build_problem
from karpenter_amd.scheduling import NewScheduler

g = json.load(open(sys.argv[1]))
eng = sys.argv[2] if len(sys.argv) > 2 else "auto"
lib = sys.argv[3] if len(sys.argv) > 3 else None
prob = build_problem(g["config"], g["pods"], g["types"], g["seed"], g["extra"])
t = time.time(); s = NewScheduler(dict(prob, options=dict(prob["options"], engine=eng)), solver_lib=lib); r = s.Solve(want_results="claims"); s.close(); dt = time.time() - t
print(json.dumps({"pin": os.path.basename(sys.argv[1]), "engine": eng, "engine_used": r["counters"]["engine"], "plan": r["counters"].get("cursorMemoryPlan"), "seconds": round(dt, 2),
                  "pack_kernel_ms": round(r["timings"][0].get("pack_kernel_ms", -1), 1),
                  "claims": [len(r["newNodeClaims"]), g["claims"]], "pods_per_claim_match": [c["podCount"] for c in r["newNodeClaims"]] == g["claimPods"],
                  "reference_bin_evaluations_match": r["counters"]["referenceBinEvaluations"] == g["binEvaluations"],
                  "packing_cost_bit_identical": float(r["packingCost"]).hex() == g["packingCost"], "scheduled": r["scheduledPods"]}))
