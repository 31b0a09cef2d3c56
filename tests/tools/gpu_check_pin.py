"""One committed (or candidate) full-size pin against the device (TEST TOOL, GPU): solves the pin's seeded problem through the product
library and compares digest, NodeClaim count and reference evaluation count.  usage: gpu_check_pin.py tests/golden/fullsize/<pin>.json [engine]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import parity
from make_fullsize_digests import build_problem
from karpenter_amd.scheduling import NewScheduler

g = json.load(open(sys.argv[1]))
eng = sys.argv[2] if len(sys.argv) > 2 else "auto"
prob = build_problem(g["config"], g["pods"], g["types"], g["seed"], g["extra"])
t = time.time(); s = NewScheduler(dict(prob, options=dict(prob["options"], engine=eng)), **({"solver_lib": os.environ["KSOLVE_LIB"]} if os.environ.get("KSOLVE_LIB") else {})); r = s.Solve(); s.close(); dt = time.time() - t
digest, _ = parity.results_digest(r)
print(json.dumps({"pin": os.path.basename(sys.argv[1]), "engine": eng, "engine_used": r["counters"]["engine"], "plan": r["counters"].get("cursorMemoryPlan"), "seconds": round(dt, 2), "pack_kernel_ms": r["timings"][0].get("pack_kernel_ms"), "fallback_reason": r["counters"].get("engineFallbackReason"), **({"phase_cycles": r["counters"].get("phaseCycles")} if os.environ.get("KSOLVE_LIB") else {}),
                  "claims": [len(r["newNodeClaims"]), g["claims"]], "digest_matches": digest == g["digest"],
                  "reference_bin_evaluations_match": r["counters"]["referenceBinEvaluations"] == g["binEvaluations"]}))
