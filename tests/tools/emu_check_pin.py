"""One committed full-size pin against the HOST EMULATION of the device code (TEST TOOL, CPU): the development loop of the cursor
engine — solves the pin's seeded problem on tests/emu/libksolve_emu.so and compares digest, NodeClaim count and reference
evaluation count.  usage: emu_check_pin.py tests/golden/fullsize/<pin>.json [engine ...]"""
import os as _os
_os.environ.setdefault("KSOLVE_TEST_SOLVER_LIB", "1")
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import parity
from make_fullsize_digests import build_problem
from karpenter_amd.scheduling import NewScheduler

lib = os.environ.get("KSOLVE_EMU_LIB") or parity.build_emu()
g = json.load(open(sys.argv[1]))
prob = build_problem(g["config"], g["pods"], g["types"], g["seed"], g["extra"])
for eng in (sys.argv[2:] or ["auto"]):
    t = time.time(); s = NewScheduler(dict(prob, options=dict(prob["options"], engine=eng)), solver_lib=lib); t1 = time.time(); r = s.Solve(); s.close(); dt = time.time() - t1
    digest, _ = parity.results_digest(r)
    c = r["counters"]
    print(json.dumps({"pin": os.path.basename(sys.argv[1]), "engine": eng, "engine_used": c["engine"], "plan": c.get("cursorMemoryPlan"), "new_scheduler_s": round(t1 - t, 2), "solve_s": round(dt, 2),
                      "pack_ms": r["timings"][0].get("pack_kernel_ms"), "claims": [len(r["newNodeClaims"]), g["claims"]], "digest_matches": digest == g["digest"],
                      "reference_bin_evaluations_match": c["referenceBinEvaluations"] == g["binEvaluations"],
                      "counters": {k: c[k] for k in ("binEvaluations", "fullEvaluations", "slowSorts", "columnResets", "cursorAttempts", "engineFallbackReason") if k in c}}))
