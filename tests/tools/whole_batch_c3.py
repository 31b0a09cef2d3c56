"""BASELINE configs[3] (pods x 1000 types x 16 NodePools) as ONE Solve() of the whole batch — the exact form, bit-identical to
the reference — beside the component split bench.py reports. At 1M pods the Results digest is checked
against the oracle's offline pin (tests/golden/fullsize/config4_p1000000_t1000_s42_x16.json); at 10M pods the pin is
used when committed (the round-4 oracle needs ~10 h for it), and the claim invariants (tests/invariants.py) are checked either way. Round 4:
the 10M batch ends with 27,345 in-flight NodeClaims and runs on the cursor engine with claim state and claim order in HBM (plan 2) instead
of the general engine; the first Solve() of the handle includes the attempts with the smaller plans, the second starts on the plan that held.
Usage: python tests/tools/whole_batch_c3.py [--pods 1000000 10000000] [--out file.json]"""
import os as _os
_os.environ.setdefault("KSOLVE_TEST_SOLVER_LIB", "1")   # a test tool: may hand a test build of the solver library to NewScheduler(solver_lib=)
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from karpenter_amd import fixtures as fx                    # noqa: E402
from karpenter_amd.components import split_by_nodepool      # noqa: E402
from karpenter_amd.scheduling import NewScheduler, SolveBatch  # noqa: E402
import parity                                               # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pods", type=int, nargs="+", default=[1000000, 10000000])
    ap.add_argument("--out", default=None)
    ap.add_argument("--solver-lib", default=None)
    ap.add_argument("--no-components", action="store_true")
    a = ap.parse_args()
    out = []
    for pods in a.pods:
        prob = fx.config4(pods=pods, n_types=1000, n_pools=16, seed=42)
        if pods > 1000000:   # device capacity for in-flight NodeClaims (default: one per pod); the batch needs about 2.8 per 1000 pods
            prob["options"] = dict(prob["options"], maxClaims=max(65536, pods // 100))
        t0 = time.time(); s = NewScheduler(prob, solver_lib=a.solver_lib); t_new = time.time() - t0
        want_results = True if pods <= 1000000 else "claims"
        t0 = time.time(); r0 = s.Solve(want_results=False); t_first = time.time() - t0
        t0 = time.time(); r = s.Solve(want_results=want_results); t_solve = time.time() - t0
        c = r["counters"]
        row = {"pods": pods, "new_scheduler_s": t_new, "first_solve_s": t_first, "first_solve_attempts": r0["counters"].get("cursorAttempts"), "solve_s": t_solve, "pods_per_s": pods / t_solve, "engine": c["engine"],
               "cursor_memory_plan": c.get("cursorMemoryPlan"), "engine_fallback_reason": c.get("engineFallbackReason"), "slow_sorts": c.get("slowSorts"),
               "node_claims": c["claims"], "pack_kernel_ms": r["timings"][-1]["pack_kernel_ms"] if r.get("timings") else None,
               "reference_bin_evaluations": c["referenceBinEvaluations"], "pods_scheduled": c["pods"] - len(r.get("podErrors", {})) }
        if "packingVector" in r:
            row["packing_cost_per_hour"] = sum(d for _, _, d in r["packingVector"]); row["claims_from_vector"] = sum(c for _, c, _ in r["packingVector"])
        pin = os.path.join(ROOT, "tests", "golden", "fullsize", f"config4_p{pods}_t1000_s42_x16.json")
        import invariants
        row["invariants"] = invariants.check_claims(prob, r, expect_pods=pods)
        if os.path.exists(pin):
            g = json.load(open(pin))
            if want_results is not True:
                r = s.Solve(want_results=True)
            digest, _ = parity.results_digest(r)
            row["oracle_pin"] = {"pin": os.path.relpath(pin, ROOT), "digest_matches_oracle": digest == g["digest"], "claims_match": len(r["newNodeClaims"]) == g["claims"],
                                 "reference_bin_evaluations_match": c["referenceBinEvaluations"] == g["binEvaluations"], "oracle_seconds_offline": g.get("oracleSeconds")}
        s.close()
        if a.no_components:
            print(json.dumps(row), flush=True); out.append(row); continue
        # the same batch as 16 NodePool components in one launch (what bench.py's config3_components times)
        subs = [sub for _, sub in split_by_nodepool(prob)]
        hs = [NewScheduler(p, solver_lib=a.solver_lib) for p in subs]
        t0 = time.time(); rs = SolveBatch(hs, want_results="claims"); t_comp = time.time() - t0
        row["components"] = {"n": len(subs), "solve_batch_s": t_comp, "pods_per_s": pods / t_comp, "node_claims": sum(x["counters"]["claims"] for x in rs), "engines": sorted({x["counters"]["engine"] for x in rs}),
                             "packing_cost_per_hour": sum(d for x in rs for _, _, d in x.get("packingVector", []))}
        for h in hs: h.close()
        print(json.dumps(row), flush=True)
        out.append(row)
    if a.out:
        json.dump(out, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
