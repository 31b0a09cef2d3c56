"""Where the time between Solve() and re-hydrated NodeClaims goes at 1M pods (TEST TOOL, GPU): the C call with and without results,
the JSON document, the by-position arrays."""
import ctypes, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from karpenter_amd import fixtures as fx
import karpenter_amd.scheduling as ks

p = fx.config2(pods=1_000_000, n_types=500, seed=42)
t = time.perf_counter(); s = ks.NewScheduler(p); print("new_scheduler_s", round(time.perf_counter() - t, 4))
t = time.perf_counter(); s2 = ks.NewScheduler(p); print("new_scheduler_s (second session)", round(time.perf_counter() - t, 4)); s2.close()
for w in (False, False, "claims-compact", "claims-compact", False):
    t = time.perf_counter(); ptr = s._lib.ksched_solve(s._session, ks._want(w)); t1 = time.perf_counter()
    raw = ctypes.string_at(ptr); t2 = time.perf_counter(); doc = json.loads(raw.decode()); t3 = time.perf_counter(); s._lib.ksched_free(ptr)
    row = {"want": w, "ksched_solve_s": round(t1 - t, 4), "pack_kernel_ms": round(doc["timings"][0]["pack_kernel_ms"], 1), "json_bytes": len(raw), "string_at_s": round(t2 - t1, 4), "json_loads_s": round(t3 - t2, 4),
           "timings": {k: round(v, 2) for k, v in doc["timings"][0].items() if isinstance(v, (int, float))}}
    if w:
        t = time.perf_counter(); po = s.PodsByClaim(len(doc["newNodeClaims"])); row["pods_by_claim_s"] = round(time.perf_counter() - t, 4)
    print(json.dumps(row))
s.close()
