"""TEST TOOL (GPU): the classing kernel (ksolve_row_hash_coop2) at 1M, 2M and 4M pod rows of the configs[1] mix — the headline's table
(220 MB) is smaller than the MI355X's 256 MiB Infinity Cache, these are not. The pack loop is stopped after one pod (options.maxSteps),
so that only the prepass runs at size. Prints one JSON document: rows, kernel ms (HIP events inside the library), GB/s, fraction of 8 TB/s."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from karpenter_amd import fixtures as fx
from karpenter_amd.scheduling import NewScheduler

out = []
for rows in (1_000_000, 2_000_000, 4_000_000):
    p = fx.config2(pods=rows, n_types=500, seed=42)
    p["options"]["maxSteps"] = 1
    s = NewScheduler(p)
    best = None
    for _ in range(5):
        r = s.Solve(want_results=False)
        t = r["timings"][0]
        c = r["counters"]
        best = t["row_hash_ms"] if best is None else min(best, t["row_hash_ms"])
    s.close()
    row_bytes = (1 if c.get("strictTableShared") else 2) * (8 * c["reqWords"] + 16) + 8 * c["resources"] + 8 + 4
    gbs = rows * row_bytes / (best * 1e-3) / 1e9
    out.append({"rows": rows, "bytes_per_row": row_bytes, "table_MB": rows * row_bytes / 1e6, "row_hash_ms": best, "GBps": gbs, "frac_of_8TBps": gbs / 8000.0})
print(json.dumps({"kernel": "ksolve_row_hash_coop2", "infinity_cache_MiB": 256, "runs": out}))
