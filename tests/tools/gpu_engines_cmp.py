"""GPU check (TEST TOOL): the cursor engine and the general engine on the same configs[1]-shaped problem, on the device —
same digest, and the pack-kernel time of each. usage: python tests/tools/gpu_engines_cmp.py [pods] [types]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import parity
from karpenter_amd import fixtures as fx
from karpenter_amd.scheduling import NewScheduler

pods = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
types = int(sys.argv[2]) if len(sys.argv) > 2 else 500
general = "--no-general" not in sys.argv
p = fx.config2(pods=pods, n_types=types, seed=42)
out = {}
for eng in (("cursor", "general") if general else ("cursor",)):
    s = NewScheduler(dict(p, options=dict(p["options"], engine=eng)))
    s.Solve(want_results=False)
    t = time.time()
    r = s.Solve()
    dt = time.time() - t
    d, _ = parity.results_digest(r)
    out[eng] = d
    print(eng, "engine_used", r["counters"]["engine"], "fallback", r["counters"]["engineFallbackReason"], "claims", r["counters"]["claims"],
          "pack_kernel_ms", r["timings"][0]["pack_kernel_ms"], "solve_s", round(dt, 3), "digest", d[:16], "steps", r["counters"]["phaseCycles"][21], flush=True)
    s.close()
if general:
    assert out["cursor"] == out["general"], out
    print("digests equal")
