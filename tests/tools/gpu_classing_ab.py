"""GPU measurement (TEST TOOL): the classing kernel variants on BASELINE configs[1] (1M pod rows), one resident scheduler,
the kernel picked per Solve() through the launcher's A/B switch (KSOLVE_ROWHASH_KERNEL); HIP-event time of the row-hash
kernel and of the whole classing phase, and the digest of the results against the oracle's full-size pin.
usage: python tests/tools/gpu_classing_ab.py [pods] [variant ...]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import parity
from karpenter_amd import fixtures as fx
from karpenter_amd.scheduling import NewScheduler

pods = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
variants = sys.argv[2:] or ["coop2", "coop1", "coop2", "coop1", "plain"]
full_every = "--digest-once" not in variants
variants = [v for v in variants if not v.startswith("--")]
p = fx.config2(pods=pods, n_types=500, seed=42)
pin = os.path.join(ROOT, "tests", "golden", "fullsize", f"config2_p{pods}_t500_s42.json")
want = json.load(open(pin))["digest"] if os.path.exists(pin) else None
s = NewScheduler(p)
s.Solve(want_results=False)
digests = set()
for v in variants:
    os.environ.pop("KSOLVE_TEST_LDS_PAD", None)
    os.environ.pop("KSOLVE_TEST_ROWS_PER_BLOCK", None)
    if v.startswith("coop2"):                      # "coop2+6000": 6000 bytes of unused LDS per block (occupancy probe); "coop2@64": rows per block
        os.environ.pop("KSOLVE_ROWHASH_KERNEL", None)
        if "+" in v: os.environ["KSOLVE_TEST_LDS_PAD"] = v.split("+")[1].split("@")[0]
        if "@" in v: os.environ["KSOLVE_TEST_ROWS_PER_BLOCK"] = v.split("@")[1]
    else: os.environ["KSOLVE_ROWHASH_KERNEL"] = v
    r = s.Solve(want_results=False, repeat=3)
    rh = sorted(t["row_hash_ms"] for t in r["timings"])
    cl = sorted(t["classify_ms"] for t in r["timings"])
    print(f"{v}: row_hash_ms {rh} classify_ms {[round(x, 3) for x in cl]} -> {pods * 421e-6 / rh[0]:.0f} GB/s best", flush=True)
    if (v != "plain" or pods <= 200000) and (full_every or not digests):
        full = s.Solve()
        d, _ = parity.results_digest(full)
        digests.add(d)
        print("   digest", d[:16], "matches the oracle's pin" if d == want else ("(no pin)" if want is None else "DIFFERS from the pin"), "classes", full["counters"].get("classes"), flush=True)
os.environ.pop("KSOLVE_ROWHASH_KERNEL", None)
os.environ.pop("KSOLVE_TEST_LDS_PAD", None)
os.environ.pop("KSOLVE_TEST_ROWS_PER_BLOCK", None)
s.close()
assert len(digests) == 1, digests
assert want is None or digests == {want}
print("all variants: same digest")
