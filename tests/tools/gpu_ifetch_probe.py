"""Instruction-fetch probe (TEST TOOL, GPU): runs the three pack kernels once each — the general engine on the configs[2] shape
(one wavefront, ~0.5 MB of inlined code), the consolidation sweep (thousands of wavefronts of the same code) and the cursor engine
(a 19 KB loop) — so that `rocprofv3 --pmc SQC_ICACHE_* / SQ_WAIT_* ...` can compare how each is fed with instructions.
usage: rocprofv3 --pmc ... --kernel-trace -- python tests/tools/gpu_ifetch_probe.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from karpenter_amd import fixtures as fx, disruption as dz
from karpenter_amd.scheduling import NewScheduler

t = time.time()
s = NewScheduler(fx.config3(pods=60_000, n_types=500, seed=42)); r = s.Solve(want_results=False); s.close()
print("config3 60k", r["counters"]["engine"], r["timings"][0]["pack_kernel_ms"], file=sys.stderr)
p = fx.config3(pods=4_000, n_types=144, seed=5); s = NewScheduler(p); r = s.Solve(want_results=False); s.close()
print("config3 4k (LDS order)", r["counters"]["engine"], r["timings"][0]["pack_kernel_ms"], file=sys.stderr)
s = NewScheduler(fx.config2(pods=200_000, n_types=500, seed=42)); r = s.Solve(want_results=False); s.close()
print("config2 200k", r["counters"]["engine"], r["timings"][0]["pack_kernel_ms"], file=sys.stderr)
p = fx.config2(pods=100_000, n_types=500, seed=42); p["options"]["engine"] = "general"
s = NewScheduler(p); r = s.Solve(want_results=False); s.close()
print("config2 100k general (lite)", r["counters"]["engine"], r["timings"][0]["pack_kernel_ms"], file=sys.stderr)
cc = dz.make_resident_cluster(n_nodes=100_000, seed=42)
rc = dz.ResidentCluster.from_compact(cc)
order = dz.compact_candidates(cc); order = order[::max(1, len(order) // 10_000)][:10_000]
cmds = rc.decisions([[cc["nodes"][i]] for i in order], arrays=True)
print("sweep 10k", rc.last_sweep["timings"]["pack_us"], file=sys.stderr)
rc.close()
print("total s", time.time() - t, file=sys.stderr)
