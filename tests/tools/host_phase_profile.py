"""Host profile of the engine (TEST TOOL): the test emulation built with -DKSOLVE_PHASE_TIMERS counts host TSC cycles per
phase of the pack loop — where the WORK is (instructions), as opposed to the device's shader-clock counters, which are
dominated by the latency of a lone wavefront. usage: python tests/tools/host_phase_profile.py [pods] [config2|config3]"""
import os as _os
_os.environ.setdefault("KSOLVE_TEST_SOLVER_LIB", "1")   # a test tool: may hand a test build of the solver library to NewScheduler(solver_lib=)
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from karpenter_amd import fixtures as fx
from karpenter_amd.scheduling import NewScheduler

lib = "/tmp/libksolve_emu_timers.so"
subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-pthread", "-DKSOLVE_PHASE_TIMERS", "-o", lib, os.path.join(ROOT, "tests", "emu", "ksolve_emu.cpp")])
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
shape = sys.argv[2] if len(sys.argv) > 2 else "config3"
p = fx.config3(pods=n, n_types=500, seed=42) if shape == "config3" else fx.config2(pods=n, n_types=500, seed=42)
if shape != "config3":
    p["options"] = dict(p.get("options") or {}, engine="general")
s = NewScheduler(p, solver_lib=lib)
t = time.time()
r = s.Solve(want_results=False)
c = r["counters"]
print("pods", n, "claims", c["claims"], "solve_s", round(time.time() - t, 3), "pack_ms", r["timings"][0].get("pack_kernel_ms"), "engine", c["engine"])
names = ["queue", "class_fetch", "sort", "scan", "rec_load", "can_add", "commit", "new_claim", "dead_mark", "try_sched", "total", "ca_pre", "ca_merge", "ca_total",
         "ca_filter", "f_ballots", "f_combine", "s_stage", "s_headroom", "s_select"]
for i, nm in enumerate(names):
    print(f"{nm:12s} {c['phaseCycles'][i] / n:12.0f} host cycles / pod")
