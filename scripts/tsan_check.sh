#!/bin/bash
# ThreadSanitizer over the paths that use more than one host thread: ksolve_solve_batch (prepass / finish of the handles on a
# thread pool), ksolve_cancel raised while a solve runs, a sweep (host threads for descriptors / verdicts) and ksolve_sweep_replicas. Host flattener + host emulation of the engine, built into /tmp.
set -e
cd "$(dirname "$0")/.."
TSAN=$(ls /usr/lib/gcc/x86_64-linux-gnu/*/libtsan.so | head -1)
STD=/usr/lib/x86_64-linux-gnu/libstdc++.so.6
mkdir -p /tmp/tsan
g++ -O1 -g -fsanitize=thread -fno-omit-frame-pointer -std=c++17 -fPIC -shared -pthread -o /tmp/tsan/libksolve_emu.so tests/emu/ksolve_emu.cpp
g++ -O1 -g -fsanitize=thread -fno-omit-frame-pointer -std=c++17 -fPIC -shared -o /tmp/tsan/libksched.so karpenter_amd/host/ksched.cpp -ldl
cat > /tmp/tsan/run.py <<'PY'
import os, sys, threading, time
os.environ["KSOLVE_TEST_SOLVER_LIB"] = "1"   # solver_lib is a test switch (karpenter_amd/scheduling.py)
sys.path.insert(0, os.getcwd())
import karpenter_amd.scheduling as ks
ks.KSCHED_LIB = "/tmp/tsan/libksched.so"
emu = "/tmp/tsan/libksolve_emu.so"
from karpenter_amd import fixtures as fx
from karpenter_amd.scheduling import NewScheduler, SolveBatch
scheds = [NewScheduler(fx.config2(pods=2000, n_types=100, seed=s), solver_lib=emu) for s in range(24)]
scheds += [NewScheduler(fx.config3(pods=600, n_types=72, seed=s), solver_lib=emu) for s in range(8)]
for _ in range(3):
    rs = SolveBatch(scheds)
print("batch:", sum(r["scheduledPods"] for r in rs), "pods")
out = {}
th = threading.Thread(target=lambda: out.update(r=scheds[0].Solve()))
th.start(); time.sleep(0.01); scheds[0].Cancel(); th.join()
print("cancel: timedOut =", out["r"]["timedOut"])
# a sweep: probe descriptors / validation / verdicts on a few host threads, then the same sweep dealt out over three handles
# ("devices" of the emulation; also what several handles of ONE device do), each share on a host thread of its own
from karpenter_amd import disruption as dz
cc = dz.make_resident_cluster(n_nodes=1500, seed=9)
reps = [NewScheduler(dz.compact_problem(dict(cc, options={"device": d}), pods=[]), solver_lib=emu) for d in range(3)]
order = dz.compact_candidates(cc)[:1200]
cands = [[i] for i in order] + [order[:k] for k in range(2, 20)]
one = reps[0].Sweep(cands, multi_node=True)
many = reps[0].Sweep(cands, multi_node=True, replicas=reps[1:])
print("sweep: equal =", one["decisions"] == many["decisions"], "probes", len(cands), "devices", many["timings"]["devices"])
PY
LD_PRELOAD="$TSAN $STD" TSAN_OPTIONS="halt_on_error=0 report_signal_unsafe=0" python /tmp/tsan/run.py > /tmp/tsan/log 2>&1 || true
tail -2 /tmp/tsan/log
echo "ThreadSanitizer reports: $(grep -c 'WARNING: ThreadSanitizer' /tmp/tsan/log)"
