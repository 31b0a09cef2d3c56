"""Where a consolidation sweep's time goes PER PROBE (TEST TOOL): a resident cluster (disruption.make_resident_cluster) is swept once
for the verdicts, then a sample of the probes of every verdict is solved again one probe handle each through a PROFILING build of
the solver (-DKSOLVE_PHASE_TIMERS: the device library's variant on a GPU box — shader cycles, i.e. latency; the host emulation here —
TSC cycles, i.e. work), whose per-probe phase counters come back with each probe's Results. The sweep kernel's duration is the
duration of its slowest wavefront, so the tail printed here is what a 10k-probe launch waits for.
usage: sweep_probe_costs.py NODES CANDIDATES SAMPLE_PER_VERDICT --solver-lib LIB [--topology]"""
import os as _os
_os.environ.setdefault("KSOLVE_TEST_SOLVER_LIB", "1")   # a test tool: hands a profiling build of the solver library to NewScheduler(solver_lib=)
import argparse, json, os, random, sys, time
from collections import Counter
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from karpenter_amd import disruption as dz  # noqa: E402
from karpenter_amd.scheduling import SolveBatch  # noqa: E402

NAMES = ["queue", "class_fetch", "sort", "scan", "rec_load", "can_add", "commit", "new_claim", "dead_mark", "try_sched", "total", "ca_pre", "ca_merge", "ca_total",
         "ca_filter", "f_ballots", "f_combine", "s_stage", "s_headroom", "s_select", "en_next_block", "en_static_ok", "en_merge", "en_commit"]

ap = argparse.ArgumentParser()
ap.add_argument("nodes", type=int); ap.add_argument("candidates", type=int); ap.add_argument("sample", type=int, nargs="?", default=40)
ap.add_argument("--solver-lib", required=True); ap.add_argument("--seed", type=int, default=42)
ap.add_argument("--topology", action="store_true")
args = ap.parse_args()
cc = dz.make_resident_cluster(n_nodes=args.nodes, seed=args.seed, topology=args.topology)
t = time.time(); rc = dz.ResidentCluster.from_compact(cc, solver_lib=args.solver_lib); t_new = time.time() - t
order = dz.compact_candidates(cc)
order = order[::max(1, len(order) // args.candidates)][:args.candidates]
cands = [[cc["nodes"][i]] for i in order]
cmds = rc.decisions(cands)
out = {"nodes": args.nodes, "candidates": len(cands), "new_scheduler_s": round(t_new, 3), "decisions": dict(Counter(c["decision"] for c in cmds)),
       "sweep_timings": rc.last_sweep["timings"]}
rng = random.Random(1)
by_dec = {}
for j, c in enumerate(cmds):
    by_dec.setdefault(c["decision"], []).append(j)
rows = {}
for d, js in sorted(by_dec.items()):
    sample = rng.sample(js, min(len(js), args.sample))
    probes = []
    for j in sample:
        probes.append(rc.scheduler.Probe([cc["nodes"][order[j]]["name"]], pods_of_removed_nodes=True))
    res = SolveBatch(probes)
    tot, per = [], Counter()
    worst = None
    for j, r in zip(sample, res):
        pc = r["counters"]["phaseCycles"]
        tot.append(pc[10])
        for i, nm in enumerate(NAMES):
            per[nm] += pc[i]
        if worst is None or pc[10] > worst[0]:
            worst = (pc[10], {"pods": r["counters"]["pods"], "claims": r["counters"]["claims"], "relaxations": r["counters"]["relaxations"],
                              "phases": {nm: pc[i] for i, nm in enumerate(NAMES) if pc[i]}})
    for p in probes:
        p.close()
    tot.sort()
    n = len(tot)
    rows[d] = {"probes": n, "cycles_mean": sum(tot) // n, "cycles_median": tot[n // 2], "cycles_p90": tot[(n * 9) // 10 if n > 1 else 0], "cycles_max": tot[-1],
               "phase_mean": {nm: per[nm] // n for nm in NAMES if per[nm]}, "worst": worst[1]}
out["by_decision"] = rows
rc.close()
print(json.dumps(out))
