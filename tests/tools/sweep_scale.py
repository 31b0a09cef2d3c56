"""BASELINE configs[4] at size: a resident cluster (disruption.make_resident_cluster) swept by single-node consolidation through
ksolve_sweep, a sample of the probes checked against the oracle's SimulateScheduling of the same candidate (decision, replacement
and the reference-equivalent evaluation count). Lives with the tests because it uses the oracle.
usage: sweep_scale.py NODES CANDIDATES [SAMPLE] [--solver-lib LIB]"""
import os as _os
_os.environ.setdefault("KSOLVE_TEST_SOLVER_LIB", "1")   # a test tool: may hand a test build of the solver library to NewScheduler(solver_lib=)
import argparse, json, os, random, sys, time
from collections import Counter
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle  # noqa: E402
from karpenter_amd import disruption as dz  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("nodes", type=int); ap.add_argument("candidates", type=int); ap.add_argument("sample", type=int, nargs="?", default=32)
ap.add_argument("--solver-lib", default=None); ap.add_argument("--seed", type=int, default=42); ap.add_argument("--repeat", type=int, default=2)
ap.add_argument("--topology", action="store_true", help="two fifths of the default pool's pod templates carry spread constraints")
args = ap.parse_args()
out = {"nodes": args.nodes}
t = time.time(); cc = dz.make_resident_cluster(n_nodes=args.nodes, seed=args.seed, topology=args.topology); out["generate_s"] = time.time() - t
out["bound_pods"] = sum(g["count"] for g in cc["podGroups"])
t = time.time(); rc = dz.ResidentCluster.from_compact(cc, solver_lib=args.solver_lib); out["new_scheduler_s"] = time.time() - t
order = dz.compact_candidates(cc)
order = order[::max(1, len(order) // args.candidates)][:args.candidates]
cands = [[cc["nodes"][i]] for i in order]
for _ in range(args.repeat):
    t = time.time(); cmds = rc.decisions(cands); dt = time.time() - t
tm = rc.last_sweep["timings"]
import hashlib  # noqa: E402
out["verdict_digest"] = hashlib.sha256(json.dumps([[c["decision"], c["replacement"], c.get("replacementCapacityType")] for c in cmds] + [list(rc.last_sweep["referenceBinEvaluations"])],
                                                  sort_keys=True).encode()).hexdigest()[:16]   # every probe's verdict and reference-equivalent evaluation count: equal across builds / kernels
out.update(candidates=len(cands), decisions=dict(Counter(c["decision"] for c in cmds)), python_call_s=dt, timings=tm,
           probes_per_s_kernel=len(cands) / (tm["pack_us"] * 1e-6), probes_per_s_library=len(cands) / ((tm["descriptors_ms"] + tm["sweep_ms"] + tm["verdicts_ms"]) * 1e-3),
           probes_per_s_python=len(cands) / dt)
if args.sample:
    rng = random.Random(1)
    by_dec = {}
    for j, c in enumerate(cmds):
        by_dec.setdefault(c["decision"], []).append(j)
    sample = []
    for d, js in sorted(by_dec.items()):   # every verdict is represented
        sample += rng.sample(js, min(len(js), max(1, args.sample // len(by_dec))))
    base = dz.compact_problem(cc, pod_groups=[])
    if args.topology:
        base["clusterPods"] = dz.compact_cluster_pods(cc)   # the oracle counts domains from them, minus the pods of the probe (excludedPods)
    probes = [{"removeNodes": [cc["nodes"][order[j]]["name"]], "pods": dz.compact_node_pods(cc, order[j])} for j in sample]
    threads = min(len(probes), os.cpu_count() or 1)
    t = time.time(); res = oracle.sweep(base, probes, threads=threads, verdicts=True); out["oracle_s"] = time.time() - t; out["oracle_threads"] = threads
    for j, r, pr in zip(sample, res, probes):
        got = cmds[j]
        assert (got["decision"], got["replacement"], got.get("replacementCapacityType")) == oracle.verdict_key(r["verdict"]), (j, got, r["verdict"])
        assert rc.last_sweep["referenceBinEvaluations"][j] == r["counters"]["binEvaluations"], (j, rc.last_sweep["referenceBinEvaluations"][j], r["counters"]["binEvaluations"])
    out["oracle_checked"] = dict(Counter(cmds[j]["decision"] for j in sample))
    out["oracle_probes_per_s"] = len(sample) / out["oracle_s"]
rc.close()
print(json.dumps(out))
