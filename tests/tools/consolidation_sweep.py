"""BASELINE configs[4]-shaped consolidation sweep on a RESIDENT cluster: the cluster is flattened and
uploaded once (ksolve_create), every candidate is a probe descriptor (ksolve_probe_create: removed-node bitmap +
displaced-pod rows), all probes run in ONE ksolve_solve_batch launch. A sample of the probes is checked against the oracle
on the simulation assembled from scratch (disruption.simulate_scheduling), and a few are also timed on the per-probe
rebuild path this replaces. Usage: consolidation_sweep.py NODES PROBES [SAMPLE] [--solver-lib PATH] [--out FILE]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from karpenter_amd import disruption as dz  # noqa: E402
from karpenter_amd.scheduling import NewScheduler, SolveBatch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("nodes", type=int)
    ap.add_argument("probes", type=int)
    ap.add_argument("sample", type=int, nargs="?", default=8)
    ap.add_argument("--types", type=int, default=500)
    ap.add_argument("--pods-per-node", type=int, default=6)
    ap.add_argument("--squeeze", type=float, default=0.9, help="fraction of nodes with (almost) no cpu left: displaced pods then need replacements")
    ap.add_argument("--solver-lib", default=None)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    import random
    import parity
    import oracle

    t0 = time.perf_counter()
    cluster = dz.make_cluster(n_nodes=args.nodes, pods_per_node=args.pods_per_node, n_types=args.types, seed=7, utilisation=0.8)
    rng = random.Random(7)
    for n in cluster["nodes"]:
        if rng.random() < args.squeeze:
            n["available"] = dict(n["available"], cpu=f"{rng.choice([0, 100, 300])}m")
    ordered = dz.sort_candidates(cluster, cluster["nodes"])
    cands = ordered[:: max(1, len(ordered) // args.probes)][: args.probes]      # from the emptiest to the fullest node
    t_gen = time.perf_counter() - t0

    t0 = time.perf_counter()
    rc = dz.ResidentCluster(cluster, cands, args.solver_lib)
    t_create = time.perf_counter() - t0
    t0 = time.perf_counter()
    probes = [rc._probe([c]) for c in cands]
    t_probe = time.perf_counter() - t0
    t0 = time.perf_counter()
    results = SolveBatch(probes)
    t_solve = time.perf_counter() - t0
    t0 = time.perf_counter()
    results2 = SolveBatch(probes)          # a second sweep on the same probes: nothing is created again
    t_solve2 = time.perf_counter() - t0
    for cs, res in zip(cands, results):
        rc._cache[rc._key([cs])] = dz._finish_simulation(cluster, res, rc._deleting_uids)
    decisions = [dz.compute_consolidation(cluster, [c], rc) for c in cands]
    pack_ms = [r["timings"][0].get("pack_kernel_ms", 0.0) for r in results2]

    # parity on a sample: the simulation assembled from scratch, solved by the oracle
    step = max(1, len(cands) // max(1, args.sample))
    sample = list(range(0, len(cands), step))[: args.sample]
    t0 = time.perf_counter()
    for i in sample:
        want = dz.compute_consolidation(cluster, [cands[i]], oracle.solve)
        assert want["decision"] == decisions[i]["decision"] and want.get("replacement") == decisions[i].get("replacement"), i
        parity.assert_same_results(decisions[i]["results"], want["results"])
        assert decisions[i]["results"]["counters"]["referenceBinEvaluations"] == want["results"]["counters"]["binEvaluations"]
    t_oracle = time.perf_counter() - t0

    # the path this replaces: one problem document, one flatten, one ksolve_create per probe
    t0 = time.perf_counter()
    n_rebuild = min(4, len(cands))
    for c in cands[:n_rebuild]:
        got = dz.compute_consolidation(cluster, [c], lambda p: NewScheduler(p, args.solver_lib).Solve())
        assert got["decision"] == decisions[cands.index(c)]["decision"]
    t_rebuild = (time.perf_counter() - t0) / n_rebuild

    out = {
        "workload": f"single-node consolidation sweep: {args.nodes} existing nodes x {args.pods_per_node} pods/node, {args.types} kwok types, {len(cands)} candidates (probes)",
        "nodes": args.nodes, "bound_pods": sum(len(n["pods"]) for n in cluster["nodes"]), "probes": len(cands), "displaced_pods": sum(len(c["pods"]) for c in cands),
        "seconds": {"cluster_flatten_upload_once": t_create, "probe_descriptors": t_probe, "first_batched_solve": t_solve, "second_batched_solve": t_solve2,
                    "per_probe_rebuild_path_each": t_rebuild, "oracle_sample_each": t_oracle / max(1, len(sample)), "synthetic_cluster_generation": t_gen},
        "probes_per_second": {"resident_sweep_incl_create_and_descriptors": len(cands) / (t_create + t_probe + t_solve), "resident_solve_only": len(cands) / t_solve2,
                              "per_probe_rebuild_path": 1.0 / t_rebuild},
        "pack_kernel_ms": {"max": max(pack_ms), "mean": sum(pack_ms) / len(pack_ms)},
        "decisions": {d: sum(1 for c in decisions if c["decision"] == d) for d in (dz.DELETE, dz.REPLACE, dz.NOOP)},
        "parity": {"sampled_probes_checked_against_oracle": len(sample), "exact": True},
    }
    print(json.dumps(out))
    if args.out:
        with open(args.out, "w") as f:
            json.dump(out, f, indent=1)
    rc.close()


if __name__ == "__main__":
    main()
