#!/usr/bin/env python3
"""Population-scale pins of the consolidation sweeps (TEST INFRASTRUCTURE): the ORACLE, offline in the CPU container, simulates and
judges a stratified sample of the probes bench.py's configs[4] legs sweep on the device — single-node probes of the plain cluster
and of the cluster with spread constraints on its bound pods, and multi-node prefixes — and commits ONE sha256 per leg over
(decision, replacement instance types, capacity type, reference bin evaluations) of every sampled probe, with the probes' positions.
bench.py and tests/test_gpu_parity.py compute the same digest from the device's verdicts of the same probes.

  python tests/golden/make_sweep_pins.py single 100000 1000 [threads]          # every 10th of the 10k swept candidates
  python tests/golden/make_sweep_pins.py single-topology 100000 1000 [threads]
  python tests/golden/make_sweep_pins.py multi 100000 320 [threads]            # every 10th of the 3,200 prefixes of 32 windows

Each probe is a fresh oracle Scheduler over the whole cluster minus its candidates (what the reference does per simulation,
helpers.go:53-155): ~25 core-seconds per single-node probe of the 100k-node cluster."""
import hashlib
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

SWEEP_CANDIDATES = 10_000     # bench.py --sweep-candidates
WINDOWS, WINDOW = 32, 101     # bench.py --sweep-windows / --sweep-window-size + 1


def probe_key(decision, replacement, capacity_type, evaluations):
    """One probe's line of the digest. replacement: instance type names (any order) or None."""
    return json.dumps([decision, sorted(replacement) if replacement is not None else None, capacity_type, int(evaluations)], separators=(",", ":"))


def digest_of(keys):
    h = hashlib.sha256()
    for k in keys:
        h.update(k.encode()); h.update(b"\n")
    return h.hexdigest()


def single_positions(n_candidates, sample):
    """Positions (in bench.py's list of swept candidates) of the sampled probes: every k-th, the whole list's mix."""
    step = max(1, n_candidates // sample)
    return list(range(0, n_candidates, step))[:sample]


def multi_positions(n_windows, window, sample):
    """(window, prefix size) of the sampled prefixes: every k-th of the windows' prefixes 2..window, in bench.py's order."""
    keys = [(w, k) for w in range(n_windows) for k in range(2, window + 1)]
    step = max(1, len(keys) // sample)
    return keys[::step][:sample]


def pin_path(leg, nodes):
    return os.path.join(HERE, "sweeps", f"{leg}_n{nodes}_s42.json")


def main():
    leg, nodes, sample = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    threads = int(sys.argv[4]) if len(sys.argv) > 4 else (os.cpu_count() or 1)
    import oracle
    from karpenter_amd import disruption as dz
    topology = leg == "single-topology"
    t0 = time.time()
    cc = dz.make_resident_cluster(n_nodes=nodes, seed=42, topology=topology)
    full = dz.compact_candidates(cc)
    base = dz.compact_problem(cc, pod_groups=[])
    if topology:
        base["clusterPods"] = dz.compact_cluster_pods(cc)
    n_cand = min(SWEEP_CANDIDATES, len(full))
    swept = full[::max(1, len(full) // max(1, n_cand))][:n_cand]
    if leg == "multi":
        n_windows = min(WINDOWS, len(full) // WINDOW)
        where = multi_positions(n_windows, WINDOW, sample)
        sets = [full[w * WINDOW:w * WINDOW + k] for w, k in where]
    else:
        where = single_positions(len(swept), sample)
        sets = [[swept[j]] for j in where]
    probes = []
    for idx in sets:
        pods = [dz.compact_node_pods(cc, i) for i in idx]
        probes.append({"removeNodes": [cc["nodes"][i]["name"] for i in idx], "pods": [p for ps in pods for p in ps]})
    print(f"{leg}: {len(probes)} probes over {nodes} nodes, {threads} threads; cluster built in {time.time() - t0:.0f} s", flush=True)
    t1 = time.time()
    keys, decisions = [], {}
    CH = max(threads, 48)
    for a in range(0, len(probes), CH):       # in chunks: one call's Results documents stay small, progress is visible
        res = oracle.sweep(base, probes[a:a + CH], threads=threads, verdicts=True, multi_node=(leg == "multi"))
        for r in res:
            d, rep, ct = oracle.verdict_key(r["verdict"])
            keys.append(probe_key(d, rep, ct, r["counters"]["binEvaluations"]))
            decisions[d] = decisions.get(d, 0) + 1
        print(f"  {min(a + CH, len(probes))}/{len(probes)} probes, {time.time() - t1:.0f} s", flush=True)
    out = {"leg": leg, "nodes": nodes, "seed": 42, "sample": len(probes), "positions": where, "swept_candidates": n_cand if leg != "multi" else None,
           "windows": [WINDOWS, WINDOW] if leg == "multi" else None, "digest": digest_of(keys), "decisions": decisions, "keys": keys,
           "oracleSeconds": round(time.time() - t1, 1), "oracleThreads": threads,
           "digest_of": "sha256 over one line per sampled probe, in the order of `positions`: JSON [decision, sorted replacement instance types or null, capacity type or null, reference bin evaluations]"}
    os.makedirs(os.path.dirname(pin_path(leg, nodes)), exist_ok=True)
    with open(pin_path(leg, nodes), "w") as f:
        json.dump(out, f)
    print(json.dumps({k: v for k, v in out.items() if k not in ("keys", "positions")}))


if __name__ == "__main__":
    main()
