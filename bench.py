#!/usr/bin/env python3
"""bench.py — pods scheduled/sec through Solve() on MI355X (BASELINE.json metric).

A "step" is one pass of the hot path — ksolve_solve(): pod classing, queue sort, the pack engine, finalize, results
download over the C ABI — over one batch of synthetic pods whose flattened problem is already resident in HBM
(ksolve_create is outside the timed region, like b.ResetTimer() after setupScheduler in scheduling_benchmark_test.go:166-169;
the JSON re-hydration of Results for Python is outside too: want_results=False, the flat C-ABI Results ARE downloaded).

  N = 1 : BASELINE.json configs[1] — 1M pods, 500 instance types, nodeSelector + taint/toleration constraints.
  N > 1 : weak scaling. Solve() is a serial chain inside one coupled problem, so the path shards across INDEPENDENT
          scheduling problems (NodePool components): rank r solves its own configs[1]-shaped problem
          (seed 42 + r), no collective on the data path; after the timed solves ONE all-reduce (RCCL over xGMI) sums the
          per-instance-type (NodeClaim count, $/h) vector of every rank's packing — the north_star's global packing summary.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md


def source_sha():
    """sha256 over the kernel sources (csrc/, include/): identifies the build a profile was taken on."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "karpenter_amd", "csrc")
    for f in sorted(f for f in os.listdir(d) if os.path.isfile(os.path.join(d, f))) + ["../../include/ksolve.h"]:
        with open(os.path.join(d, f), "rb") as fh:
            h.update(f.encode() + b"\0" + fh.read())
    return h.hexdigest()[:16]


def algorithmic_bytes(c):
    """bytes = P*B_pod + V*B_bin + P*B_bin(writeback) + N_it*T*B_it with the record sizes of the
    layouts actually used (DESIGN.md §Data layout). V = the bins the REFERENCE evaluates."""
    R, RW, IW, K = c["resources"], c["reqWords"], c["itWords"], c["keys"]
    tables = 1 if c.get("strictTableShared") else 2   # PodData.StrictRequirements: its own table only when some pod has a preference
    b_pod = 8 * R + tables * (8 * RW + 16) + 8 + 4 + 8 + 16 + 1
    b_bin = 8 * RW + 16 + 8 * IW + 16 * R + 20 * K + 16
    b_it = 16 * R + 8 * RW + 16 + 8 + 512
    total = c["pods"] * b_pod + c["referenceBinEvaluations"] * b_bin + c["pods"] * b_bin + c["instanceTypes"] * c.get("templates", 1) * b_it
    return total, {"B_pod": b_pod, "B_bin": b_bin, "B_it": b_it}


def launch_type_vector(prob, res):
    """Per-instance-type (NodeClaim count, $/h): every claim is booked on the instance type that gives its cheapest launch
    price — the cheapest available offering its requirements admit (types.go:336-355; ties: catalogue order)."""
    import numpy as np
    its = prob["instanceTypes"]
    names = {t["name"]: i for i, t in enumerate(its)}
    zones = sorted({r["values"][0] for t in its for o in t["offerings"] for r in o["requirements"] if r["key"] == "topology.kubernetes.io/zone"})
    cts = sorted({r["values"][0] for t in its for o in t["offerings"] for r in o["requirements"] if r["key"] == "karpenter.sh/capacity-type"})
    price = np.full((len(its), len(zones), len(cts)), np.inf)
    for i, t in enumerate(its):
        for o in t["offerings"]:
            if not o.get("available", True):
                continue
            rq = {r["key"]: r["values"][0] for r in o["requirements"]}
            z, c = zones.index(rq["topology.kubernetes.io/zone"]), cts.index(rq["karpenter.sh/capacity-type"])
            price[i, z, c] = min(price[i, z, c], o["price"])
    vec = np.zeros((len(its), 2))
    for cl in res["newNodeClaims"]:
        rq = {r["key"]: r for r in cl["requirements"]}

        def allowed(key, universe):
            r = rq.get(key)
            if r is None:
                return np.ones(len(universe), bool)
            return np.array([(v in r["values"]) != r["complement"] for v in universe])
        zm, cm = allowed("topology.kubernetes.io/zone", zones), allowed("karpenter.sh/capacity-type", cts)
        idx = np.array([names[n] for n in cl["instanceTypes"]])
        best = price[idx][:, zm][:, :, cm].reshape(len(idx), -1).min(axis=1)
        k = int(idx[int(np.argmin(best))])
        vec[k, 0] += 1
        vec[k, 1] += float(cl["cheapestPrice"])
    return vec


def config4_sweep(args, device_index, rank, world, topology=False):
    """BASELINE configs[4]: the consolidation replay (topology=True: the same cluster with spread constraints on two fifths of the
    default pool's pod templates — every probe then takes its candidates' share out of the cluster-wide domain counts, and the
    oracle re-simulates the sampled probes over all 2M bound pods). A resident cluster (disruption.make_resident_cluster: nodes packed with the
    benchmark's pods, then scaled down — test/suites/performance/basic_test.go:61-68) is uploaded once; single-node consolidation
    (singlenodeconsolidation.go:55-126) then simulates every candidate — SimulateScheduling (disruption/helpers.go:53-155) +
    computeConsolidation (consolidation.go:159-256) — as ONE ksolve_sweep launch, one wavefront per candidate. With N ranks the
    candidates are dealt out round-robin (the cluster tables are replicated, no collective on the data path); the verdict counts
    are summed afterwards. A sample of the probes is re-simulated by the oracle: the check, and this leg's CPU baseline."""
    from collections import Counter
    from karpenter_amd import disruption as dz
    out = {"workload": f"BASELINE configs[4]: single-node consolidation sweep over a resident cluster of {args.sweep_nodes} existing nodes" + (", bound pods with zonal / hostname spread constraints" if topology else "")}
    t = time.perf_counter(); cc = dz.make_resident_cluster(n_nodes=args.sweep_nodes, seed=42, topology=topology); out["generate_s"] = time.perf_counter() - t
    n_sample = args.sweep_topology_sample if topology else args.sweep_sample
    cc["options"] = {"device": device_index}
    out["nodes"], out["bound_pods"] = args.sweep_nodes, sum(g["count"] for g in cc["podGroups"])
    t = time.perf_counter(); rc = dz.ResidentCluster.from_compact(cc, solver_lib=args.solver_lib); out["new_scheduler_s"] = time.perf_counter() - t
    order = dz.compact_candidates(cc)
    order = order[::max(1, len(order) // max(1, args.sweep_candidates))][:args.sweep_candidates]   # every k-th of sortCandidates' order: the whole list's mix
    mine = order[rank::world]
    cands = [[cc["nodes"][i]] for i in mine]
    rc.decisions(cands[:64])                       # warm-up: the arena, the per-cluster tables (ksolve_node_dead0)
    t = time.perf_counter(); cmds = rc.decisions(cands); dt = time.perf_counter() - t
    tm = rc.last_sweep["timings"]
    first_call_s = (tm["descriptors_ms"] + tm["sweep_ms"] + tm["verdicts_ms"]) * 1e-3
    # the sweep once more: the first call of this size also grows the arena the base handle keeps between sweeps (a device
    # allocation of up to 4 GB); a disruption controller sweeps every pass and pays that once
    t = time.perf_counter(); cmds2 = rc.decisions(cands); dt = min(dt, time.perf_counter() - t)
    if [(c["decision"], c["replacement"]) for c in cmds2] != [(c["decision"], c["replacement"]) for c in cmds]:
        raise SystemExit("bench.py: two sweeps of the same candidates disagree")
    tm2 = rc.last_sweep["timings"]
    if (tm2["descriptors_ms"] + tm2["sweep_ms"] + tm2["verdicts_ms"]) * 1e-3 <= first_call_s:
        tm = tm2
    json_form_s = (tm["descriptors_ms"] + tm["sweep_ms"] + tm["verdicts_ms"]) * 1e-3
    # the same sweep through the binary form of the call (ksched_sweep_arrays: candidates as a CSR of node positions, verdicts as
    # arrays; prices and capacity types from the library's node table) — what a cgo caller uses, and what `value` is quoted on
    for _ in range(2):
        t = time.perf_counter(); cmds3 = rc.decisions(cands, library_prices=True, arrays=True); dt3 = time.perf_counter() - t
        if [(c["decision"], c["replacement"], c.get("replacementCapacityType")) for c in cmds3] != [(c["decision"], c["replacement"], c.get("replacementCapacityType")) for c in cmds]:
            raise SystemExit("bench.py: the binary form of the sweep disagrees with the JSON form")
        tm3 = rc.last_sweep["timings"]
        if (tm3["descriptors_ms"] + tm3["sweep_ms"] + tm3["verdicts_ms"]) * 1e-3 < (tm["descriptors_ms"] + tm["sweep_ms"] + tm["verdicts_ms"]) * 1e-3:
            tm, dt = tm3, min(dt, dt3)
    lib_s = (tm["descriptors_ms"] + tm["sweep_ms"] + tm["verdicts_ms"]) * 1e-3
    out["first_call_s"] = first_call_s
    out["library_call_json_form_s"] = json_form_s
    out.update(candidates=len(cands), decisions=dict(Counter(c["decision"] for c in cmds)), displaced_pods=tm["pods"],
               seconds={"descriptors": tm["descriptors_ms"] * 1e-3, "upload": tm["upload_us"] * 1e-6, "pack_kernel": tm["pack_us"] * 1e-6, "finalize": tm["finalize_us"] * 1e-6,
                        "download": tm["download_us"] * 1e-6, "verdicts": tm["verdicts_ms"] * 1e-3, "library_call": lib_s, "python_call": dt},
               value=len(cands) / lib_s, unit="probes/s",
               probes_per_s={"pack_kernel_only": len(cands) / (tm["pack_us"] * 1e-6), "ksolve_sweep_call": len(cands) / (tm["sweep_ms"] * 1e-3),
                             "with_descriptors_and_verdicts": len(cands) / lib_s, "through_python": len(cands) / dt},
               timed_region="ksched_sweep_arrays() (the faster of it and ksched_sweep(), the JSON form: library_call_json_form_s): probe descriptors (host library), ksolve_sweep (upload, one launch, finalize, download), verdicts")
    out["kernels"] = sweep_rooflines(tm, len(cands))
    if args.sweep_contexts > 1 and not topology:
        # Several contexts of the same cluster on THIS device (ksolve_sweep_replicas over handles that share a device): each handle
        # has its own stream, page-locked staging and arena, its share of the probes runs on a host thread of its own — one share's
        # descriptors / upload / finalize / download go beside another share's kernel. What a controller that keeps K handles per
        # GPU gets from the same library call; the cluster tables are replicated K times (0.6 GB each at 100k nodes).
        ctx = {}
        extra = []
        try:
            for k in sorted({2, args.sweep_contexts}):
                while len(extra) < k - 1:
                    extra.append(dz.ResidentCluster.from_compact(cc, solver_lib=args.solver_lib))
                best = None
                for _ in range(4):
                    cm = rc.decisions(cands, library_prices=True, arrays=True, replicas=extra[:k - 1])
                    if [(c["decision"], c["replacement"], c.get("replacementCapacityType")) for c in cm] != [(c["decision"], c["replacement"], c.get("replacementCapacityType")) for c in cmds]:
                        raise SystemExit("bench.py: the sweep over several contexts of one device disagrees with the sweep over one")
                    t_ = rc.last_sweep["timings"]
                    sec = (t_["descriptors_ms"] + t_["sweep_ms"] + t_["verdicts_ms"]) * 1e-3
                    if best is None or sec < best[0]:
                        best = (sec, t_)
                ctx[str(k)] = {"library_call": best[0], "probes_per_s": len(cands) / best[0], "descriptors": best[1]["descriptors_ms"] * 1e-3, "ksolve_sweep_replicas": best[1]["sweep_ms"] * 1e-3,
                               "verdicts": best[1]["verdicts_ms"] * 1e-3, "slowest_share": {"upload": best[1]["upload_us"] * 1e-6, "pack_kernel": best[1]["pack_us"] * 1e-6, "finalize": best[1]["finalize_us"] * 1e-6, "download": best[1]["download_us"] * 1e-6}}
        finally:
            for e in extra:
                e.close()
        out["contexts_on_one_device"] = dict(ctx, note="ksched_sweep_arrays over K handles of the same cluster on one GPU (ksolve_sweep_replicas: contiguous shares of about equal displaced pods, one host thread and one stream per handle); verdicts equal to the one-handle sweep's; `value` of this leg stays the one-handle call")
    if world == 1 and not args.no_parity_pin:
        # population-scale pin: the oracle's verdicts of a stratified 1,000 of these probes, made offline (tests/golden/make_sweep_pins.py)
        out["oracle_pin"] = sweep_pin_check("single-topology" if topology else "single", args.sweep_nodes, len(cands),
                                            lambda j: (cmds[j]["decision"], cmds[j]["replacement"], cmds[j].get("replacementCapacityType"), rc.last_sweep["referenceBinEvaluations"][j]))
    if n_sample > 0 and rank == 0:
        import random
        import oracle   # the checker: re-simulates sampled probes; its rate is this leg's CPU baseline
        rng = random.Random(1)
        by_dec = {}
        for j, c in enumerate(cmds):
            by_dec.setdefault(c["decision"], []).append(j)
        sample = []
        for _, js in sorted(by_dec.items()):       # every verdict is represented
            sample += rng.sample(js, min(len(js), max(1, n_sample // len(by_dec))))
        base = dz.compact_problem(cc, pod_groups=[])
        if topology:
            base["clusterPods"] = dz.compact_cluster_pods(cc)   # countDomains reads every bound pod (topology.go:361-459), minus the probe's own
        probes = [{"removeNodes": [cc["nodes"][mine[j]]["name"]], "pods": dz.compact_node_pods(cc, mine[j])} for j in sample]
        threads = min(len(probes), os.cpu_count() or 1)
        t = time.perf_counter(); res = oracle.sweep(base, probes, threads=threads, verdicts=True); osec = time.perf_counter() - t
        bad = []
        for j, r, pr in zip(sample, res, probes):
            got = cmds[j]      # judged by the oracle's own computeConsolidation (oracle/consolidation.hpp), not by karpenter_amd.disruption
            if (got["decision"], got["replacement"], got.get("replacementCapacityType")) != oracle.verdict_key(r["verdict"]) \
                    or rc.last_sweep["referenceBinEvaluations"][j] != r["counters"]["binEvaluations"]:
                bad.append(cc["nodes"][mine[j]]["name"])
        if bad:
            raise SystemExit(f"bench.py: configs[4] probes differ from the oracle's simulation: {bad[:5]}")
        out["oracle_check"] = {"probes": len(sample), "by_decision": dict(Counter(cmds[j]["decision"] for j in sample)), "all_identical": True,
                               "compared": "decision, replacement instance types, capacity type (all three from oracle/consolidation.hpp's restatement of computeConsolidation on the oracle's own Results), reference-equivalent evaluation count"}
        out["cpu_baseline"] = {"value": len(sample) / osec, "unit": "probes/s", "cores": threads, "kind": "port",
                               "sample": f"{len(sample)} of the swept probes, each a fresh oracle Scheduler over the {args.sweep_nodes}-node cluster (what the reference does per simulation), {threads} at a time", "seconds": osec}
    if args.sweep_windows > 0 and not topology:
        out["multi_node"] = config4_multi_node(args, cc, rc, rank, world)
    rc.close()
    return out


def sweep_pin_check(leg, nodes, n_swept, verdict_at):
    """The device's verdicts of the probes a committed sweep pin samples (tests/golden/sweeps/<leg>_n<nodes>_s42.json: the oracle's
    simulation AND decision of ~1,000 stratified probes, one sha256) against that pin. verdict_at(position) -> (decision,
    replacement instance types, capacity type, reference bin evaluations) of the probe at `position` of this sweep. None when no pin
    matches this sweep's shape; a mismatch stops the run."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import make_sweep_pins as msp
    path = msp.pin_path(leg, nodes)
    if not os.path.exists(path):
        return None
    with open(path) as f:
        g = json.load(f)
    if leg != "multi" and g.get("swept_candidates") != n_swept:
        return None
    keys = []
    for pos in g["positions"]:
        d, rep, ct, ev = verdict_at(tuple(pos) if isinstance(pos, list) else pos)
        keys.append(msp.probe_key(d, rep, ct, ev))
    ok = msp.digest_of(keys) == g["digest"]
    if not ok:
        bad = [pos for pos, a, b in zip(g["positions"], keys, g["keys"]) if a != b]
        raise SystemExit(f"bench.py: the configs[4] {leg} sweep differs from the oracle's pin at {len(bad)} of {len(keys)} probes (first: {bad[:3]})")
    return {"pin": os.path.relpath(path, ROOT), "probes": len(keys), "by_decision": g["decisions"], "digest_matches_oracle": True,
            "oracle_seconds_offline": g["oracleSeconds"], "oracle_threads": g["oracleThreads"],
            "compared": "per probe: decision, replacement instance types, capacity type, reference bin evaluations (the oracle's own simulation and computeConsolidation); one sha256 over the sample"}


def sweep_rooflines(tm, n_probes):
    """The two kernels of the consolidation path that fill the chip, with the bytes their algorithm has to move:
    ksolve_pack_sweep — per displaced pod its class record and outputs, per 4096-node step of an existing-node scan 512 B of
    the class's rejection row, per node / NodeClaim actually evaluated its record; ksolve_node_dead0 — the node tables once,
    one bit per (class, node) out. `traffic` = measured HBM bytes of the same kernels (profiles/round6/pmc_traffic.json, when
    it was taken on this build)."""
    rw, nr, iw = tm.get("req_words", 0), tm.get("resources", 0), tm.get("it_words", 0)
    if not rw:
        return None
    b_cls = 8 * rw + 16 + 8 * nr + 8              # class record: requirement masks, flag words, requests, toleration mask
    b_node = 8 * rw + 8 + 8 * nr + 8 + 4          # pristine node: masks, defined / complement, remaining, taints, pod count
    b_claim = 8 * rw + 16 + 8 * iw + 16 * nr + 16   # NodeClaim hot record: masks, flags, instance types, total + headroom, meta
    b_out = 4 + 4 + 1 + 1 + 4                     # per pod: queue entry, assignment, error, diag, slot
    claim_evals = max(0, tm["bin_evaluations"] - tm["node_evaluations"])
    alg = tm["pods"] * (b_cls + b_out) + tm["node_block_steps"] * 512 + tm["node_evaluations"] * b_node + claim_evals * 2 * b_claim
    pmc = {}
    try:
        with open(os.path.join(ROOT, "profiles", "round6", "pmc_traffic.json")) as f:
            doc = json.load(f)
        if doc.get("source_sha") == source_sha():
            pmc = doc.get("sweep_kernels", {})
    except (OSError, ValueError):
        pass
    ps = tm["pack_us"] * 1e-6
    # the compact form of the launch (ksolve_pack_sweep4: four wavefronts per workgroup sharing the read-only tables, eight probes per CU)
    # runs whenever the cluster's dictionaries fit its working set (<= 32 requirement words, <= 512 instance types): every KWOK cluster
    compact = rw <= 32 and iw <= 8
    kname = "ksolve_pack_sweep4" if compact and "ksolve_pack_sweep4" in pmc else "ksolve_pack_sweep"
    if compact and kname != "ksolve_pack_sweep4" and "ksolve_pack_sweep" in pmc:
        pmc = dict(pmc, ksolve_pack_sweep={})   # counters of the one-wavefront kernel do not describe this launch
    pmc = dict(pmc, ksolve_pack_sweep=pmc.get(kname, {}))
    out = {"ksolve_pack_sweep": {"kernel": "ksolve_pack_sweep4" if compact else "ksolve_pack_sweep",
                                 "bound": "latency per probe (one wavefront each), probes in flight per CU (8 compact / 4 general: registers and LDS)",
                                 "grid": "workgroups of four wavefronts, as many as the chip holds at once (2 per CU), each wavefront striding over the probes" if compact else f"{min(n_probes, 8192)} blocks of one wavefront", "probes": n_probes,
                                 "algorithmic_bytes": alg, "avg_kernel_ms": ps * 1e3, "achieved": alg / ps / 1e9 if ps > 0 else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": alg / ps / 1e9 / HBM_PEAK_GBS if ps > 0 else None, "traffic": pmc.get("ksolve_pack_sweep", {}).get("traffic_bytes_largest_launch"),
                                 "waves": pmc.get("ksolve_pack_sweep", {}).get("SQ_WAVES", {}).get("largest_launch"),
                                 "terms": {"displaced_pods": tm["pods"], "class_record_bytes": b_cls, "node_block_steps": tm["node_block_steps"], "nodes_evaluated": tm["node_evaluations"], "node_record_bytes": b_node,
                                           "claims_evaluated": claim_evals, "claim_record_bytes": b_claim}}}
    ds = tm.get("node_dead0_us", 0.0) * 1e-6
    if ds > 0:
        alg0 = tm["nodes"] * b_node + tm["classes"] * b_cls + tm["classes"] * ((tm["nodes"] + 63) // 64) * 8
        out["ksolve_node_dead0"] = {"bound": "compare throughput (classes x nodes tests): a lane holds its node's scalars in registers across the 32 classes of its wavefront's chunk; the node tables are read once per chunk, from L2 after the first",
                                    "grid": f"{(tm['nodes'] + 63) // 64} x {(tm['classes'] + 31) // 32} blocks of one wavefront (64 nodes x 32 classes each)",
                                    "class_node_tests": tm["classes"] * tm["nodes"], "tests_per_s": tm["classes"] * tm["nodes"] / ds,
                                    "algorithmic_bytes": alg0, "avg_kernel_ms": ds * 1e3, "achieved": alg0 / ds / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg0 / ds / 1e9 / HBM_PEAK_GBS,
                                    "traffic": pmc.get("ksolve_node_dead0", {}).get("traffic_bytes_largest_launch")}
    return out


def config4_multi_node(args, cc, rc, rank, world):
    """The multi-node half of the replay: MultiNodeConsolidation.ComputeCommands (multinodeconsolidation.go:51-111) takes the first
    100 candidates of sortCandidates' order and binary-searches the longest prefix that consolidates (firstNConsolidationOption,
    :117-207: about seven dependent simulations). Here every prefix the search can reach — sizes 2..101 of a window of 101
    candidates — is a probe, the windows are consecutive slices of the sorted candidate list (one per pass of a replay that has
    consolidated the earlier ones away), all windows' prefixes are ONE ksolve_sweep (several launches when their workspaces pass
    the arena budget), verdicts incl. filterOutSameInstanceType (:209-246) from the host library, and the binary search is a walk
    over the verdicts. With N ranks the windows are dealt out round-robin. A few prefixes are re-simulated by the oracle."""
    from collections import Counter
    from karpenter_amd import disruption as dz
    K = args.sweep_window_size + 1
    full = dz.compact_candidates(cc)                   # sortCandidates (consolidation.go:149-154)
    n_windows = min(args.sweep_windows, len(full) // K)
    mine = list(range(n_windows))[rank::world]
    sets, key = [], []
    for w in mine:
        win = [cc["nodes"][i] for i in full[w * K:(w + 1) * K]]
        for k in range(2, K + 1):
            sets.append(win[:k]); key.append((w, k))
    out = {"workload": f"multi-node consolidation: {n_windows} windows of {K} candidates (sortCandidates' order), every prefix of 2..{K} nodes simulated: {(K - 1) * n_windows} probes",
           "windows": n_windows, "window_candidates": K, "probes": len(sets)}
    if not sets:
        return out
    # timed in steady state like the single-node sweep: the first call also grows the handle's device arena (GBs for 3,200 prefixes of
    # up to 2,000 pods: an allocation of 3-30 ms by box) and its page-locked staging memory; a controller sweeps every few seconds
    t = time.perf_counter(); rc.decisions(sets, multi_node=True, library_prices=True); first_s = time.perf_counter() - t
    t = time.perf_counter(); cmds = rc.decisions(sets, multi_node=True, library_prices=True); dt = time.perf_counter() - t
    tm = rc.last_sweep["timings"]
    json_form_s = (tm["descriptors_ms"] + tm["sweep_ms"] + tm["verdicts_ms"]) * 1e-3
    # the binary form of the call, as in the single-node leg (ksched_sweep_arrays: the prefixes as a CSR of node positions — what a cgo
    # caller hands over; the JSON form spends most of its descriptor phase parsing 160k node positions): same commands, the faster is quoted
    for _ in range(2):
        t = time.perf_counter(); cmds3 = rc.decisions(sets, multi_node=True, library_prices=True, arrays=True); dt3 = time.perf_counter() - t
        if [(c["decision"], c["replacement"], c.get("replacementCapacityType")) for c in cmds3] != [(c["decision"], c["replacement"], c.get("replacementCapacityType")) for c in cmds]:
            raise SystemExit("bench.py: the binary form of the multi-node sweep disagrees with the JSON form")
        tm3 = rc.last_sweep["timings"]
        if (tm3["descriptors_ms"] + tm3["sweep_ms"] + tm3["verdicts_ms"]) < (tm["descriptors_ms"] + tm["sweep_ms"] + tm["verdicts_ms"]):
            tm, dt = tm3, min(dt, dt3)
    lib_s = (tm["descriptors_ms"] + tm["sweep_ms"] + tm["verdicts_ms"]) * 1e-3
    out["first_call_python_s"] = first_s
    out["library_call_json_form_s"] = json_form_s
    by = dict(zip(key, cmds))
    chosen = []
    for w in mine:
        cmd, probes = dz.first_n_from_commands(K, lambda k: by[(w, k)], args.sweep_window_size)
        chosen.append((len(cmd.get("candidates") or []), cmd["decision"], len(probes)))
    out.update(displaced_pods=tm["pods"], decisions_of_all_prefixes=dict(Counter(c["decision"] for c in cmds)),
               commands={"decisions": dict(Counter(d for _, d, _ in chosen)), "nodes_consolidated": sum(n for n, _, _ in chosen),
                         "largest_prefix": max(n for n, _, _ in chosen), "binary_search_steps_replaced_per_window": max(p for _, _, p in chosen)},
               seconds={"descriptors": tm["descriptors_ms"] * 1e-3, "upload": tm["upload_us"] * 1e-6, "pack_kernel": tm["pack_us"] * 1e-6, "finalize": tm["finalize_us"] * 1e-6,
                        "download": tm["download_us"] * 1e-6, "verdicts": tm["verdicts_ms"] * 1e-3, "library_call": lib_s, "python_call": dt},
               value=len(sets) / lib_s, unit="probes/s", pods_placed_per_s=tm["pods"] / lib_s, windows_per_s=len(mine) / lib_s)
    if world == 1 and not args.no_parity_pin and n_windows == 32 and K == 101:
        pos_of = {kk: j for j, kk in enumerate(key)}
        out["oracle_pin"] = sweep_pin_check("multi", args.sweep_nodes, None,
                                            lambda wk: (by[wk]["decision"], by[wk]["replacement"], by[wk].get("replacementCapacityType"), rc.last_sweep["referenceBinEvaluations"][pos_of[wk]]))
    if args.sweep_sample > 0 and rank == 0:
        import oracle
        picks = sorted({(mine[0], 2), (mine[0], 7), (mine[len(mine) // 2], 23), (mine[-1], min(K, 60)), (mine[-1], K)} & set(key))
        base = dz.compact_problem(cc, pod_groups=[])
        probes, cand_sets = [], []
        for w, k in picks:
            idx = full[w * K:w * K + k]
            pods = [dz.compact_node_pods(cc, i) for i in idx]
            probes.append({"removeNodes": [cc["nodes"][i]["name"] for i in idx], "pods": [p for ps in pods for p in ps]})
            cand_sets.append([dict(cc["nodes"][i], pods=ps) for i, ps in zip(idx, pods)])
        threads = min(len(probes), os.cpu_count() or 1)
        t = time.perf_counter(); res = oracle.sweep(base, probes, threads=threads, verdicts=True, multi_node=True); osec = time.perf_counter() - t
        pos = {kk: j for j, kk in enumerate(key)}
        for kk, r, cs in zip(picks, res, cand_sets):
            got = by[kk]
            if (got["decision"], got["replacement"], got.get("replacementCapacityType")) != oracle.verdict_key(r["verdict"]) or rc.last_sweep["referenceBinEvaluations"][pos[kk]] != r["counters"]["binEvaluations"]:
                raise SystemExit(f"bench.py: configs[4] multi-node probe (window {kk[0]}, {kk[1]} nodes) differs from the oracle's simulation")
        out["oracle_check"] = {"probes": len(picks), "prefix_sizes": [k for _, k in picks], "all_identical": True, "compared": "decision, replacement instance types after filterOutSameInstanceType, capacity type (the oracle's own restatement of the multi-node step), reference-equivalent evaluation count"}
        out["cpu_baseline"] = {"value": len(picks) / osec, "unit": "probes/s", "cores": threads, "kind": "port", "seconds": osec,
                               "sample": f"{len(picks)} of the swept prefixes ({sum(len(p['pods']) for p in probes)} displaced pods), each a fresh oracle Scheduler over the cluster, {threads} at a time"}
    return out


def packing_vector(res, n_its):
    """The same vector from the solver library (ksolve_packing_vector through the Results document): [type, count, $/h] triples."""
    import numpy as np
    vec = np.zeros((n_its, 2))
    for i, c, d in res["packingVector"]:
        vec[i] = (c, d)
    return vec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--pods", type=int, default=1_000_000, help="pods per GPU (configs[1] = 1M)")
    ap.add_argument("--types", type=int, default=500)
    ap.add_argument("--cpu-sample", type=int, default=120_000, help="pods in the bounded cpu_baseline sample (about 10-30 s of one core with the round-4 oracle)")
    ap.add_argument("--cpu-runs", type=int, default=3, help="oracle runs of the sample (median reported)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-engine-baseline", action="store_true", help="skip timing the engine's own source compiled for one host core")
    ap.add_argument("--topology-pods", type=int, default=1_000_000, help="BASELINE configs[2] shape (anti-affinity + 3-zone spread) reported beside the headline, 0 = skip")
    ap.add_argument("--batch-problems", type=int, default=512, help="independent problems solved with ONE batched launch (reported beside the headline, 0 = skip)")
    ap.add_argument("--batch-pods", type=int, default=20_000, help="pods per problem of the batched measurement")
    ap.add_argument("--components-pods", type=int, default=10_000_000, help="BASELINE configs[3]: pods of the 16-NodePool batch solved as NodePool components (one block each, one launch), 0 = skip")
    ap.add_argument("--components-types", type=int, default=1000)
    ap.add_argument("--beyond-lds-pods", type=int, default=2_000_000, help="the configs[1] mix at a size whose NodeClaims no longer fit the cursor engine's LDS plan (~3,000): "
                    "the engine moves its claim state to HBM (round 4) instead of handing the batch to the general engine; digest-checked against the oracle's pin of that size when committed; 0 = skip")
    ap.add_argument("--whole-batch-exact-pods", type=int, default=10_000_000, help="BASELINE configs[3] at its own size as ONE exact Solve() (27,345 in-flight NodeClaims: the cursor engine's HBM plan), ~40 s; 0 = skip")
    ap.add_argument("--whole-batch-pods", type=int, default=1_000_000, help="BASELINE configs[3] as ONE exact Solve() of the whole batch (all 16 NodePools in one claim order), "
                    "digest-checked against the oracle's pin of that size when one is committed; 0 = skip. 10M pods take the general engine ~700 s (tests/tools/whole_batch_c3.py)")
    ap.add_argument("--components-calibration-pods", type=int, default=200_000, help="size at which the component split is compared with ONE Solve() of the whole batch (L2-canonical deltas)")
    ap.add_argument("--sweep-nodes", type=int, default=100_000, help="BASELINE configs[4]: existing nodes of the resident cluster swept by single-node consolidation (about 20 bound pods each), 0 = skip")
    ap.add_argument("--sweep-candidates", type=int, default=10_000, help="candidates (probes) per launch of the sweep")
    ap.add_argument("--sweep-contexts", type=int, default=0, help="configs[4]: also sweep through K handles of the same cluster on one device (2 and K are measured), <= 1 = skip")
    ap.add_argument("--sweep-windows", type=int, default=32, help="multi-node consolidation: windows of the sorted candidate list whose prefixes are all simulated in one sweep, 0 = skip")
    ap.add_argument("--sweep-window-size", type=int, default=100, help="MultiNodeConsolidation's batch (multinodeconsolidation.go:80): the search covers prefixes of up to this many + 1 candidates")
    ap.add_argument("--sweep-sample", type=int, default=32, help="probes of the sweep re-simulated by the oracle (checker + CPU baseline of this leg)")
    ap.add_argument("--sweep-topology-sample", type=int, default=12, help="the second configs[4] leg — the same cluster with spread constraints on its bound pods — re-simulates this many probes by the oracle; 0 = skip the leg")
    ap.add_argument("--no-parity-pin", action="store_true", help="skip the digest check of the timed problem against the committed oracle pin")
    ap.add_argument("--engine", default="auto", choices=["auto", "general", "cursor"], help="pack engine (auto: the cursor engine for purely positive batches)")
    ap.add_argument("--solver-lib", default=None, help="TEST HOOK (tests/test_bench_contract.py, needs KSOLVE_BENCH_TEST_HOOK=1): a host build of the engine behind the same "
                    "C ABI, so that the launcher / collective / JSON contract can be exercised without a GPU; never a measurement, never a default")
    args = ap.parse_args()
    if args.solver_lib and os.environ.get("KSOLVE_BENCH_TEST_HOOK") != "1":
        raise SystemExit("bench.py: --solver-lib is a test hook (set KSOLVE_BENCH_TEST_HOOK=1); the product has no CPU path")
    if args.solver_lib:
        os.environ["KSOLVE_TEST_SOLVER_LIB"] = "1"      # the same gate on NewScheduler(solver_lib=) (karpenter_amd/scheduling.py)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    torch = None
    device_index = local_rank
    reduce_device = "cpu"
    if world > 1:
        import torch
        import torch.distributed as dist
        ngpu = torch.cuda.device_count()
        if ngpu == 0:
            if not args.solver_lib:
                raise RuntimeError("bench.py: no GPU visible to torch (the product has no CPU path)")
            dist.init_process_group(backend="gloo")     # contract test without a GPU: see --solver-lib
        elif ngpu >= world:
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))  # nccl == RCCL on ROCm
            reduce_device = "cuda"
        else:
            # fewer GPUs than ranks (smoke-testing the launcher on a 1-GPU box): ranks share GPUs, collectives on gloo
            device_index = local_rank % max(1, ngpu)
            torch.cuda.set_device(device_index)
            dist.init_process_group(backend="gloo")

    import __graft_entry__
    if rank == 0:
        __graft_entry__.build()
    if dist is not None:
        dist.barrier()
    from karpenter_amd import fixtures as fx
    from karpenter_amd.scheduling import NewScheduler

    seed = 42 + rank
    prob = fx.config2(pods=args.pods, n_types=args.types, seed=seed)
    prob["options"]["device"] = device_index
    if args.engine != "auto":
        prob["options"]["engine"] = args.engine
    # the process's first NewScheduler also loads the libraries and initialises the HIP runtime (~0.1 s): a controller pays that at
    # start-up, not per provisioning pass, so it is taken out of new_scheduler_s by a tiny problem first
    t_first = time.perf_counter()
    NewScheduler(dict(fx.config1(pods=64, n_types=8, seed=1), options={"device": device_index}), solver_lib=args.solver_lib).close()
    t_first = time.perf_counter() - t_first
    t_open = time.perf_counter()
    sched = NewScheduler(prob, solver_lib=args.solver_lib)  # flatten + upload: inputs resident in HBM before the timed region
    t_open = time.perf_counter() - t_open

    def sync():
        if torch is not None and torch.cuda.is_available():
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        sched.Solve(want_results=False)
    if dist is not None:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    last = None
    timings = []
    for _ in range(args.steps):
        last = sched.Solve(want_results=False)  # synchronous: returns after the device finished and the flat Results are on the host
        timings += last["timings"]
    sync()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    scheduled = last["scheduledPods"]
    cost = last["packingCost"]
    claims = last["counters"]["claims"]

    # ---- untimed: the packing itself (digest against the oracle's pin, per-instance-type summary) ----
    t_full = time.perf_counter()
    full = sched.Solve(want_results=True)
    t_full = time.perf_counter() - t_full
    # ... and re-hydrated the way a caller that keeps its pods by position does it (the cgo shim: go/ksolve_rehydrate.go walks
    # pod_assignment / pod_slot; here Scheduler.PodsByClaim): the NodeClaims as objects + two flat arrays, no uid text
    t_pos = time.perf_counter()
    by_pos = sched.Solve(want_results="claims-compact")
    pods_of = sched.PodsByClaim(len(by_pos["newNodeClaims"]))
    t_pos = time.perf_counter() - t_pos
    if [len(x) for x in pods_of] != [len(c["pods"]) for c in full["newNodeClaims"]]:
        raise SystemExit("bench.py: re-hydration by position disagrees with the Results document")
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import parity   # canonical Results digest (shared with the tests; does not touch oracle/)
    digest, _ = parity.results_digest(full)
    vec = packing_vector(full, len(prob["instanceTypes"]))   # ksolve_packing_vector; tests/test_distributed_gloo.py checks it against launch_type_vector on the oracle's claims

    if dist is not None:
        # max over ranks of the timed region; the global packing summary = the per-instance-type (count, $/h) vectors of all
        # ranks summed with one all-reduce (RCCL over xGMI when every rank has its GPU)
        t = torch.tensor([elapsed], dtype=torch.float64, device=reduce_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        v = torch.tensor(vec.reshape(-1).tolist() + [float(scheduled)], dtype=torch.float64, device=reduce_device)
        dist.all_reduce(v, op=dist.ReduceOp.SUM)
        scheduled = int(round(v[-1].item()))
        import numpy as np
        vec = np.array(v[:-1].tolist()).reshape(-1, 2)
        claims, cost = int(round(vec[:, 0].sum())), float(vec[:, 1].sum())

    # ---- BASELINE configs[3] (all ranks take part): 10M pods x 1k instance types x 16 NodePools. Every pod pins its NodePool, so the
    # batch falls into 16 components (karpenter_amd/components.py) that cannot share a claim; each is solved EXACTLY as its own
    # problem. One GPU: all components in one launch (one wavefront each). N GPUs: the components are dealt over the ranks by pod count (LPT, in the host library), every rank runs
    # its components in one launch, and the per-instance-type (NodeClaim count, $/h) vectors are summed with ONE all-reduce — the
    # north_star's global packing summary. The union is a packing of equal quality, not the reference's pod-for-pod answer for the
    # whole batch (the reference re-sorts ALL claims before every scan, scheduler.go:598): L2-canonical, see `calibration`.
    comp = None
    if args.components_pods > 0:
        from karpenter_amd.components import split_components
        from karpenter_amd.scheduling import SolveBatch
        import numpy as np

        def solve_components(pods, repeat, want_results=False, shard=True):
            whole = fx.config4(pods=pods, n_types=args.components_types, n_pools=16, seed=42)
            # the host library's split (ksched_split_components: union-find over NodePool pins and topology selectors) and its deal of
            # the components over the ranks by pod count, largest first (LPT) — the same call a Go controller makes
            parts, bins = split_components(whole, bins=world if shard else 1)
            mine = [parts[i] for i in (bins[rank] if shard else range(len(parts)))]
            scheds = [NewScheduler(dict(sub, options=dict(sub["options"], device=device_index)), solver_lib=args.solver_lib) for _, sub in mine]
            best, rs = None, []
            for _ in range(repeat):
                if dist is not None and shard:
                    dist.barrier()
                sync()
                tb = time.perf_counter()
                rs = SolveBatch(scheds, want_results=False) if scheds else []   # the flat C-ABI Results of every component, downloaded
                sync()
                if dist is not None and shard:
                    dist.barrier()
                dt = time.perf_counter() - tb
                best = dt if best is None else min(best, dt)
            if want_results and scheds:
                # the NodeClaims as Python objects for the per-instance-type vector: JSON re-hydration, outside the timed region
                # (as for the headline, whose timed region is ksolve_solve())
                rs = SolveBatch(scheds, want_results=want_results)
            for sc_ in scheds:
                sc_.close()
            return whole, parts, mine, rs, best

        whole, parts, mine, rs, dt = solve_components(args.components_pods, 2, want_results="claims" if not args.no_parity_pin else False)
        comp_inv = None
        if not args.no_parity_pin:
            # no oracle pin exists at this size (27k NodeClaims, 1e11 reference evaluations): every NodeClaim of every component is
            # checked against the reference's rules instead (tests/invariants.py check_claims), outside the timed region
            import invariants
            t_inv = time.perf_counter()
            n_checked = sum(invariants.check_claims(sub, r, expect_pods=sum(g["count"] for g in sub["podGroups"]))["node_claims"] for (_, sub), r in zip(mine, rs))
            comp_inv = {"node_claims_checked": n_checked, "violations": 0, "seconds": time.perf_counter() - t_inv,
                        "checked": "every instance type option of every NodeClaim holds its requests, pod counts add up to the component's pods, requirements inside the NodePool's (tests/invariants.py check_claims)"}
        cvec = np.zeros((len(whole["instanceTypes"]), 2))
        for r in rs:
            cvec += packing_vector(r, len(whole["instanceTypes"]))   # per component by the library (ksolve_packing_vector); summed here, all-reduced below
        totals = [float(sum(r["scheduledPods"] for r in rs)), float(sum(r["counters"]["claims"] for r in rs)), float(sum(r["packingCost"] for r in rs)),
                  max([r["timings"][0]["pack_kernel_ms"] for r in rs], default=0.0)]
        engines = sorted({r["counters"].get("engine") for r in rs})
        if dist is not None:
            t = torch.tensor([dt, totals[3]], dtype=torch.float64, device=reduce_device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt, totals[3] = float(t[0].item()), float(t[1].item())
            v = torch.tensor(cvec.reshape(-1).tolist() + totals[:3], dtype=torch.float64, device=reduce_device)
            dist.all_reduce(v, op=dist.ReduceOp.SUM)       # the per-instance-type (count, $/h) vector of the whole batch: one collective
            cvec = np.array(v[:-3].tolist()).reshape(-1, 2)
            totals[:3] = [float(x) for x in v[-3:].tolist()]
        if rank == 0:
            nzc = cvec[:, 0] > 0
            comp = {"workload": f"BASELINE configs[3]: {args.components_pods} pods x {args.components_types} types x 16 NodePools, every pod pinned to its pool",
                    "components": len(parts), "ranks": world, "sharding": "components dealt over the ranks by pod count, largest first (ksched_split_components), one batched launch per rank" if world > 1 else "all components in one launch on one GPU",
                    "pods": int(round(totals[0])), "seconds": dt, "value": totals[0] / dt, "unit": "pods/s",
                    "timed_region": "ksolve_solve_batch(): classing, queue sort, pack, finalize, download of the flat C-ABI Results of every component (+ ksolve_packing_vector per component)",
                    "node_claims": int(round(totals[1])), "packing_cost_per_hour": totals[2], "pack_kernel_ms": totals[3], "engines": engines,
                    "per_instance_type": {"launch_types_used": int(nzc.sum()), "claims_from_vector": int(round(cvec[:, 0].sum())), "cost_from_vector": float(cvec[:, 1].sum()),
                                          "vector": "count and $/h per instance type over all components" + (", summed over ranks with one all-reduce" if world > 1 else "")},
                    "parity": "each component bit-identical to the oracle on that component (tests); the union vs ONE Solve() of the whole batch: L2-canonical, see calibration",
                    "invariants": comp_inv}
            if args.whole_batch_pods > 0 and world == 1:
                # the same shape as ONE coupled problem: what the reference's Solve() computes for the whole batch, bit for bit
                wp = fx.config4(pods=args.whole_batch_pods, n_types=args.components_types, n_pools=16, seed=42)
                wp["options"]["device"] = device_index
                sw = NewScheduler(wp, solver_lib=args.solver_lib)
                tb = time.perf_counter(); rw_ = sw.Solve(want_results=False); wdt = time.perf_counter() - tb
                wb = {"pods": args.whole_batch_pods, "seconds": wdt, "value": rw_["scheduledPods"] / wdt, "unit": "pods/s", "node_claims": rw_["counters"]["claims"],
                      "engine": rw_["counters"].get("engine"), "pack_kernel_ms": rw_["timings"][0]["pack_kernel_ms"], "oracle_pin": None,
                      "note": "ONE Solve() over all 16 NodePools: exact (bit-identical to the reference's answer for the batch), not shardable"}
                wpin = os.path.join(ROOT, "tests", "golden", "fullsize", f"config4_p{args.whole_batch_pods}_t{args.components_types}_s42_x16.json")
                if os.path.exists(wpin) and not args.no_parity_pin:
                    with open(wpin) as f:
                        g = json.load(f)
                    fullw = sw.Solve(want_results=True)
                    dw, _ = parity.results_digest(fullw)
                    wb["oracle_pin"] = {"pin": os.path.relpath(wpin, ROOT), "digest_matches_oracle": dw == g["digest"], "reference_bin_evaluations_match": fullw["counters"]["referenceBinEvaluations"] == g["binEvaluations"]}
                    if not (wb["oracle_pin"]["digest_matches_oracle"] and wb["oracle_pin"]["reference_bin_evaluations_match"]):
                        raise SystemExit(f"bench.py: the whole-batch configs[3] Results differ from the oracle's pin {wb['oracle_pin']}")
                sw.close()
                comp["whole_batch"] = wb
            if args.whole_batch_exact_pods > 0 and world == 1:
                # ... and at the configuration's own size: 10M pods as ONE Solve(). 27,345 in-flight NodeClaims: the cursor engine with
                # claim state and claim order in HBM (plan 2, round 4; round 3: the general engine, 702 s). One Solve() is timed — the
                # first of the handle, including the attempt on the LDS plan that tells the library which plan to take.
                ep = fx.config4(pods=args.whole_batch_exact_pods, n_types=args.components_types, n_pools=16, seed=42)
                ep["options"] = dict(ep["options"], device=device_index, maxClaims=max(65536, args.whole_batch_exact_pods // 100))
                tb = time.perf_counter(); se = NewScheduler(ep, solver_lib=args.solver_lib); e_new = time.perf_counter() - tb
                tb = time.perf_counter(); re_ = se.Solve(want_results="claims"); edt = time.perf_counter() - tb
                ce = re_["counters"]
                ex = {"pods": args.whole_batch_exact_pods, "new_scheduler_s": e_new, "seconds": edt, "value": re_["scheduledPods"] / edt, "unit": "pods/s", "node_claims": ce["claims"],
                      "engine": ce.get("engine"), "cursor_memory_plan": ce.get("cursorMemoryPlan"), "cursor_attempts": ce.get("cursorAttempts"), "pack_kernel_ms": re_["timings"][0]["pack_kernel_ms"],
                      "reference_bin_evaluations": ce["referenceBinEvaluations"], "packing_cost_per_hour": sum(d for _, _, d in re_.get("packingVector", [])) or None,
                      "components_ratio": {"cost": None, "claims": None}, "oracle_pin": None,
                      "note": "the exact answer of the reference for the whole batch; the component split above is what the north_star shards and what `value` of this leg reports"}
                if comp.get("packing_cost_per_hour") and ex["packing_cost_per_hour"] and comp.get("pods") == args.whole_batch_exact_pods:
                    ex["components_ratio"] = {"cost": comp["packing_cost_per_hour"] / ex["packing_cost_per_hour"], "claims": comp["node_claims"] / max(1, ce["claims"])}
                if not args.no_parity_pin:
                    import invariants
                    ex["invariants"] = invariants.check_claims(ep, re_, expect_pods=args.whole_batch_exact_pods)
                    epin = os.path.join(ROOT, "tests", "golden", "fullsize", f"config4_p{args.whole_batch_exact_pods}_t{args.components_types}_s42_x16.json")
                    if os.path.exists(epin):
                        with open(epin) as f:
                            g = json.load(f)
                        fulle = se.Solve(want_results=True)
                        de, _ = parity.results_digest(fulle)
                        ex["oracle_pin"] = {"pin": os.path.relpath(epin, ROOT), "digest_matches_oracle": de == g["digest"], "reference_bin_evaluations_match": fulle["counters"]["referenceBinEvaluations"] == g["binEvaluations"], "oracle_seconds_offline": g.get("oracleSeconds")}
                        if not (ex["oracle_pin"]["digest_matches_oracle"] and ex["oracle_pin"]["reference_bin_evaluations_match"]):
                            raise SystemExit(f"bench.py: the exact configs[3] batch differs from the oracle's pin {ex['oracle_pin']}")
                se.close()
                comp["whole_batch_exact"] = ex
            if args.components_calibration_pods > 0:
                cp = args.components_calibration_pods
                whole_p = fx.config4(pods=cp, n_types=args.components_types, n_pools=16, seed=42)
                whole_p["options"]["device"] = device_index
                sw = NewScheduler(whole_p, solver_lib=args.solver_lib)
                rw = sw.Solve(want_results=False)
                sw.close()
                _, cparts, cmine, rc, _ = solve_components(cp, 1, want_results=True, shard=False)
                if not args.no_cpu_baseline and not args.no_parity_pin:
                    # every component of the calibration batch against the oracle's Solve() of that component, in the run
                    # (the 10M-pod components are beyond the oracle: O(pods x claims))
                    import oracle
                    bad = [name for (name, sub), r in zip(cmine, rc) if parity.results_digest(r)[0] != parity.results_digest(oracle.solve(sub))[0]]
                    if bad:
                        raise SystemExit(f"bench.py: configs[3] components differ from the oracle: {bad}")
                    comp["components_check"] = {"pods": cp, "components": len(cmine), "all_digests_match_oracle": True}
                comp["calibration"] = {"pods": cp, "whole_batch": {"node_claims": rw["counters"]["claims"], "packing_cost_per_hour": rw["packingCost"], "engine": rw["counters"].get("engine")},
                                       "components": {"node_claims": sum(r["counters"]["claims"] for r in rc), "packing_cost_per_hour": sum(r["packingCost"] for r in rc)},
                                       "claims_delta": sum(r["counters"]["claims"] for r in rc) - rw["counters"]["claims"],
                                       "cost_rel_delta": (sum(r["packingCost"] for r in rc) - rw["packingCost"]) / rw["packingCost"]}

    # ---- BASELINE configs[4] (all ranks take part): the consolidation replay over a resident cluster ----
    sweep = None
    if args.sweep_nodes > 0:
        sweep = config4_sweep(args, device_index, rank, world)
        if args.sweep_topology_sample > 0 and world == 1:
            sweep["with_topology_pods"] = config4_sweep(args, device_index, rank, world, topology=True)
        if dist is not None:
            names = ("delete", "replace", "no-op")
            v = torch.tensor([float(sweep["candidates"])] + [float(sweep["decisions"].get(k, 0)) for k in names], dtype=torch.float64, device=reduce_device)
            dist.all_reduce(v, op=dist.ReduceOp.SUM)
            t = torch.tensor([sweep["seconds"]["library_call"]], dtype=torch.float64, device=reduce_device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            sweep["ranks"] = world
            sweep["candidates_all_ranks"] = int(v[0].item())
            sweep["decisions_all_ranks"] = {k: int(v[1 + i].item()) for i, k in enumerate(names)}
            sweep["value"] = v[0].item() / t.item()   # probes of all ranks / slowest rank's call
            sweep["sharding"] = "candidates dealt out round-robin over the ranks, cluster tables replicated, no data-path collective; verdict counts summed with one all-reduce"
            mn = sweep.get("multi_node")
            if mn is not None:     # windows dealt out round-robin: probes and displaced pods of all ranks / the slowest rank's call
                v = torch.tensor([float(mn.get("probes", 0)), float(mn.get("displaced_pods", 0)), float(mn.get("commands", {}).get("nodes_consolidated", 0))], dtype=torch.float64, device=reduce_device)
                dist.all_reduce(v, op=dist.ReduceOp.SUM)
                t = torch.tensor([mn.get("seconds", {}).get("library_call", 0.0)], dtype=torch.float64, device=reduce_device)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                mn.update(ranks=world, probes_all_ranks=int(v[0].item()), displaced_pods_all_ranks=int(v[1].item()), nodes_consolidated_all_ranks=int(v[2].item()))
                if t.item() > 0:
                    mn["value"] = v[0].item() / t.item(); mn["pods_placed_per_s"] = v[1].item() / t.item()

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    ms_per_step = elapsed * 1e3 / args.steps
    value = scheduled / (elapsed / args.steps)
    c = dict(last["counters"])
    c["templates"] = len(prob["nodePools"])
    abytes, rec = algorithmic_bytes(c)
    pack_ms = sum(t["pack_kernel_ms"] for t in timings) / len(timings)
    cls_ms = sum(t["classify_ms"] for t in timings) / len(timings)
    rh_ms = sum(t.get("row_hash_ms", 0.0) for t in timings) / len(timings)
    achieved = abytes / (pack_ms * 1e-3) / 1e9
    kernel = "ksolve_pack_fast" if c.get("engine") == "cursor" else "ksolve_pack_lite"
    # HBM bytes per launch from the TCC counters: collected by scripts/gpu_pmc_traffic.sh (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in
    # separate passes, --kernel-trace only; FETCH_SIZE doubled on gfx950 as MI355X_MICROARCH.md prescribes) on this workload and
    # written to profiles/round3/pmc_traffic.json together with a hash of the kernel sources. A figure is used only for the build
    # it was measured on: after any change of the sources it reads null until the script has run again.
    traffic, stream_traffic, traffic_note, sq, pmc_doc = None, None, "not measured for this build (scripts/gpu_r6_pmc.sh)", None, None
    try:
        with open(os.path.join(ROOT, "profiles", "round6", "pmc_traffic.json")) as f:
            pmc = json.load(f)
        if pmc.get("source_sha") == source_sha() and pmc.get("pods") == args.pods and pmc.get("types") == args.types:
            pk = next((n for n in sorted(pmc["kernels"]) if n.startswith(kernel)), kernel)   # ksolve_pack_fast_g<plan>r<rows>: one kernel per memory plan and row count
            traffic = pmc["kernels"].get(pk, {}).get("traffic_bytes_per_launch")
            stream_traffic = pmc["kernels"].get("ksolve_row_hash_coop2", {}).get("traffic_bytes_per_launch")
            traffic_note = "TCC FETCH_SIZE x2 + WRITE_SIZE per launch, rocprofv3 --pmc on this build (profiles/round6/pmc_traffic.json)"
            k = pmc["kernels"].get(pk, {})
            pmc_doc = pmc
            if "SQ_INSTS_VALU" in k:
                per = lambda name: k[name]["per_launch"] / args.pods if name in k else None
                sq = {"waves": k.get("SQ_WAVES", {}).get("per_launch"), "valu_per_pod": per("SQ_INSTS_VALU"), "salu_per_pod": per("SQ_INSTS_SALU"), "lds_per_pod": per("SQ_INSTS_LDS"),
                      "wave_cycles_per_pod": 4 * per("SQ_WAVE_CYCLES") if per("SQ_WAVE_CYCLES") else None, "wait_cycles_per_pod": 4 * per("SQ_WAIT_ANY") if per("SQ_WAIT_ANY") else None,
                      "note": "SQ_WAVE_CYCLES / SQ_WAIT_ANY count in units of four shader cycles"}
    except (OSError, KeyError, ValueError):
        pass
    pin = None
    if not args.no_parity_pin:
        pin_path = os.path.join(ROOT, "tests", "golden", "fullsize", f"config2_p{args.pods}_t{args.types}_s{seed}.json")
        if os.path.exists(pin_path):
            with open(pin_path) as f:
                g = json.load(f)
            pin = {"pin": os.path.relpath(pin_path, ROOT), "digest_matches_oracle": digest == g["digest"], "claims_match": len(full["newNodeClaims"]) == g["claims"],
                   "reference_bin_evaluations_match": full["counters"]["referenceBinEvaluations"] == g["binEvaluations"], "oracle_seconds_offline": g["oracleSeconds"]}
            if not (pin["digest_matches_oracle"] and pin["reference_bin_evaluations_match"]):
                raise SystemExit(f"bench.py: the timed problem's Results differ from the oracle's pin {pin}")   # the reference's in-bench gate (scheduling_benchmark_test.go:176-181), bit-exact
    # what the classing kernel reads and writes per row: the mask table(s) + flag words, requests, toleration mask, the slot it
    # writes; all-nil minValues tables are not streamed, and the row's bookkeeping words (next_variant, creation, uid, pending:
    # 29 B of B_pod) belong to other kernels — they are NOT in this numerator (round-3 review)
    row_bytes = (1 if c.get("strictTableShared") else 2) * (8 * c["reqWords"] + 16) + 8 * c["resources"] + 8 + 4
    stream_bytes = c["rows"] * row_bytes
    nz = vec[:, 0] > 0
    out = {
        "metric": "pods scheduled/sec (Solve())", "value": value, "unit": "pods/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int64", "data": "synthetic",
        "config": {"workload": f"BASELINE configs[1]: {args.pods} pods/GPU x {args.types} kwok instance types, nodeSelector + taint/toleration, 2 NodePools",
                   "pods_per_gpu": args.pods, "instance_types": args.types, "sharding": "one independent scheduling problem per GPU (weak scaling)" if world > 1 else "single coupled problem on one GPU",
                   "timed_region": "ksolve_solve(): classing, queue sort, pack, finalize, download of the flat C-ABI Results; JSON re-hydration for Python excluded"},
        "packing": {"node_claims": claims, "packing_cost_per_hour": cost, "pods_scheduled": scheduled,
                    "per_instance_type": {"launch_types_used": int(nz.sum()), "max_claims_on_one_type": int(vec[:, 0].max()) if len(vec) else 0,
                                          "vector": "count and $/h per instance type, all ranks summed" + (" with one all-reduce" if world > 1 else "")}},
        "engine": c.get("engine"),
        "parity": {"results_digest": digest, "oracle_pin": pin},
        # The kernel of the path that streams from HBM: pod classing — every pod row (requirement masks of both tables, flag words,
        # requests, toleration mask) is read once, hashed and matched against the class table. rows x B_pod algorithmic bytes
        # ("the feasibility pre-pass ... is the one to hold to the >= 40% HBM target"); HIP events around the kernel alone.
        "roofline": {"kernel": "ksolve_row_hash_coop2", "bound": "hbm", "achieved": stream_bytes / (rh_ms * 1e-3) / 1e9 if rh_ms > 0 else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": stream_bytes / (rh_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if rh_ms > 0 else None, "traffic": stream_traffic, "traffic_source": traffic_note,
                     "algorithmic_bytes": stream_bytes, "bytes_per_row": row_bytes, "rows": c["rows"], "avg_kernel_ms": rh_ms, "classing_phase_ms": cls_ms,
                     "achieved_from_traffic": (stream_traffic / (rh_ms * 1e-3) / 1e9) if (stream_traffic and rh_ms > 0) else None,
                     "frac_from_traffic": (stream_traffic / (rh_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if (stream_traffic and rh_ms > 0) else None,
                     "mask_tables": 1 if c.get("strictTableShared") else 2,
                     # ... and beside the streaming kernel, the kernel the step's time goes to (round-5 review: the record the driver
                     # keeps must show both): its share of the step, what it moves, and SURVEY §8(d)'s notional bytes over its time
                     "dominant": {"kernel": kernel, "share_of_step_time": pack_ms / ms_per_step if ms_per_step else None, "avg_kernel_ms": pack_ms, "bound": "latency / instruction issue of ONE wavefront (1 of the chip's 1,024 SIMDs)",
                                  "waves": (sq or {}).get("waves", 1), "traffic": traffic,
                                  "frac_notional_survey_sizes": (c["pods"] * 188 + c["referenceBinEvaluations"] * 248 + c["pods"] * 248 + c["instanceTypes"] * c.get("templates", 1) * 185) / (pack_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if pack_ms else None,
                                  "frac_notional_layout_sizes": achieved / HBM_PEAK_GBS,
                                  "note": "notional = the bytes SURVEY §8(d) charges the REFERENCE's scan for the same answer (V = its bin evaluations) over this kernel's time; "
                                          "the kernel itself moves `traffic` bytes per launch: it is not an HBM-bound kernel and claims no HBM roofline"},
                     "note": "numerator = the bytes this kernel reads and writes per row (not the whole B_pod row of DESIGN.md §3: 245 B with its bookkeeping words); "
                             "the table of the headline problem (220 MB) is smaller than the 256 MiB Infinity Cache — profiles/round4/classing_rows.json has the same kernel at 2M and 4M rows"},
        # The pack kernel is a serial first-fit chain on ONE wavefront: bound by the instruction issue and the dependent LDS round
        # trips of a lone wave, not by HBM — no roofline is claimed for it. `achieved` is what it really moves.
        # sq_counters: instructions and wave cycles per pod of THIS build's pack kernel (rocprofv3 --pmc SQ_*, profiles/round6/pmc_traffic.json),
        # null when the sources changed since they were measured.
        "pack_kernel": {"kernel": kernel, "bound": "latency / instruction issue of one wavefront", "avg_kernel_ms": pack_ms, "us_per_pod": pack_ms * 1e3 / max(1, args.pods), "sq_counters": sq, "traffic": traffic, "traffic_source": traffic_note,
                        "achieved": (traffic / (pack_ms * 1e-3) / 1e9) if traffic else None, "unit": "GB/s",
                        "reference_equivalent": {"bytes": abytes, "GBps": achieved, "frac_of_hbm_peak": achieved / HBM_PEAK_GBS, "records": rec,
                                                 "note": "SURVEY §8(d) formula P*B_pod + V*B_bin + P*B_bin + N_it*T*B_it with V = the bins the REFERENCE evaluates (referenceBinEvaluations, equal to the "
                                                         "oracle's count) and the record sizes of the layout in use (SURVEY's sketch: 188 / 248 / 185 B): the bytes the reference's scan would "
                                                         "touch, not what this kernel moves — cursors and permanent rejections make almost all of those evaluations unnecessary"}},
        # What Solve() costs through the real boundary: the cgo shim flattens and uploads INSIDE Solve() (go/ksolve_shim.go:124-150)
        # and rehydrates Results after it, and the reference's own Solve() includes updateCachedPodData for every pod
        # (scheduler.go:453-455). Here: NewScheduler (the host library parses the problem document — pods arrive as a few hundred
        # groups, which spares JSON, not flattening: every pod becomes a row —, flattens, ksolve_create uploads), one timed solve, and
        # a solve that also rehydrates every NodeClaim and pod assignment into the Results document.
        "end_to_end": {"new_scheduler_s": t_open, "process_start_up_s": t_first, "upload_ms": timings[0].get("upload_us", 0.0) * 1e-3, "solve_s": elapsed / args.steps,
                       "solve_and_rehydrate_by_position_s": t_pos, "pods_per_s_through_the_boundary": scheduled / world / (t_open + t_pos),
                       "solve_and_rehydrate_uid_text_s": t_full, "pods_per_s_with_uid_text": scheduled / world / (t_open + t_full),
                       "note": "rank 0's problem; flatten (a few host threads) + upload + solve + download + re-hydration, nothing overlapped. by_position: NodeClaims as objects, "
                               "every pod put on its NodeClaim from the flat pod_assignment / pod_slot arrays (what the cgo shim does); uid_text: the Results document with "
                               "a million uid strings formatted by the host library and parsed by Python"},
        "phases_ms": {k: sum(t.get(k, 0.0) for t in timings) / len(timings) for k in ("pack_kernel_ms", "classify_ms", "row_hash_ms", "sort_ms", "it_index_ms")},
        "counters": c,
    }
    sched.close()
    if args.topology_pods > 0 and world == 1:
        # BASELINE configs[2] shape (podAntiAffinity + 3-zone topologySpreadConstraints, the reference benchmark's diverse mix): ONE
        # Solve() of the whole batch on the general engine (BIG variant: every anti-affinity pod is its own NodeClaim). The size asked
        # for is timed; its Results are checked against the oracle's pin when one exists for that size (the oracle is O(pods x
        # claims): tests/golden/make_fullsize_digests.py), and the largest pinned size below it is solved and checked as well.
        import glob
        import re

        def topology_run(pods, timed_solves):
            p3 = fx.config3(pods=pods, n_types=args.types, seed=42)
            p3["options"]["device"] = device_index
            s3 = NewScheduler(p3, solver_lib=args.solver_lib)
            best, r3 = None, None
            for _ in range(timed_solves):
                tb = time.perf_counter()
                r3 = s3.Solve(want_results=False)
                dt = time.perf_counter() - tb
                best = dt if best is None else min(best, dt)
            e = {"pods": pods, "seconds": best, "value": r3["scheduledPods"] / best, "unit": "pods/s", "node_claims": r3["counters"]["claims"],
                 "pack_kernel_ms": r3["timings"][0]["pack_kernel_ms"], "engine": r3["counters"].get("engine"), "oracle_pin": None, "invariants": None}
            # the leg's dominant kernel: the spread engine (csrc/topo_engine.h, one wavefront) — bound by the instruction issue and the
            # dependent round trips of a lone wavefront like every pack kernel here; `reference_equivalent` = SURVEY §8(d)'s bytes for
            # the bins the REFERENCE evaluates (2.0e10 at 1M pods: every anti-affinity pod walks every claim) over the kernel's time
            c3 = dict(r3["counters"]); c3["templates"] = len(p3["nodePools"])
            ab3, rec3 = algorithmic_bytes(c3)
            pk_ms = e["pack_kernel_ms"]
            kname = {"spread": "ksolve_pack_topo", "general": "ksolve_pack_big" if c3["claims"] > 8192 else "ksolve_pack"}.get(e["engine"], "?")
            sq3 = None
            if pmc_doc is not None and pods == args.topology_pods and kname in pmc_doc.get("topology_kernels", {}):
                k3 = pmc_doc["topology_kernels"][kname]
                per3 = lambda name: k3[name]["per_launch"] / pods if name in k3 else None
                sq3 = {"waves": k3.get("SQ_WAVES", {}).get("per_launch"), "valu_per_pod": per3("SQ_INSTS_VALU"), "salu_per_pod": per3("SQ_INSTS_SALU"), "lds_per_pod": per3("SQ_INSTS_LDS"),
                       "vmem_per_pod": (per3("SQ_INSTS_VMEM_RD") or 0) + (per3("SQ_INSTS_VMEM_WR") or 0) if per3("SQ_INSTS_VMEM_RD") is not None else None,
                       "wave_cycles_per_pod": 4 * per3("SQ_WAVE_CYCLES") if per3("SQ_WAVE_CYCLES") else None, "wait_cycles_per_pod": 4 * per3("SQ_WAIT_ANY") if per3("SQ_WAIT_ANY") else None,
                       "traffic_bytes_per_launch": k3.get("traffic_bytes_per_launch"), "note": "SQ_WAVE_CYCLES / SQ_WAIT_ANY count in units of four shader cycles; rocprofv3 --pmc on this build (profiles/round6/pmc_traffic.json)"}
            e["pack_kernel"] = {"kernel": kname, "bound": "latency / instruction issue of one wavefront", "avg_kernel_ms": pk_ms, "us_per_pod": pk_ms * 1e3 / max(1, pods), "sq_counters": sq3,
                                "reference_equivalent": {"bytes": ab3, "GBps": ab3 / (pk_ms * 1e-3) / 1e9 if pk_ms else None, "frac_of_hbm_peak": ab3 / (pk_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if pk_ms else None,
                                                         "records": rec3, "reference_bin_evaluations": c3["referenceBinEvaluations"]}}
            pin_path = os.path.join(ROOT, "tests", "golden", "fullsize", f"config3_p{pods}_t{args.types}_s42.json")
            full3 = None
            if not args.no_parity_pin:
                # with or without a pin: the placements replayed in queue order against the reference's topology rules
                # (tests/invariants.py check_topology_mix: anti-affinity, hostname spread, the zonal skew rule with its minimum-count
                # choice, zonal self-affinity) and the claim-level packing rules; a violation stops the run
                import invariants
                full3 = s3.Solve(want_results=True)
                t_inv = time.perf_counter()
                rep = invariants.check_topology_mix(p3, full3)
                invariants.check_claims(p3, full3, expect_pods=pods)
                e["invariants"] = {"violations": 0, "seconds": time.perf_counter() - t_inv, "decisions_replayed": rep,
                                   "checked": "every pod placed once, in queue order; every zonal-spread / hostname-spread / zonal-affinity / hostname-anti-affinity decision re-derived from the domain counts at that moment; every instance type option holds its NodeClaim's requests"}
            if os.path.exists(pin_path) and not args.no_parity_pin:
                with open(pin_path) as f:
                    g = json.load(f)
                d3, _ = parity.results_digest(full3)
                e["oracle_pin"] = {"pin": os.path.relpath(pin_path, ROOT), "digest_matches_oracle": d3 == g["digest"], "reference_bin_evaluations_match": full3["counters"]["referenceBinEvaluations"] == g["binEvaluations"],
                                   "oracle_seconds_offline": g["oracleSeconds"], "oracle_threads": g.get("oracleThreads", 1)}
                if not (e["oracle_pin"]["digest_matches_oracle"] and e["oracle_pin"]["reference_bin_evaluations_match"]):
                    raise SystemExit(f"bench.py: the configs[2] problem's Results differ from the oracle's pin {e['oracle_pin']}")
            s3.close()
            return e
        e = topology_run(args.topology_pods, 1 if args.topology_pods > 300_000 else 2)
        e["workload"] = f"BASELINE configs[2] shape: {args.topology_pods} pods, anti-affinity + zonal / hostname spread + zonal affinity, {args.types} types"
        if e["oracle_pin"] is None:
            pinned = sorted(int(re.search(r"_p(\d+)_", os.path.basename(f)).group(1)) for f in glob.glob(os.path.join(ROOT, "tests", "golden", "fullsize", f"config3_p*_t{args.types}_s42.json")))
            pinned = [n for n in pinned if n < args.topology_pods]
            if pinned and not args.no_parity_pin:
                e["largest_pinned_size"] = topology_run(pinned[-1], 1)
        out["config2_topology"] = e
    if args.beyond_lds_pods > 0 and world == 1:
        # right above the benchmarked size: the same mix with more NodeClaims than the cursor engine's LDS plan holds
        pb = fx.config2(pods=args.beyond_lds_pods, n_types=args.types, seed=42)
        pb["options"]["device"] = device_index
        sb = NewScheduler(pb, solver_lib=args.solver_lib)
        tb = time.perf_counter(); rb1 = sb.Solve(want_results=False); first_s = time.perf_counter() - tb     # LDS plan until it runs out of claims, then the HBM plan
        tb = time.perf_counter(); rb = sb.Solve(want_results=False); steady_s = time.perf_counter() - tb     # the handle remembers: the HBM plan from the start
        bl = {"workload": f"BASELINE configs[1] mix at {args.beyond_lds_pods} pods x {args.types} types: more in-flight NodeClaims than the cursor engine's LDS plan holds",
              "pods": args.beyond_lds_pods, "node_claims": rb["counters"]["claims"], "engine": rb["counters"].get("engine"), "claim_state_in_hbm": rb["counters"].get("cursorClaimStateInHBM"),
              "first_solve_s": first_s, "seconds": steady_s, "value": rb["scheduledPods"] / steady_s, "unit": "pods/s", "pack_kernel_ms": rb["timings"][0]["pack_kernel_ms"],
              "us_per_pod": steady_s * 1e6 / max(1, rb["scheduledPods"]), "oracle_pin": None,
              "note": "first_solve_s includes the attempt with the LDS plan (it stops when it needs claim 3,009) and the run with the claims' state in HBM; later solves of the handle start there"}
        bpin = os.path.join(ROOT, "tests", "golden", "fullsize", f"config2_p{args.beyond_lds_pods}_t{args.types}_s42.json")
        if os.path.exists(bpin) and not args.no_parity_pin:
            with open(bpin) as f:
                g = json.load(f)
            fb = sb.Solve(want_results=True)
            db, _ = parity.results_digest(fb)
            bl["oracle_pin"] = {"pin": os.path.relpath(bpin, ROOT), "digest_matches_oracle": db == g["digest"], "reference_bin_evaluations_match": fb["counters"]["referenceBinEvaluations"] == g["binEvaluations"], "oracle_seconds_offline": g["oracleSeconds"]}
            if not (bl["oracle_pin"]["digest_matches_oracle"] and bl["oracle_pin"]["reference_bin_evaluations_match"]):
                raise SystemExit(f"bench.py: the configs[1] mix at {args.beyond_lds_pods} pods differs from the oracle's pin {bl['oracle_pin']}")
        elif not args.no_parity_pin:
            import invariants
            fb = sb.Solve(want_results="claims")
            bl["invariants"] = invariants.check_claims(pb, fb, expect_pods=args.beyond_lds_pods)
        sb.close()
        out["config1_beyond_lds"] = bl
    if comp is not None:
        out["config3_components"] = comp
    if sweep is not None:
        out["config4_sweep"] = sweep
    if args.batch_problems > 0 and world == 1:
        # Independent problems (NodePool components / consolidation probes) in ONE launch of the pack kernel:
        # block b = the wavefront of problem b. Reported beside the headline, never part of `value`.
        from karpenter_amd.scheduling import SolveBatch
        scheds = [NewScheduler(dict(fx.config2(pods=args.batch_pods, n_types=args.types, seed=1000 + i), options={"device": device_index, "maxClaims": 1024}), solver_lib=args.solver_lib) for i in range(args.batch_problems)]   # 1024 in-flight claims per problem keep the LDS plan under 80 KB: two problems per CU
        SolveBatch(scheds, want_results=False)   # warm-up
        tb = time.perf_counter()
        rs = SolveBatch(scheds, want_results=False)
        dt = time.perf_counter() - tb
        out["batched"] = {"problems": args.batch_problems, "pods_each": args.batch_pods, "seconds": dt,
                          "value": sum(r["scheduledPods"] for r in rs) / dt, "unit": "pods/s",
                          "pack_kernel_ms": rs[0]["timings"][0]["pack_kernel_ms"],
                          "note": "ksolve_solve_batch: one wavefront per independent problem, one kernel launch; includes prepass and result download of every problem"}
        for sc_ in scheds:
            sc_.close()
    if not args.no_cpu_baseline and world == 1:
        import oracle  # the checker, used here only as the reported CPU baseline
        sample = fx.config2(pods=args.cpu_sample, n_types=args.types, seed=42)
        runs = []
        for _ in range(max(1, args.cpu_runs)):
            r = oracle.solve(sample)
            runs.append(r["counters"]["solveSeconds"])
        secs = statistics.median(runs)
        out["cpu_baseline"] = {"value": args.cpu_sample / secs, "unit": "pods/s", "cores": 1, "kind": "port",
                               "sample": f"{args.cpu_sample} pods of the same configs[1] mix x {args.types} instance types, fresh scheduler, oracle C++ restatement of the Go Solve(): "
                                         f"median of {len(runs)} runs ({secs:.2f} s); the oracle is O(pods x claims), so its rate falls with size "
                                         f"(offline at the full 1M pods: see parity.oracle_pin.oracle_seconds_offline)",
                               "seconds": secs, "runs_seconds": runs, "bin_evaluations": r["counters"]["binEvaluations"]}
        # like for like: the same oracle on the HEADLINE problem itself (all args.pods pods), timed offline when its pin was made —
        # the oracle walks every in-flight claim for every pod, O(pods x claims), so its rate at the full size is far below the sample's
        if pin is not None and pin.get("oracle_seconds_offline"):
            out["cpu_baseline_full_size"] = {"value": args.pods / pin["oracle_seconds_offline"], "unit": "pods/s", "cores": 1, "kind": "port", "seconds": pin["oracle_seconds_offline"],
                                             "sample": f"the whole timed problem ({args.pods} pods), one run, offline (tests/golden/make_fullsize_digests.py; the pin's oracleSeconds): "
                                                       "O(pods x claims) — 1.0e9 bin evaluations at 1M pods"}
        # (An N-thread form of the oracle exists — ORACLE_THREADS: the candidates of one addToInflightNode call over a worker pool, as
        # parallelizeUntil does, scheduler.go:939-961 — and is not reported: at this mix a scan is a few dozen candidates and the
        # fan-out costs more than it saves; rounds 3 and 4 measured 1.00x on 64 cores and 0.75x on 6.)
        if not args.no_host_engine_baseline and not args.solver_lib:
            # the honest baseline for the ALGORITHM: this repository's own engine source compiled for ONE host core (the test
            # emulation of the device code, tests/emu — a checker, never a product path), same problem, same Results
            try:
                emu = parity.build_emu()
                os.environ["KSOLVE_TEST_SOLVER_LIB"] = "1"   # baseline leg only, after every measured solve: the emulation is handed in through the test hook
                eb = {}
                for eng in ("general", "cursor"):
                    pe = fx.config2(pods=args.pods, n_types=args.types, seed=42)
                    pe["options"]["engine"] = eng
                    se = NewScheduler(pe, solver_lib=emu)
                    re_ = se.Solve(want_results=False)
                    eb[eng] = {"pack_seconds": re_["timings"][0]["pack_kernel_ms"] * 1e-3, "pods_per_s": args.pods / (re_["timings"][0]["pack_kernel_ms"] * 1e-3)}
                    se.close()
                out["cpu_baseline_engine_host"] = {"cores": 1, "kind": "this repository's engine source on one host core (test emulation of the device code: a wave-wide operation is a loop over 64 lanes there)", "pods": args.pods, "engines": eb,
                                                   "device_over_host_pack": (args.pods / (pack_ms * 1e-3)) / eb["cursor"]["pods_per_s"] if c.get("engine") == "cursor" else None}
            except Exception as e:  # noqa: BLE001 - a missing host compiler must not lose the measurement
                out["cpu_baseline_engine_host"] = {"error": str(e)[:200]}
    if args.solver_lib:
        out["data"] = "synthetic; TEST HOOK --solver-lib (host emulation of the engine): contract check only, not a measurement"
    print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
