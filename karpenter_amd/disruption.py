"""Consolidation simulator — the part of pkg/controllers/disruption that drives Solve():

    SimulateScheduling                       helpers.go:53-155
    consolidation.computeConsolidation       consolidation.go:159-256   (DELETE / REPLACE / no-op, price filter)
    MultiNodeConsolidation.firstNConsolidationOption   multinodeconsolidation.go:117-207   (binary search over a prefix)
    SingleNodeConsolidation.ComputeCommands  singlenodeconsolidation.go:55-126             (first candidate that works)
    sortCandidates / SavingsRatio            consolidation.go:149-154 ; types.go:146
    NodeClaim.RemoveInstanceTypeOptionsByPriceAndMinValues   nodeclaim.go:411-420 ; Offerings.WorstLaunchPrice types.go:587-598

Every probe is an independent Solve() on (cluster − candidates, pending + displaced pods), which is what makes a sweep
data-parallel: probes are independent scheduling problems, each one wavefront on its own CU (`sweep` runs them on
several device sessions concurrently). `solver` is any callable problem -> Results document; the product passes
`lambda p: NewScheduler(p).Solve()`, the parity tests pass the oracle.
Out of scope here (kube side effects): budgets, PDBs, validation delay, taint/launch orchestration.
"""
from __future__ import annotations

import copy
import math
from concurrent.futures import ThreadPoolExecutor

from . import fixtures as fx

DELETE, REPLACE, NOOP = "delete", "replace", "no-op"


def _req_has(r, value):
    """Requirement.Has (requirement.go:275-280) on a rehydrated requirement {complement, values, gte, lte}."""
    if r.get("gte") is not None or r.get("lte") is not None:
        try:
            v = int(value)
        except ValueError:
            return False
        if r.get("gte") is not None and v < r["gte"]:
            return False
        if r.get("lte") is not None and v > r["lte"]:
            return False
    return (value not in r["values"]) if r["complement"] else (value in r["values"])


def _offering_compatible(reqs_by_key, off):
    """reqs.IsCompatible(offering.Requirements, AllowUndefinedWellKnownLabels) for single-valued offering labels."""
    for r in off["requirements"]:
        q = reqs_by_key.get(r["key"])
        if q is not None and not _req_has(q, r["values"][0]):
            return False
    return True


def _capacity_type(off):
    return next(r["values"][0] for r in off["requirements"] if r["key"] == fx.CAPACITY_TYPE)


def worst_launch_price(it, reqs_by_key):
    """Offerings.Available().WorstLaunchPrice(reqs) — types.go:587-598: reserved, then spot, then on-demand."""
    for ct in ("reserved", "spot", "on-demand"):
        prices = [o["price"] for o in it["offerings"] if o.get("available", True) and _offering_compatible(reqs_by_key, o) and _capacity_type(o) == ct]
        if prices:
            return max(prices)
    return math.inf


def candidate_price(cluster, node):
    """Candidate price: the offering the node was launched with (disruption/types.go:161-211)."""
    it = next(t for t in cluster["instanceTypes"] if t["name"] == node["labels"][fx.INSTANCE_TYPE])
    for o in it["offerings"]:
        zone = next(r["values"][0] for r in o["requirements"] if r["key"] == fx.ZONE)
        if zone == node["labels"][fx.ZONE] and _capacity_type(o) == node["labels"][fx.CAPACITY_TYPE]:
            return o["price"]
    return math.inf


def savings_ratio(cluster, node):
    """SavingsRatio = price / disruption cost (types.go:146); disruption cost here = number of reschedulable pods + 1."""
    return candidate_price(cluster, node) / (len(node.get("pods", [])) + 1.0)


def sort_candidates(cluster, nodes):
    """consolidation.sortCandidates — consolidation.go:149-154 (descending ratio; name breaks ties deterministically)."""
    return sorted(nodes, key=lambda n: (-savings_ratio(cluster, n), n["name"]))


def simulate_scheduling(cluster, candidates, solver):
    """helpers.go:53-155: Solve() with the candidates removed and their pods added to the pending set."""
    names = {c["name"] for c in candidates}
    state_nodes = [{k: v for k, v in n.items() if k != "pods"} for n in cluster["nodes"] if n["name"] not in names and not n.get("markedForDeletion")]
    deleting = [n for n in cluster["nodes"] if n.get("markedForDeletion") and n["name"] not in names]
    pods = list(cluster.get("pendingPods", []))
    for c in candidates:
        pods += c.get("pods", [])
    deleting_pods = [p for n in deleting for p in n.get("pods", [])]
    pods += deleting_pods
    # the pods that stay where they are seed the topology counts and the inverse anti-affinity groups of the simulation
    # (NewTopology / countDomains read the cluster's bound pods, topology.go:68-103, :361-459)
    staying = [p for n in cluster["nodes"] if n["name"] not in names and not n.get("markedForDeletion") for p in n.get("pods", [])]
    prob = fx.problem(cluster["instanceTypes"], cluster["nodePools"], copy.deepcopy(pods), well_known=cluster.get("wellKnownLabels", fx.KWOK_WELL_KNOWN),
                      state_nodes=state_nodes, cluster_pods=copy.deepcopy(staying), options=dict(cluster.get("options", {}), consolidationSimulation=True),
                      deleting_node_names=[n["name"] for n in deleting])
    res = solver(prob)
    # pods that landed on an uninitialized node make the decision unsafe (helpers.go:133-153)
    deleting_uids = {p["uid"] for p in deleting_pods}
    errors = dict(res["podErrors"])
    for en in res["existingNodes"]:
        if not en.get("initialized", True):
            for uid in en["pods"]:
                if uid not in deleting_uids:
                    errors[uid] = {"code": 100, "diag": 0}  # UninitializedNodeError
    pending_uids = {p["uid"] for p in cluster.get("pendingPods", [])}
    res = dict(res)
    res["podErrors"] = errors
    res["allNonPendingPodsScheduled"] = not [u for u in errors if u not in pending_uids]  # AllNonPendingPodsScheduled, scheduler.go:388-392
    return res


def compute_consolidation(cluster, candidates, solver):
    """consolidation.go:159-256 → {"decision", "candidates", "replacement": instance type names}."""
    res = simulate_scheduling(cluster, candidates, solver)
    cmd = {"decision": NOOP, "candidates": [c["name"] for c in candidates], "replacement": None, "results": res}
    if not res["allNonPendingPodsScheduled"]:
        return cmd
    claims = res["newNodeClaims"]
    if len(claims) == 0:
        cmd["decision"] = DELETE
        return cmd
    if len(claims) != 1:
        return cmd
    price = sum(candidate_price(cluster, c) for c in candidates)
    claim = claims[0]
    reqs = {r["key"]: r for r in claim["requirements"]}
    all_spot = all(c["labels"][fx.CAPACITY_TYPE] == "spot" for c in candidates)
    ct = reqs.get(fx.CAPACITY_TYPE)
    if all_spot and (ct is None or _req_has(ct, "spot")):
        return cmd  # SpotToSpotConsolidation feature gate is off by default (consolidation.go:261-270)
    by_name = {t["name"]: t for t in cluster["instanceTypes"]}
    cheaper = [n for n in claim["instanceTypes"] if worst_launch_price(by_name[n], reqs) < price]   # nodeclaim.go:411-420
    if not cheaper:
        return cmd
    cmd["decision"] = REPLACE
    cmd["replacement"] = sorted(cheaper)
    cmd["replacementCapacityType"] = "spot" if (ct is None or (_req_has(ct, "spot") and _req_has(ct, "on-demand"))) else None  # consolidation.go:238-243
    return cmd


def first_n_consolidation_option(cluster, candidates, solver, max_n=100):
    """multinodeconsolidation.go:117-207: binary search for the longest prefix that consolidates; same probe sequence."""
    if len(candidates) < 2:
        return {"decision": NOOP, "candidates": []}, []
    lo, hi = 1, min(len(candidates) - 1, max_n - 1) if len(candidates) <= max_n else max_n
    if len(candidates) <= max_n:
        hi = len(candidates) - 1
    last, probes = {"decision": NOOP, "candidates": []}, []
    while lo <= hi:
        mid = (lo + hi) // 2
        cmd = compute_consolidation(cluster, candidates[: mid + 1], solver)
        probes.append((mid + 1, cmd["decision"]))
        valid = cmd["decision"] == DELETE
        if cmd["decision"] == REPLACE:
            # filterOutSameInstanceType (multinodeconsolidation.go:209-246)
            existing = {}
            for c in candidates[: mid + 1]:
                n = c["labels"][fx.INSTANCE_TYPE]
                existing[n] = min(existing.get(n, math.inf), candidate_price(cluster, c))
            max_price = min([existing[n] for n in cmd["replacement"] if n in existing], default=math.inf)
            reqs = {r["key"]: r for r in cmd["results"]["newNodeClaims"][0]["requirements"]}
            by_name = {t["name"]: t for t in cluster["instanceTypes"]}
            cmd["replacement"] = [n for n in cmd["replacement"] if worst_launch_price(by_name[n], reqs) < max_price]
            valid = bool(cmd["replacement"])
        if valid:
            last, lo = cmd, mid + 1
        else:
            hi = mid - 1
    return last, probes


def single_node_consolidation(cluster, candidates, solver):
    """singlenodeconsolidation.go:55-126: the first candidate (in sorted order) with a valid command."""
    for c in candidates:
        cmd = compute_consolidation(cluster, [c], solver)
        if cmd["decision"] != NOOP:
            return cmd
    return {"decision": NOOP, "candidates": []}


class _Recorder:
    """Collects the problems a decision procedure wants solved / replays their results, so that all probes of a sweep can
    go to the device as ONE batched launch."""

    def __init__(self):
        self.problems, self.results, self.replay = [], None, 0

    def __call__(self, prob):
        if self.results is None:
            self.problems.append(prob)
            raise _Deferred()
        r = self.results[self.replay]
        self.replay += 1
        return r


class _Deferred(Exception):
    pass


def sweep_batched(cluster, candidates, batch_solver):
    """Single-node consolidation sweep with every probe in one device launch: `batch_solver(list of problems) -> list of
    Results` (the product passes SolveBatch over NewScheduler sessions: one wavefront per probe)."""
    rec = _Recorder()
    for c in candidates:
        try:
            compute_consolidation(cluster, [c], rec)
        except _Deferred:
            pass
    rec.results = batch_solver(rec.problems)
    return [compute_consolidation(cluster, [c], rec) for c in candidates]


def sweep(cluster, candidates, solver, workers=1):
    """Evaluates computeConsolidation for every single-node candidate. Probes are independent Solve() calls; with
    workers > 1 they run concurrently (each device session owns a stream, each pack kernel is one wavefront on its own
    CU), which is how a consolidation pass fills the chip."""
    if workers <= 1:
        return [compute_consolidation(cluster, [c], solver) for c in candidates]
    with ThreadPoolExecutor(max_workers=workers) as ex:
        return list(ex.map(lambda c: compute_consolidation(cluster, [c], solver), candidates))


def make_cluster(n_nodes=60, pods_per_node=6, n_types=144, seed=1, utilisation=0.5):
    """A synthetic under-utilised cluster (mirrors test/suites/performance/basic_test.go:61-68: scale out, then scale the
    workload down): nodes of random kwok types, each with a few running pods."""
    import random
    rng = random.Random(seed)
    its = fx.kwok_catalog(n_types)
    np_ = fx.node_pool("default")
    np_["nodeClassLabelKey"] = "karpenter.kwok.sh/kwoknodeclass"
    nodes = []
    uid = [0]
    for i in range(n_nodes):
        it = rng.choice([t for t in its if 2 <= int(t["capacity"]["cpu"]) <= 64 and "linux" in t["name"]])
        zone = rng.choice(fx.KWOK_ZONES)
        ct = rng.choice(["spot", "on-demand", "on-demand"])
        cpu_m = int(int(it["capacity"]["cpu"]) * 1000 * utilisation * rng.random())
        k = max(1, min(pods_per_node, cpu_m // 100))
        pods = []
        for j in range(k):
            uid[0] += 1
            pods.append(fx.pod(uid=f"10000000-0000-0000-0000-{uid[0]:012d}", requests={"cpu": f"{max(100, cpu_m // k)}m", "memory": "256Mi"}, phase="Running", node_name=f"node-{i:05d}"))
        used = {"cpu": f"{sum(int(p['requests']['cpu'][:-1]) for p in pods)}m", "memory": f"{256 * len(pods)}Mi", "pods": str(len(pods))}
        n = fx.state_node(f"node-{i:05d}", it, zone, ct, "default", used=used)
        n["pods"] = pods
        nodes.append(n)
    return {"instanceTypes": its, "nodePools": [np_], "nodes": nodes, "pendingPods": [], "wellKnownLabels": fx.KWOK_WELL_KNOWN}
