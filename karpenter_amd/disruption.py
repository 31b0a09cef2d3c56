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

Around the simulator sits the float arithmetic that turns a simulation into a decision, also here:

    EvictionCost / ReschedulingCost          utils/disruption/disruption.go:48-76
    Candidate.Price / RescheduleDisruptionCost / SavingsRatio   disruption/types.go:94-146,209-210
    computeNodePoolTotals / ScoreMove / EvaluateBalancedMove / balancedEvaluator   balanced.go:47-299
    Command.EstimatedSavings / SourceCost / PoolDisruptionCost  types.go:272-284,355-386
    validation.validateCommand (re-simulation + subset check)   validation.go:297-357

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
    """resolveNodePrice (disruption/types.go:113-127): the price of the offering matching the node's zone and
    capacity-type labels; 0 when the instance type is unknown, no offering matches, or the price is NaN."""
    it = next((t for t in cluster["instanceTypes"] if t["name"] == node["labels"].get(fx.INSTANCE_TYPE)), None)
    if it is None:
        return 0.0
    for o in it["offerings"]:
        zone = next(r["values"][0] for r in o["requirements"] if r["key"] == fx.ZONE)
        if zone == node["labels"].get(fx.ZONE) and _capacity_type(o) == node["labels"].get(fx.CAPACITY_TYPE):
            return 0.0 if math.isnan(o["price"]) else o["price"]
    return 0.0


POD_DELETION_COST = "controller.kubernetes.io/pod-deletion-cost"
PER_NODE_BASE_DISRUPTION_COST = 1.0   # types.go:132
BALANCED_K = 2                        # apis/v1/nodepool.go:172
BALANCED = "Balanced"


def eviction_cost(pod):
    """EvictionCost (utils/disruption/disruption.go:48-70): 1 + deletion-cost/2^27 + priority/2^25, clamped to [-10, 10].
    An unparsable annotation is ignored (the reference logs and carries on)."""
    cost = 1.0
    raw = pod.get("annotations", {}).get(POD_DELETION_COST)
    if raw is not None:
        try:
            cost += float(raw) / 2.0 ** 27
        except ValueError:
            pass
    if pod.get("priority") is not None:
        cost += float(pod["priority"]) / 2.0 ** 25
    return min(10.0, max(-10.0, cost))


def reschedule_disruption_cost(pods):
    """computeRescheduleDisruptionCost (types.go:137-143): per-node base + the positive eviction costs."""
    cost = PER_NODE_BASE_DISRUPTION_COST
    for p in pods:
        cost += max(0.0, eviction_cost(p))
    return cost


def savings_ratio(cluster, node):
    """Candidate.SavingsRatio = Price / RescheduleDisruptionCost (types.go:146)."""
    return candidate_price(cluster, node) / reschedule_disruption_cost(node.get("pods", []))


def _pool_name(node):
    return node["labels"].get(fx.NODEPOOL)


def _pool(cluster, name):
    return next((np_ for np_ in cluster["nodePools"] if np_["name"] == name), None)


def _is_balanced(cluster, pool_name):
    np_ = _pool(cluster, pool_name)
    return bool(np_) and np_.get("consolidationPolicy") == BALANCED


class ScoreResult:
    """ScoreResult (types.go:94-111): approved when (savings/total cost) / (disruption/total disruption) >= 1/k."""

    def __init__(self, savings_fraction=0.0, disruption_fraction=0.0, k=BALANCED_K):
        self.savings_fraction, self.disruption_fraction, self.k = savings_fraction, disruption_fraction, k

    def score(self):
        if self.savings_fraction <= 0:
            return 0.0
        if self.disruption_fraction == 0:
            return math.inf
        return self.savings_fraction / self.disruption_fraction

    def threshold(self):
        return 1.0 / float(self.k)

    def approved(self):
        return self.score() >= self.threshold()


def score_move(savings, disruption_cost, totals, k=BALANCED_K):
    """ScoreMove (balanced.go:105-121). `totals` = {"totalCost", "totalDisruptionCost"}."""
    if totals.get("totalCost", 0.0) <= 0 or totals.get("totalDisruptionCost", 0.0) <= 0:
        return ScoreResult(k=k)
    return ScoreResult(savings / totals["totalCost"], disruption_cost / totals["totalDisruptionCost"], k)


def compute_nodepool_totals(cluster, candidates, cluster_cost=None):
    """computeNodePoolTotals (balanced.go:47-103). Cost: the pool's tracked cluster cost when positive, else the sum of
    its candidates' prices. Disruption: every node of the pool (candidate or not) contributes its
    RescheduleDisruptionCost — "non-candidate nodes still contribute to the denominators"."""
    fallback = {}
    for c in candidates:
        fallback[_pool_name(c)] = fallback.get(_pool_name(c), 0.0) + candidate_price(cluster, c)
    disruption = {}
    for n in cluster["nodes"]:
        name = _pool_name(n)
        if name is None:
            continue
        disruption[name] = disruption.get(name, 0.0) + reschedule_disruption_cost(n.get("pods", []))
    totals = {}
    for name, cost in fallback.items():
        cc = (cluster_cost or {}).get(name, 0.0)
        totals[name] = {"totalCost": cc if cc > 0 else cost, "totalDisruptionCost": disruption.get(name, 0.0)}
    return totals


def estimated_savings(cluster, candidates, cmd):
    """Command.EstimatedSavings (types.go:364-386): source price, minus — for a replace — the cheapest available
    offering compatible with the claim's requirements of each new claim's FIRST instance type option."""
    source = sum(candidate_price(cluster, c) for c in candidates)
    if cmd["decision"] != REPLACE:
        return source
    by_name = {t["name"]: t for t in cluster["instanceTypes"]}
    dest = 0.0
    for i, claim in enumerate(cmd["results"]["newNodeClaims"]):
        # the command's claim was put in OrderByPrice order before the price filter (consolidation.go:209), so option 0
        # is the option with the lowest cheapest-compatible-available price; ties carry the same price
        options = cmd["replacement"] if i == 0 else claim["instanceTypes"]
        reqs = {r["key"]: r for r in claim["requirements"]}
        best = math.inf
        for n in options:
            best = min([best] + [o["price"] for o in by_name[n]["offerings"] if o.get("available", True) and _offering_compatible(reqs, o)])
        if best < math.inf:
            dest += best
    return source - dest


def evaluate_balanced_move(cluster, candidates, cmd, totals):
    """EvaluateBalancedMove (balanced.go:131-183) → (approved, {pool: ScoreResult}). Every Balanced pool among the
    command's candidates must approve; other pools are skipped. Cross-pool moves split the net savings by each pool's
    share of the source cost."""
    if not candidates:
        return False, None
    by_pool = {}
    for c in candidates:
        by_pool.setdefault(_pool_name(c), []).append(c)
    savings = estimated_savings(cluster, candidates, cmd)
    total_cost = sum(candidate_price(cluster, c) for c in candidates)
    approved, per_pool = True, {}
    for name, members in by_pool.items():
        if not _is_balanced(cluster, name):
            continue
        disruption = sum(reschedule_disruption_cost(c.get("pods", [])) for c in members)
        pool_savings = savings
        if total_cost > 0 and len(by_pool) > 1:
            pool_savings = savings * (sum(candidate_price(cluster, c) for c in members) / total_cost)
        r = score_move(pool_savings, disruption, totals.get(name, {}), BALANCED_K)
        per_pool[name] = r
        approved = approved and r.approved()
    return approved, per_pool


class BalancedEvaluator:
    """balancedEvaluator (balanced.go:207-299) without the metrics/events. `NoopEvaluator` is what consolidation uses
    until SetNodePoolTotals is called (consolidation.go:61-85)."""

    def __init__(self, cluster, totals):
        self.cluster, self.totals = cluster, totals

    def approve_command(self, candidates, cmd):
        return evaluate_balanced_move(self.cluster, candidates, cmd, self.totals)

    def can_pass_threshold(self, candidate):
        """A DELETE saves the whole node price: if even that cannot pass, no REPLACE will (balanced.go:284-299)."""
        name = _pool_name(candidate)
        if not _is_balanced(self.cluster, name):
            return True
        t = self.totals.get(name)
        if not t or t["totalCost"] <= 0:
            return True
        return score_move(candidate_price(self.cluster, candidate), reschedule_disruption_cost(candidate.get("pods", [])), t, BALANCED_K).approved()


class NoopEvaluator:
    def approve_command(self, candidates, cmd):
        return True, None

    def can_pass_threshold(self, candidate):
        return True


def sort_candidates(cluster, nodes):
    """consolidation.sortCandidates — consolidation.go:149-154 (descending ratio; name breaks ties deterministically)."""
    return sorted(nodes, key=lambda n: (-savings_ratio(cluster, n), n["name"]))


def interweave_by_nodepool(candidates, previously_unseen=()):
    """SingleNodeConsolidation.SortCandidates' second step (singlenodeconsolidation.go:141-172): the ratio-sorted list is
    dealt out round-robin over the NodePools, so that one pool's long list cannot starve the others within the time
    budget; pools that a timed-out earlier run never reached go first. (Among the remaining pools the reference follows
    Go's map order, i.e. any; here: order of first appearance in the sorted list.)"""
    by_pool, order = {}, [p for p in previously_unseen]
    for c in candidates:
        name = _pool_name(c)
        by_pool.setdefault(name, []).append(c)
        if name not in order:
            order.append(name)
    order = [p for p in order if p in by_pool]
    out = []
    for i in range(max((len(v) for v in by_pool.values()), default=0)):
        for name in order:
            if i < len(by_pool[name]):
                out.append(by_pool[name][i])
    return out


MAX_INSTANCE_TYPES = 600   # scheduling.MaxInstanceTypes, nodeclaimtemplate.go:50
PARTITION = "example.com/partition"   # make_resident_cluster: the custom label that splits its dedicated NodePool


def _finish_simulation(cluster, res, deleting_uids):
    """helpers.go:133-153 on the Results of one simulation: pods that landed on an uninitialized node make the decision
    unsafe; AllNonPendingPodsScheduled (scheduler.go:388-392)."""
    errors = dict(res["podErrors"])
    for en in res["existingNodes"]:
        if not en.get("initialized", True):
            for uid in en["pods"]:
                if uid not in deleting_uids:
                    errors[uid] = {"code": 100, "diag": 0}  # UninitializedNodeError
    pending_uids = {p["uid"] for p in cluster.get("pendingPods", [])}
    res = dict(res)
    res["podErrors"] = errors
    res["allNonPendingPodsScheduled"] = not [u for u in errors if u not in pending_uids]
    return res


def _has_topology(pod):
    return bool(pod.get("topologySpreadConstraints") or pod.get("podAffinity") or pod.get("podAntiAffinity"))


class ResidentCluster:
    """The cluster uploaded ONCE (ksolve_create), every simulation a probe of it (ksolve_probe_create: removed-node bitmap
    + displaced-pod rows) — SimulateScheduling (helpers.go:53-155) without assembling, flattening or uploading a problem
    per candidate set. Passed wherever this module takes a `solver`: simulate_scheduling() asks it for the Results.

    `prefetch(candidate_sets)` sends all those simulations to the device as ONE batched launch (one wavefront per probe)
    and keeps their Results; single-node consolidation prefetches every candidate, the multi-node binary search every
    prefix it can reach. Clusters whose pods carry topology constraints are probed too: every bound pod becomes a pod row,
    the device counts the whole cluster once and a probe takes its candidates' share out of its own counters."""

    def __init__(self, cluster, candidates, solver_lib=None):
        from .scheduling import NewScheduler, Unsupported
        self.cluster = cluster
        deleting = [n for n in cluster["nodes"] if n.get("markedForDeletion")]
        self._deleting_pods = [p for n in deleting for p in n.get("pods", [])]
        self._deleting_uids = {p["uid"] for p in self._deleting_pods}
        self._always = list(cluster.get("pendingPods", [])) + self._deleting_pods          # part of every simulation
        displaced = [p for c in candidates if not c.get("markedForDeletion") for p in c.get("pods", [])]
        # With topology constraints anywhere in the cluster, EVERY bound pod is a row of the base problem: the device counts the
        # whole cluster once (options.residentCluster) and a probe takes its displaced pods' share out again (countDomains
        # without the pods being scheduled, topology.go:92-94, :361-459)
        topo = any(_has_topology(p) for n in cluster["nodes"] for p in n.get("pods", [])) or any(_has_topology(p) for p in self._always)
        if topo:
            displaced = [p for n in cluster["nodes"] if not n.get("markedForDeletion") for p in n.get("pods", [])]
        state_nodes = [{k: v for k, v in n.items() if k != "pods"} for n in cluster["nodes"] if not n.get("markedForDeletion")]
        prob = fx.problem(cluster["instanceTypes"], cluster["nodePools"], copy.deepcopy(self._always + displaced),
                          well_known=cluster.get("wellKnownLabels", fx.KWOK_WELL_KNOWN), state_nodes=state_nodes,
                          options=dict(cluster.get("options", {}), consolidationSimulation=True, truncateInstanceTypes=MAX_INSTANCE_TYPES, residentCluster=True),
                          namespaces=cluster.get("namespaces"), deleting_node_names=[n["name"] for n in deleting])
        self.scheduler = NewScheduler(prob, solver_lib)
        self._cache = {}

    @classmethod
    def from_compact(cls, cc, solver_lib=None):
        """The resident cluster of a compact cluster (make_resident_cluster): every bound pod is a pod row of the base problem,
        so any node can be a candidate. Only decisions() is available (no per-probe pod lists on the Python side)."""
        from .scheduling import NewScheduler
        self = object.__new__(cls)
        self.cluster = cc
        self._deleting_pods, self._deleting_uids, self._always, self._cache = [], set(), list(cc.get("pendingPods", [])), {}
        self.scheduler = NewScheduler(compact_problem(cc, pods=copy.deepcopy(self._always)), solver_lib)
        self._position = {n["name"]: i for i, n in enumerate(cc["nodes"])}      # candidates travel as positions in the stateNodes list
        return self

    def _key(self, candidates):
        return tuple(sorted(c["name"] for c in candidates))

    def _probe(self, candidates):
        live = [c for c in candidates if not c.get("markedForDeletion")]
        pods = [p["uid"] for p in self._always] + [p["uid"] for c in live for p in c.get("pods", [])]
        return self.scheduler.Probe([c["name"] for c in live], pods)

    def prefetch(self, candidate_sets):
        from .scheduling import SolveBatch
        todo = [cs for cs in candidate_sets if self._key(cs) not in self._cache]
        if not todo:
            return
        probes = [self._probe(cs) for cs in todo]
        for cs, res, pr in zip(todo, SolveBatch(probes), probes):
            self._cache[self._key(cs)] = _finish_simulation(self.cluster, res, self._deleting_uids)
            pr.close()

    def simulate(self, candidates):
        self.prefetch([candidates])
        return self._cache[self._key(candidates)]

    def decisions(self, candidate_sets, detail=False, multi_node=False, library_prices=False, arrays=False, replicas=()):
        """computeConsolidation (consolidation.go:159-256) for every candidate set in ONE device launch, verdicts included
        (Scheduler.Sweep / ksolve_sweep): [{"decision", "candidates", "replacement", "replacementCapacityType"}], the commands
        compute_consolidation() returns without their Results. Descriptors and verdicts are computed by the host library.
        multi_node: the sets are prefixes of MultiNodeConsolidation's search — a REPLACE over several candidates goes through
        filterOutSameInstanceType (multinodeconsolidation.go:209-246) and comes back as NOOP when it does not stand.
        library_prices: the candidates' prices and capacity types are taken from the host library's node table instead of
        being summed here. arrays: the binary form of the call (Scheduler.SweepArrays / ksched_sweep_arrays: no JSON on either side
        of the host library, what a cgo caller does; the library's prices; no `reason` texts). replicas: further ResidentClusters
        of the same cluster document — on other devices, or further contexts on this one (each handle has its own stream, staging
        memory and arena, so one share's upload / finalize / download runs beside another share's kernel): the probes are dealt
        out over all of them inside the call (ksolve_sweep_replicas)."""
        cluster = self.cluster
        live = [[c for c in cs if not c.get("markedForDeletion")] for cs in candidate_sets]
        prices = None if library_prices else [sum(self._price(c) for c in cs) for cs in candidate_sets]
        all_spot = None if library_prices else [all(c["labels"][fx.CAPACITY_TYPE] == "spot" for c in cs) for cs in candidate_sets]
        pos = getattr(self, "_position", None)
        names = [[pos[c["name"]] for c in cs] for cs in live] if pos is not None else [[c["name"] for c in cs] for cs in live]
        if arrays:
            if pos is None or detail:
                raise ValueError("the binary sweep takes node positions (ResidentCluster.from_compact) and has no detail form")
            if not hasattr(self, "_it_names"):
                self._it_names = [t["name"] for t in cluster["instanceTypes"]]
            out = self.scheduler.SweepArrays(names, multi_node=multi_node, replicas=tuple(r.scheduler for r in replicas), instance_type_names=self._it_names)
        else:
            out = self.scheduler.Sweep(names, prices, all_spot, detail=detail, multi_node=multi_node, replicas=tuple(r.scheduler for r in replicas))
        repl = {r["probe"]: r for r in out["replacements"]}
        cmds = []
        for i, cs in enumerate(candidate_sets):
            d = out["decisions"][i]
            if d == 3:   # spot-to-spot behind its feature gate: the per-probe path has the whole ordered list
                cmd = dict(compute_consolidation(cluster, cs, self))
                if multi_node and cmd["decision"] == REPLACE and len(cs) > 1 and not filter_out_same_instance_type(cluster, cs, cmd):
                    cmd = {"decision": NOOP, "candidates": [], "replacement": None}
                cmd.pop("results", None)
                cmds.append(cmd)
                continue
            cmd = {"decision": (NOOP, DELETE, REPLACE)[d], "candidates": [c["name"] for c in cs], "replacement": None}
            if d == 2:
                cmd["replacement"] = repl[i]["instanceTypes"]
                cmd["replacementCapacityType"] = repl[i]["capacityType"]
            if str(i) in out["reasons"]:
                cmd["reason"] = out["reasons"][str(i)]
            cmds.append(cmd)
        self.last_sweep = out
        return cmds

    def first_n(self, candidates, max_n=100, evaluator=None):
        """MultiNodeConsolidation.firstNConsolidationOption (multinodeconsolidation.go:117-207) with every prefix the binary
        search can reach simulated in ONE launch (the search itself is a chain of dependent simulations — about seven for
        100 candidates; all 99 prefixes at once cost one sweep and the walk over their verdicts is a lookup).
        Returns (command, probe sequence) like first_n_consolidation_option."""
        if len(candidates) < 2:
            return {"decision": NOOP, "candidates": []}, []
        lo, hi = first_n_bounds(len(candidates), max_n)
        sizes = list(range(lo + 1, hi + 2))
        cmds = self.decisions([candidates[:k] for k in sizes], multi_node=True)
        by_size = dict(zip(sizes, cmds))
        return first_n_from_commands(len(candidates), lambda k: by_size[k], max_n, evaluator, candidates)

    def _price(self, node):
        key = (node["labels"].get(fx.INSTANCE_TYPE), node["labels"].get(fx.ZONE), node["labels"].get(fx.CAPACITY_TYPE))
        cache = self.__dict__.setdefault("_price_cache", {})
        if key not in cache:
            cache[key] = candidate_price(self.cluster, node)
        return cache[key]

    def close(self):
        self.scheduler.close()


# ---------------------------------------------------------------------------------------------------------------
# BASELINE configs[4]: a cluster of 100k nodes / 2M bound pods does not fit a list of pod dicts. The COMPACT form keeps the
# state nodes as dicts (without their pods) and the bound pods as pod groups — {count, uidSeed, template, nodeIndex}: pod i of
# the group runs on nodes[nodeIndex[i]] — which is also how the host library takes them (podGroups[].nodeIndexB64).
# ---------------------------------------------------------------------------------------------------------------
def make_resident_cluster(n_nodes=100_000, seed=42, n_types=144, dedicated_fraction=0.3, scale_down=0.3, topology=False):
    """A synthetic under-utilised cluster in compact form, shaped like test/suites/performance/basic_test.go:61-68 (scale the
    workload out, let the provisioner pack nodes, scale it down by 30%): every node is packed with pods of the benchmark's
    cpu x memory grid (scheduling_benchmark_test.go:447-455) until its instance type is full, then pods are removed.
    Two NodePools so that a single-node sweep meets every verdict of computeConsolidation (consolidation.go:159-256):
      default    ~70% of the nodes, every node loses ~scale_down of its pods: a candidate's pods fit the free room of the
                 others (delete);
      dedicated  tainted dedicated=batch:NoSchedule and split into partitions (a custom label the pool admits; pods are pinned
                 to their partition by node selector + toleration). In 80% of the partitions the nodes stayed full but for a
                 few that kept a quarter of their pods: a full candidate's pods find room for some of them at most and need
                 a NodeClaim — a smaller instance type than the one they sit on (replace), the same type again (nothing to
                 do), or they all fit (delete).
    A few nodes are not initialized or are under consolidateAfter (helpers.go:133-153, scheduler.go:628).
    topology=True: two fifths of the default pool's pod templates carry a spread constraint — zonal (maxSkew 2) over one of three
    app labels, or per hostname (maxSkew 8) — so that every probe re-derives domain counts without its candidates.
    Returns {"instanceTypes", "nodePools", "wellKnownLabels", "nodes", "podGroups", "pendingPods", "nodePodCount"}."""
    import base64
    import numpy as np
    rng = np.random.default_rng(seed)
    its = fx.kwok_catalog(n_types)
    sizes = [t for t in its if "linux" in t["name"] and "amd64" in t["name"] and int(t["capacity"]["cpu"]) in (8, 16, 32, 48) and t["name"].startswith(("c-", "s-", "m-"))]
    n_part = max(4, min(192, n_nodes // 400))
    parts = [f"p{i:03d}" for i in range(n_part)]
    pools = [fx.node_pool("dedicated", weight=10, taints=[{"key": "dedicated", "value": "batch", "effect": "NoSchedule"}],
                          requirements=[fx.req(PARTITION, "In", *parts)]),
             fx.node_pool("default", weight=0)]
    for np_ in pools:
        np_["nodeClassLabelKey"] = "karpenter.kwok.sh/kwoknodeclass"
    # the default pool runs the benchmark's small pods; the dedicated pool's batch pods are large — the crumbs a packed node has
    # left (less than the pod that did not fit) are no home for them
    small = [(c, m) for c in fx.BENCH_CPU_M for m in fx.BENCH_MEM_MI]
    combos = small + [(c, m) for c in (4000, 6000) for m in (4096, 8192, 16384)]
    ccpu = np.array([c for c, _ in combos], dtype=np.int64)                    # milli
    cmem = np.array([m for _, m in combos], dtype=np.int64)                    # Mi
    n_ded = int(n_nodes * dedicated_fraction)
    pool_of = np.zeros(n_nodes, dtype=np.int8)                                 # 1 = dedicated
    pool_of[rng.choice(n_nodes, n_ded, replace=False)] = 1
    type_of = rng.integers(0, len(sizes), n_nodes)
    zone_of = rng.integers(0, len(fx.KWOK_ZONES), n_nodes)
    spot = rng.random(n_nodes) < 0.35
    alloc_cpu = np.array([int(t["capacity"]["cpu"]) * 1000 - 100 for t in sizes], dtype=np.int64)[type_of]
    alloc_mem = np.array([int(t["capacity"]["memory"][:-2]) * 1024 - 10 for t in sizes], dtype=np.int64)[type_of]   # Mi
    alloc_pods = np.array([int(t["capacity"]["pods"]) for t in sizes], dtype=np.int64)[type_of]
    # pack: 96 random pods per node, keep the prefix that fits (what first-fit packing leaves on a node: full)
    part_of = rng.integers(0, n_part, n_nodes)
    tight = rng.random(n_part) < 0.8                                           # partitions whose nodes stayed full but for a few
    p_small = min(0.5, 0.7 * n_part / max(1, n_ded))                          # about one scaled-down node per tight partition
    keep_ded = np.where(tight[part_of], np.where(rng.random(n_nodes) < 1.0 - p_small, 1.0, 0.25), 1.0 - scale_down)
    keep_p = np.where(pool_of == 0, 1.0 - scale_down, keep_ded)
    draw = np.empty((n_nodes, 96), dtype=np.int8)
    kept = np.empty((n_nodes, 96), dtype=bool)
    used_cpu = np.empty(n_nodes, dtype=np.int64); used_mem = np.empty(n_nodes, dtype=np.int64)
    for lo in range(0, n_nodes, 8192):                                         # in slices: the temporaries stay small
        hi = min(n_nodes, lo + 8192)
        d = np.where(pool_of[lo:hi, None] == 1, rng.integers(len(small), len(combos), (hi - lo, 96)), rng.integers(0, len(small), (hi - lo, 96)))
        fits = (np.cumsum(ccpu[d], axis=1) <= alloc_cpu[lo:hi, None]) & (np.cumsum(cmem[d], axis=1) <= alloc_mem[lo:hi, None]) & (np.arange(1, 97)[None, :] <= alloc_pods[lo:hi, None])
        k = fits.cumprod(axis=1).astype(bool) & (rng.random((hi - lo, 96)) < keep_p[lo:hi, None])   # packed, then scaled down
        draw[lo:hi] = d; kept[lo:hi] = k
        used_cpu[lo:hi] = (ccpu[d] * k).sum(axis=1); used_mem[lo:hi] = (cmem[d] * k).sum(axis=1)
    n_kept = kept.sum(axis=1)
    uninit = rng.random(n_nodes) < 0.002
    uca = rng.random(n_nodes) < 0.01
    label_cache = {}
    nodes = []
    import gc
    gc_was = gc.isenabled()
    gc.disable()                                                               # 100k dicts of dicts: no cycles to find, a lot of generations to scan
    type_of, zone_of, spot_l, pool_l, uninit_l, part_l = type_of.tolist(), zone_of.tolist(), spot.tolist(), pool_of.tolist(), uninit.tolist(), part_of.tolist()
    free_cpu, free_mem, free_pods, uca_l = (alloc_cpu - used_cpu).tolist(), (alloc_mem - used_mem).tolist(), (alloc_pods - n_kept).tolist(), uca.tolist()
    for i in range(n_nodes):
        key = (type_of[i], zone_of[i], spot_l[i], pool_l[i], uninit_l[i])
        base = label_cache.get(key)
        if base is None:
            nd = fx.state_node("x", sizes[type_of[i]], fx.KWOK_ZONES[zone_of[i]], "spot" if spot_l[i] else "on-demand", "dedicated" if pool_l[i] else "default",
                               taints=pools[0]["taints"] if pool_l[i] else None, initialized=not uninit_l[i])
            base = label_cache[key] = nd
        name = f"node-{i:06d}"
        labels = dict(base["labels"]); labels[fx.HOSTNAME] = name
        if pool_l[i]:
            labels[PARTITION] = parts[part_l[i]]
        avail = dict(base["available"])
        avail["cpu"] = f"{free_cpu[i] * 1_000_000}n"
        avail["memory"] = f"{free_mem[i] * 1048576 * 10**9}n"
        avail["pods"] = f"{free_pods[i] * 10**9}n"
        nodes.append({"name": name, "labels": labels, "taints": base["taints"], "available": avail, "capacity": base["capacity"], "initialized": base["initialized"],
                      "managed": True, "underConsolidateAfter": uca_l[i]})
    if gc_was:
        gc.enable()
    # bound pods: one group per (pool, cpu, memory)
    groups = []
    node_idx, slot_idx = np.nonzero(kept)
    combo_idx = draw[node_idx, slot_idx]
    pod_part = np.where(pool_of[node_idx] == 1, part_of[node_idx], -1)         # -1: the default pool
    gkey = (pod_part + 1) * len(combos) + combo_idx                            # one group per (partition, cpu, memory)
    by_group = np.argsort(gkey, kind="stable")
    bounds = np.searchsorted(gkey[by_group], np.arange((n_part + 1) * len(combos) + 1))
    gi = 0
    for part in range(-1, n_part):
        for ci, (c, m) in enumerate(combos):
            g_ = (part + 1) * len(combos) + ci
            sel = node_idx[by_group[bounds[g_]:bounds[g_ + 1]]]
            gi += 1
            if not len(sel):
                continue
            kw = dict(requests={"cpu": f"{c}m", "memory": f"{m}Mi"}, phase="Running")
            if topology and part < 0 and ci % 5 == 0:
                lab = {"app": f"zonal-{ci % 3}"}
                kw.update(labels=lab, topology_spread=[fx.spread(fx.ZONE, lab, max_skew=2)])
            elif topology and part < 0 and ci % 5 == 1:
                lab = {"app": f"host-{ci % 2}"}
                kw.update(labels=lab, topology_spread=[fx.spread(fx.HOSTNAME, lab, max_skew=8)])
            if part >= 0:
                kw.update(node_selector={PARTITION: parts[part]}, tolerations=[{"key": "dedicated", "operator": "Exists", "effect": "NoSchedule"}])
            groups.append({"count": int(len(sel)), "uidSeed": seed * 100003 + gi, "template": fx.pod(uid="t", **kw),
                           "nodeIndexB64": base64.b64encode(sel.astype("<i4").tobytes()).decode(), "_nodeIndex": sel.astype(np.int32)})
    return {"instanceTypes": its, "nodePools": pools, "wellKnownLabels": fx.KWOK_WELL_KNOWN, "nodes": nodes, "podGroups": groups, "pendingPods": [],
            "nodePodCount": n_kept.astype(int).tolist()}


def compact_node_pods(cc, node_index):
    """The pods bound to node `node_index` of a compact cluster as pod dicts (uids as the host library derives them)."""
    import numpy as np
    out = []
    for g in cc["podGroups"]:
        for pos in np.nonzero(g["_nodeIndex"] == node_index)[0]:
            out.append(dict(g["template"], uid=fx.group_pod_uid(g["uidSeed"], int(pos)), nodeName=cc["nodes"][node_index]["name"]))
    return out


def compact_candidates(cc):
    """sortCandidates (consolidation.go:149-154) over a compact cluster: descending price / disruption cost, names break ties;
    returns node indices. Every bound pod costs 1 (no deletion-cost annotations or priorities in the synthetic cluster)."""
    price_cache = {}
    keyed = []
    for i, n in enumerate(cc["nodes"]):
        k = (n["labels"].get(fx.INSTANCE_TYPE), n["labels"].get(fx.ZONE), n["labels"].get(fx.CAPACITY_TYPE))
        if k not in price_cache:
            price_cache[k] = candidate_price(cc, n)
        keyed.append((-(price_cache[k] / (PER_NODE_BASE_DISRUPTION_COST + cc["nodePodCount"][i])), n["name"], i))
    keyed.sort()
    return [i for _, _, i in keyed]


def compact_cluster_pods(cc):
    """Every bound pod of a compact cluster as a pod dict (uid, nodeName): the clusterPods of a problem assembled for one
    simulation — what seeds the topology counts there (topology.go:361-459)."""
    out = []
    for g in cc["podGroups"]:
        names = [cc["nodes"][i]["name"] for i in g["_nodeIndex"].tolist()]
        for pos, nm in enumerate(names):
            out.append(dict(g["template"], uid=fx.group_pod_uid(g["uidSeed"], pos), nodeName=nm))
    return out


def compact_problem(cc, pods=None, pod_groups=None, strip=True):
    groups = [{k: v for k, v in g.items() if not k.startswith("_")} for g in (cc["podGroups"] if pod_groups is None else pod_groups)]
    # maxClaims: the base handle of a resident cluster is never solved itself; a probe may create this many NodeClaims
    return fx.problem(cc["instanceTypes"], cc["nodePools"], pods or [], pod_groups=groups, well_known=cc["wellKnownLabels"], state_nodes=cc["nodes"],
                      options=dict(cc.get("options", {}), consolidationSimulation=True, truncateInstanceTypes=MAX_INSTANCE_TYPES, maxClaims=cc.get("maxClaimsPerProbe", 2048), residentCluster=True))


def decide(cluster, candidates, res):
    """computeConsolidation (consolidation.go:159-256) from the Results of the simulation."""
    return compute_consolidation(cluster, candidates, lambda _prob: None, results=res)


def simulate_scheduling(cluster, candidates, solver):
    """helpers.go:53-155: Solve() with the candidates removed and their pods added to the pending set."""
    if isinstance(solver, ResidentCluster):
        return solver.simulate(candidates)
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
                      state_nodes=state_nodes, cluster_pods=copy.deepcopy(staying), options=dict(cluster.get("options", {}), consolidationSimulation=True, truncateInstanceTypes=MAX_INSTANCE_TYPES),   # results.TruncateInstanceTypes(ctx, scheduling.MaxInstanceTypes): price order, capped at 600, minValues re-checked after the cut (helpers.go:131, scheduler.go:419-437)
                      namespaces=cluster.get("namespaces"),
                      deleting_node_names=[n["name"] for n in deleting])
    res = solver(prob)
    return _finish_simulation(cluster, res, {p["uid"] for p in deleting_pods})


MIN_INSTANCE_TYPES_FOR_SPOT_TO_SPOT = 15      # consolidation.go:46


def _min_types_for_min_values(names, reqs, by_name):
    """InstanceTypes.SatisfiesMinValues (types.go:399-433): the length of the shortest prefix of `names` whose instance
    types carry at least minValues distinct values for every key that asks for them, and whether the whole list does."""
    wanted = {k: q["minValues"] for k, q in reqs.items() if q.get("minValues")}
    if not wanted:
        return 0, True
    seen = {k: set() for k in wanted}
    for i, n in enumerate(names):
        for r in by_name[n]["requirements"]:
            if r["key"] in wanted:       # Requirement.Values(): the stored list whatever the operator (NotIn: the excluded values), unfiltered
                seen[r["key"]].update(r.get("values") or [])
        if all(len(seen[k]) >= wanted[k] for k in wanted):
            return i + 1, True
    return len(names), False


def _spot_to_spot(cluster, candidates, cmd, claim, reqs, price, by_name):
    """computeSpotToSpotConsolidation (consolidation.go:261-342). The claim's instance types arrive in OrderByPrice order
    (simulate_scheduling asks Solve() for it): pin the claim to spot, keep the types with a compatible available offering,
    keep the cheaper ones; several candidates may then be replaced by whatever is left, a single candidate only if at
    least 15 cheaper types remain, of which the 15 cheapest (more if minValues needs more) are launched — so that the
    node that comes up is among them and is not consolidated again at once."""
    if not cluster.get("options", {}).get("spotToSpotConsolidation"):
        return cmd                                                       # feature gate off (the default)
    reqs = dict(reqs)
    old = reqs.get(fx.CAPACITY_TYPE) or {}
    reqs[fx.CAPACITY_TYPE] = {"key": fx.CAPACITY_TYPE, "complement": False, "values": ["spot"], "gte": None, "lte": None, "minValues": old.get("minValues")}
    compatible = [n for n in claim["instanceTypes"] if any(o.get("available", True) and _offering_compatible(reqs, o) for o in by_name[n]["offerings"])]
    cheaper = [n for n in compatible if worst_launch_price(by_name[n], reqs) < price]
    need, ok = _min_types_for_min_values(cheaper, reqs, by_name)
    if not cheaper or not ok:
        return cmd
    if len(candidates) == 1:
        if len(cheaper) < MIN_INSTANCE_TYPES_FOR_SPOT_TO_SPOT:
            cmd["reason"] = f"SpotToSpotConsolidation requires {MIN_INSTANCE_TYPES_FOR_SPOT_TO_SPOT} cheaper instance type options than the current candidate to consolidate, got {len(cheaper)}"
            return cmd
        cheaper = cheaper[: max(MIN_INSTANCE_TYPES_FOR_SPOT_TO_SPOT, need)]
    cmd["decision"] = REPLACE
    cmd["replacement"] = sorted(cheaper)
    cmd["replacementInPriceOrder"] = cheaper
    cmd["replacementCapacityType"] = "spot"
    return cmd


def compute_consolidation(cluster, candidates, solver, results=None):
    """consolidation.go:159-256 → {"decision", "candidates", "replacement": instance type names}. `results`: the Results of the
    simulation when the caller already has them (finished by _finish_simulation)."""
    res = simulate_scheduling(cluster, candidates, solver) if results is None else results
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
    by_name = {t["name"]: t for t in cluster["instanceTypes"]}
    if all_spot and (ct is None or _req_has(ct, "spot")):
        return _spot_to_spot(cluster, candidates, cmd, claim, reqs, price, by_name)
    cheaper = [n for n in claim["instanceTypes"] if worst_launch_price(by_name[n], reqs) < price]   # nodeclaim.go:411-420
    if not _min_types_for_min_values(cheaper, reqs, by_name)[1]:
        cmd["reason"] = "minValues requirement is not met after filtering by price"                # SatisfiesMinValues, nodeclaim.go:416-418
        return cmd
    if not cheaper:
        return cmd
    cmd["decision"] = REPLACE
    cmd["replacement"] = sorted(cheaper)
    cmd["replacementCapacityType"] = "spot" if (ct is None or (_req_has(ct, "spot") and _req_has(ct, "on-demand"))) else None  # consolidation.go:238-243
    return cmd


def filter_out_same_instance_type(cluster, candidates, cmd):
    """filterOutSameInstanceType (multinodeconsolidation.go:209-246) on a REPLACE command of several candidates: when a replacement
    option is one of the instance types being removed, every option must be cheaper than the cheapest candidate of that type
    (else deleting the others is the better command). Narrows cmd["replacement"] in place; returns whether the command stands."""
    existing = {}
    for c in candidates:
        n = c["labels"][fx.INSTANCE_TYPE]
        existing[n] = min(existing.get(n, math.inf), candidate_price(cluster, c))
    max_price = min([existing[n] for n in cmd["replacement"] if n in existing], default=math.inf)
    reqs = {r["key"]: r for r in cmd["results"]["newNodeClaims"][0]["requirements"]}
    if cmd.get("replacementCapacityType") == "spot":
        # the Replacement IS the NodeClaim computeConsolidation narrowed (types.go:224-226 wraps the pointer): when it pinned the
        # claim to spot (consolidation.go:238-243, :272) this filter prices every option by its spot offerings — a type without
        # one has WorstLaunchPrice MaxFloat64 and never passes the strict '<', even when max_price is MaxFloat64 itself
        old = reqs.get(fx.CAPACITY_TYPE) or {}
        reqs[fx.CAPACITY_TYPE] = {"key": fx.CAPACITY_TYPE, "complement": False, "values": ["spot"], "gte": None, "lte": None, "minValues": old.get("minValues")}
    by_name = {t["name"]: t for t in cluster["instanceTypes"]}
    cmd["replacement"] = [n for n in cmd["replacement"] if worst_launch_price(by_name[n], reqs) < max_price]
    # RemoveInstanceTypeOptionsByPriceAndMinValues (nodeclaim.go:411-420): what survives the price filter must still meet minValues
    return bool(cmd["replacement"]) and _min_types_for_min_values(cmd["replacement"], reqs, by_name)[1]


def first_n_bounds(n_candidates, max_n=100):
    """The binary search's first window (multinodeconsolidation.go:117-126): prefix sizes lo+1 .. hi+1."""
    return 1, (n_candidates - 1 if n_candidates <= max_n else max_n)


def first_n_from_commands(n_candidates, command_of_prefix, max_n=100, evaluator=None, candidates=None):
    """firstNConsolidationOption's walk (multinodeconsolidation.go:117-207) over commands that are already computed:
    command_of_prefix(k) = the command for candidates[:k] with filterOutSameInstanceType applied (decision NOOP when it does not
    stand). Same probe sequence and result as first_n_consolidation_option."""
    evaluator = evaluator or NoopEvaluator()
    if n_candidates < 2:
        return {"decision": NOOP, "candidates": []}, []
    lo, hi = first_n_bounds(n_candidates, max_n)
    last, probes = {"decision": NOOP, "candidates": []}, []
    while lo <= hi:
        mid = (lo + hi) // 2
        cmd = command_of_prefix(mid + 1)
        probes.append((mid + 1, cmd["decision"]))
        valid = cmd["decision"] in (DELETE, REPLACE)
        if valid and candidates is not None:
            approved, per_pool = evaluator.approve_command(candidates[: mid + 1], cmd)
            cmd["scores"] = per_pool
            valid = approved
        if valid:
            last, lo = cmd, mid + 1
        else:
            hi = mid - 1
    return last, probes


def first_n_consolidation_option(cluster, candidates, solver, max_n=100, evaluator=None):
    """multinodeconsolidation.go:117-207: binary search for the longest prefix that consolidates; same probe sequence.
    A valid decision is then scored by the evaluator (:166-174): Balanced pools may reject it, which shrinks the window."""
    evaluator = evaluator or NoopEvaluator()
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
            valid = filter_out_same_instance_type(cluster, candidates[: mid + 1], cmd)
        if valid:
            approved, per_pool = evaluator.approve_command(candidates[: mid + 1], cmd)
            cmd["scores"] = per_pool
            valid = approved
        if valid:
            last, lo = cmd, mid + 1
        else:
            hi = mid - 1
    return last, probes


def single_node_consolidation(cluster, candidates, solver, evaluator=None):
    """singlenodeconsolidation.go:55-126: the first candidate (in sorted order) with a valid command. Candidates whose
    best case (a DELETE) cannot pass the Balanced threshold are skipped without a simulation (:88), and a computed
    command is scored before it is accepted (:101-104)."""
    evaluator = evaluator or NoopEvaluator()
    for c in candidates:
        if not evaluator.can_pass_threshold(c):
            continue
        cmd = compute_consolidation(cluster, [c], solver)
        if cmd["decision"] == NOOP:
            continue
        approved, per_pool = evaluator.approve_command([c], cmd)
        if not approved:
            continue
        cmd["scores"] = per_pool
        return cmd
    return {"decision": NOOP, "candidates": []}


def validate_command(cluster, candidates, cmd, solver):
    """validation.validateCommand (validation.go:297-357): re-simulate on the current cluster and accept only when the
    outcome still matches the command — no new claim for a delete; for a replace exactly one claim whose instance types
    are a superset of the command's. Returns None when valid, else the reference's error text."""
    if not candidates:
        return "no candidates"
    res = simulate_scheduling(cluster, candidates, solver)
    if not res["allNonPendingPodsScheduled"]:
        return "pods would not schedule"
    claims = res["newNodeClaims"]
    expecting = cmd["decision"] == REPLACE
    if len(claims) == 0:
        return None if not expecting else "scheduling simulation produced new results"
    if len(claims) > 1 or not expecting:
        return "scheduling simulation produced new results"
    if not set(cmd["replacement"]) <= set(claims[0]["instanceTypes"]):   # instanceTypesAreSubset
        return "scheduling simulation produced new results"
    return None


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


def sweep_resident(cluster, candidates, solver_lib=None):
    """Single-node consolidation sweep over a RESIDENT cluster: one upload, one probe descriptor per candidate, one
    batched launch. Same decisions as sweep(); returns (decisions, ResidentCluster) — close() it when done."""
    rc = ResidentCluster(cluster, candidates, solver_lib)
    rc.prefetch([[c] for c in candidates])
    return [compute_consolidation(cluster, [c], rc) for c in candidates], rc


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
