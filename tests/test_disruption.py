"""Consolidation simulator: decisions from the device algorithm (host emulation here, GPU in
test_gpu_parity.py) must equal the decisions derived from the oracle's Solve() on every probe."""
import pytest

import parity
from karpenter_amd import disruption as dz
from karpenter_amd import fixtures as fx
from karpenter_amd.scheduling import NewScheduler


@pytest.fixture(scope="module")
def emu():
    import __graft_entry__  # noqa: F401
    return parity.build_emu()


def strip(cmd):
    return {k: cmd.get(k) for k in ("decision", "candidates", "replacement", "replacementCapacityType")}


@pytest.fixture(autouse=True)
def oracle_judges_every_command(monkeypatch, oracle):
    """Round 4: the consolidation DECISION has its own restatement in the oracle (oracle/consolidation.hpp: computeConsolidation,
    the price filter, the minValues re-check, the spot-to-spot branch, filterOutSameInstanceType). Every command this module
    derives with the oracle as the solver — the reference's known answers below included — is computed a second time by
    that restatement, from the cluster document alone, and must be the same command; so the product's Python (disruption.py)
    and C++ (ksched_sweep) verdict layers are both held against code that shares no line with them."""
    real, real_filter = dz.compute_consolidation, dz.filter_out_same_instance_type
    seen = {"commands": 0}

    def key(cmd):
        return (cmd["decision"], cmd["replacement"], cmd.get("replacementCapacityType"))

    def checked(cluster, candidates, solver, results=None):
        cmd = real(cluster, candidates, solver, results=results)
        if solver is oracle.solve and results is None:
            v = oracle.cluster_verdicts(cluster, [candidates], well_known=fx.KWOK_WELL_KNOWN)[0]
            assert key(cmd) == oracle.verdict_key(v), (key(cmd), v)
            seen["commands"] += 1
            cmd["_oracle_cluster"] = cluster
        return cmd

    def checked_filter(cluster, candidates, cmd):
        before = list(cmd["replacement"])
        ok = real_filter(cluster, candidates, cmd)
        if cmd.get("_oracle_cluster") is cluster:      # one step of the multi-node search, judged as a whole by the oracle
            v = oracle.cluster_verdicts(cluster, [candidates], multi_node=True, well_known=fx.KWOK_WELL_KNOWN)[0]
            want = (dz.REPLACE, sorted(cmd["replacement"])) if ok else (dz.NOOP, None)
            assert want == oracle.verdict_key(v)[:2], (want, v, before)
        return ok

    monkeypatch.setattr(dz, "compute_consolidation", checked)
    monkeypatch.setattr(dz, "filter_out_same_instance_type", checked_filter)
    yield seen


def test_simulate_scheduling_delete_replace_noop(oracle, emu):
    # consolidation_test.go Delete :2396-, Replace :1005-, "can't remove without creating N candidates"
    its = fx.kwok_catalog(144)
    by = {t["name"]: t for t in its}
    np_ = fx.node_pool("default"); np_["nodeClassLabelKey"] = "karpenter.kwok.sh/kwoknodeclass"
    def node(name, it, pods_cpu, ct="on-demand"):
        pods = [fx.pod(requests={"cpu": c, "memory": "128Mi"}, phase="Running", node_name=name) for c in pods_cpu]
        used = {"cpu": f"{sum(int(float(c[:-1])) if c.endswith('m') else int(float(c) * 1000) for c in pods_cpu)}m", "pods": str(len(pods))}
        n = fx.state_node(name, by[it], "test-zone-a", ct, "default", used=used)
        n["pods"] = pods
        return n
    big_empty = node("big", "s-16x-amd64-linux", [])
    small_full = node("small", "c-2x-amd64-linux", ["500m", "500m"])
    oversized = node("oversized", "m-32x-amd64-linux", ["1000m"])
    cluster = {"instanceTypes": its, "nodePools": [np_], "nodes": [big_empty, small_full, oversized], "pendingPods": [], "wellKnownLabels": fx.KWOK_WELL_KNOWN}
    dev = lambda p: NewScheduler(p, solver_lib=emu).Solve()
    # the small node's pods fit on the big one: DELETE
    a, b = dz.compute_consolidation(cluster, [small_full], dev), dz.compute_consolidation(cluster, [small_full], oracle.solve)
    assert strip(a) == strip(b) and a["decision"] == dz.DELETE
    # without spare capacity the oversized node is REPLACED by something cheaper
    cluster2 = dict(cluster, nodes=[oversized])
    a, b = dz.compute_consolidation(cluster2, [oversized], dev), dz.compute_consolidation(cluster2, [oversized], oracle.solve)
    assert strip(a) == strip(b) and a["decision"] == dz.REPLACE and "m-64x-amd64-linux" not in a["replacement"] and a["replacementCapacityType"] == "spot"
    # a node that is already the cheapest fit: no-op
    tight = node("tight", "c-1x-amd64-linux", ["800m"], ct="spot")
    cluster3 = dict(cluster, nodes=[tight])
    a, b = dz.compute_consolidation(cluster3, [tight], dev), dz.compute_consolidation(cluster3, [tight], oracle.solve)
    assert strip(a) == strip(b) and a["decision"] == dz.NOOP


def test_sweep_and_binary_search_match_oracle(oracle, emu):
    cluster = dz.make_cluster(n_nodes=40, pods_per_node=5, seed=3)
    cands = dz.sort_candidates(cluster, cluster["nodes"])
    dev = lambda p: NewScheduler(p, solver_lib=emu).Solve()
    got = dz.sweep(cluster, cands[:15], dev, workers=4)
    want = dz.sweep(cluster, cands[:15], oracle.solve)
    assert [strip(c) for c in got] == [strip(c) for c in want]
    assert any(c["decision"] != dz.NOOP for c in got)
    for g, w in zip(got, want):
        parity.assert_same_results(g["results"], w["results"])
    a, pa = dz.first_n_consolidation_option(cluster, cands, dev)
    b, pb = dz.first_n_consolidation_option(cluster, cands, oracle.solve)
    assert pa == pb and strip(a) == strip(b)   # same probe sequence, same command (multinodeconsolidation.go:136-199)
    assert strip(dz.single_node_consolidation(cluster, cands, dev)) == strip(dz.single_node_consolidation(cluster, cands, oracle.solve))


def test_batched_sweep_matches_sequential(oracle, emu):
    """Every probe of the sweep in one ksolve_solve_batch launch gives the same decisions as probe-by-probe solving."""
    from karpenter_amd.scheduling import NewScheduler, SolveBatch
    cluster = dz.make_cluster(n_nodes=40, pods_per_node=5, seed=5)
    cands = dz.sort_candidates(cluster, cluster["nodes"])[:12]

    def batch(problems):
        return SolveBatch([NewScheduler(p, solver_lib=emu) for p in problems])

    got = dz.sweep_batched(cluster, cands, batch)
    want = dz.sweep(cluster, cands, oracle.solve)
    keys = ("decision", "candidates", "replacement", "replacementCapacityType")
    assert [{k: c.get(k) for k in keys} for c in got] == [{k: c.get(k) for k in keys} for c in want]
    for g, w in zip(got, want):
        parity.assert_same_results(g["results"], w["results"])


def test_wont_delete_node_if_it_would_violate_anti_affinity(oracle, emu):
    """consolidation_test.go:4599-4657: three nodes of the cheapest type, one pod each, the pods repel each other on
    hostname. Deleting a node would put its pod next to another one; replacing it is not cheaper: nothing happens."""
    its = fx.kwok_catalog(144)
    np_ = fx.node_pool("default"); np_["nodeClassLabelKey"] = "karpenter.kwok.sh/kwoknodeclass"
    lab = {"app": "test"}
    fits = [t for t in its if int(t["capacity"]["cpu"]) >= 2 and "linux" in t["name"] and "amd64" in t["name"]]
    cheapest = min(fits, key=lambda t: min(o["price"] for o in t["offerings"]))
    zone_ct = min(cheapest["offerings"], key=lambda o: o["price"])
    zone = [r["values"][0] for r in zone_ct["requirements"] if r["key"] == fx.ZONE][0]
    ct = [r["values"][0] for r in zone_ct["requirements"] if r["key"] == fx.CAPACITY_TYPE][0]
    nodes = []
    for i in range(3):
        pod = fx.pod(labels=lab, requests={"cpu": "1"}, phase="Running", node_name=f"node-{i}", pod_anti_requirements=[fx.affinity_term(fx.HOSTNAME, lab)])
        n = fx.state_node(f"node-{i}", cheapest, zone, ct, "default", used={"cpu": "1", "pods": "1"})
        n["pods"] = [pod]
        nodes.append(n)
    cluster = {"instanceTypes": its, "nodePools": [np_], "nodes": nodes, "pendingPods": [], "wellKnownLabels": fx.KWOK_WELL_KNOWN}
    for solver in (oracle.solve, lambda p: NewScheduler(p, solver_lib=emu).Solve()):
        cmds = dz.sweep(cluster, nodes, solver)
        assert [c["decision"] for c in cmds] == [dz.NOOP] * 3
        multi, _ = dz.first_n_consolidation_option(cluster, nodes, solver)
        assert multi["decision"] == dz.NOOP
    # without the anti-affinity the same cluster consolidates (two pods fit one node of that type? no: delete needs room) — the
    # pods then simply move to the other nodes if they have room: here each node has 1 cpu left of 2, so deletion works
    for n in nodes:
        n["pods"][0].pop("podAntiAffinity")
    cmds = dz.sweep(cluster, nodes, oracle.solve)
    assert cmds[0]["decision"] in (dz.DELETE, dz.NOOP)


# ---- known answers from consolidation_test.go -----------------------------------------------------------------------

def _solvers(oracle, emu):
    return (oracle.solve, lambda p: NewScheduler(p, solver_lib=emu).Solve())


def _node_with_pods(name, it, zone, ct, cpus, pool="default", labels=None, **kw):
    pods = [fx.pod(labels=labels or {"app": "test"}, requests={"cpu": c}, phase="Running", node_name=name) for c in cpus]
    milli = sum(int(c[:-1]) if c.endswith("m") else int(float(c) * 1000) for c in cpus)
    n = fx.state_node(name, it, zone, ct, pool, used={"cpu": f"{milli}m", "pods": str(len(pods))}, **kw)
    n["pods"] = pods
    return n


def test_wont_replace_with_a_more_expensive_node(oracle, emu):
    """consolidation_test.go:2222-2394: the replacement's worst launch price among the offerings the claim admits must be
    below the current node's price (nodeclaim.go:411-420, types.go:587-598) — a cheap spot zone does not help if another
    admitted spot zone costs more, and an on-demand-only NodePool compares on-demand prices."""
    zones = ["test-zone-1a", "test-zone-1b", "test-zone-1c"]
    current = fx.fake_instance_type("current-on-demand", {"cpu": "32", "memory": "64Gi", "pods": "100"}, offerings=[fx.offering("on-demand", zones[0], 0.5, available=False)])
    spot_repl = fx.fake_instance_type("potential-spot-replacement", {"cpu": "32", "memory": "64Gi", "pods": "100"},
                                      offerings=[fx.offering("spot", zones[0], 1.0), fx.offering("spot", zones[1], 0.2), fx.offering("spot", zones[2], 0.4)])
    od_repl = fx.fake_instance_type("on-demand-replacement", {"cpu": "32", "memory": "64Gi", "pods": "100"},
                                    offerings=[fx.offering("on-demand", zones[0], 0.6), fx.offering("on-demand", zones[1], 0.6), fx.offering("spot", zones[1], 0.2), fx.offering("spot", zones[2], 0.3)])
    for its, pool in (([current, spot_repl], fx.node_pool()), ([current, od_repl], fx.node_pool(requirements=[fx.req(fx.CAPACITY_TYPE, "In", "on-demand")]))):
        node = _node_with_pods("node-1", current, zones[0], "on-demand", ["1"])
        cluster = {"instanceTypes": its, "nodePools": [pool], "nodes": [node], "pendingPods": []}
        for solver in _solvers(oracle, emu):
            cmd = dz.compute_consolidation(cluster, [node], solver)
            assert len(cmd["results"]["newNodeClaims"]) == 1 and cmd["decision"] == dz.NOOP
    # the same on-demand pool with a cheaper on-demand offering is replaced
    od_cheap = fx.fake_instance_type("on-demand-replacement", {"cpu": "32", "memory": "64Gi", "pods": "100"}, offerings=[fx.offering("on-demand", zones[0], 0.4), fx.offering("spot", zones[1], 0.2)])
    node = _node_with_pods("node-1", current, zones[0], "on-demand", ["1"])
    cluster = {"instanceTypes": [current, od_cheap], "nodePools": [fx.node_pool(requirements=[fx.req(fx.CAPACITY_TYPE, "In", "on-demand")])], "nodes": [node], "pendingPods": []}
    for solver in _solvers(oracle, emu):
        cmd = dz.compute_consolidation(cluster, [node], solver)
        assert cmd["decision"] == dz.REPLACE and cmd["replacement"] == ["on-demand-replacement"]


def test_can_delete_nodes(oracle, emu):
    """consolidation_test.go:2421-2460 (two pods on node 0, one on node 1: node 1 is deleted), :2539-2585 (capacity
    Karpenter does not own can take the pods), :3442-3480 (no delete if a pod would go pending)."""
    its = fx.fake_default_instance_types()
    by = {t["name"]: t for t in its}
    big = by["default-instance-type"]                                        # 4 cpu
    n0 = _node_with_pods("node-0", big, "test-zone-1", "on-demand", ["1", "1"])
    n1 = _node_with_pods("node-1", big, "test-zone-1", "on-demand", ["1"])
    cluster = {"instanceTypes": its, "nodePools": [fx.node_pool()], "nodes": [n0, n1], "pendingPods": []}
    for solver in _solvers(oracle, emu):
        cands = dz.sort_candidates(cluster, cluster["nodes"])
        assert [c["name"] for c in cands] == ["node-1", "node-0"]            # one pod is cheaper to disrupt than two
        cmd = dz.single_node_consolidation(cluster, cands, solver)
        assert cmd["decision"] == dz.DELETE and cmd["candidates"] == ["node-1"]
        assert dz.validate_command(cluster, [n1], cmd, solver) is None
    # an unmanaged node with room: the Karpenter node is deleted, its pods go there
    from test_reference_known_answers import bare_node
    foreign = bare_node("foreign", cpu="8")
    foreign["pods"] = []
    cluster = {"instanceTypes": its, "nodePools": [fx.node_pool()], "nodes": [n1, foreign], "pendingPods": []}
    for solver in _solvers(oracle, emu):
        cmd = dz.compute_consolidation(cluster, [n1], solver)
        assert cmd["decision"] == dz.DELETE and [e["name"] for e in cmd["results"]["existingNodes"] if e["pods"]] == ["foreign"]
    # a pod that fits nowhere else and no instance type is cheaper than the current one: nothing to do
    small = by["small-instance-type"]
    s0 = _node_with_pods("small-0", small, "test-zone-1", "spot", ["1500m"])
    s1 = _node_with_pods("small-1", small, "test-zone-1", "spot", ["1500m"])
    cluster = {"instanceTypes": its, "nodePools": [fx.node_pool()], "nodes": [s0, s1], "pendingPods": []}
    for solver in _solvers(oracle, emu):
        assert [c["decision"] for c in dz.sweep(cluster, [s0, s1], solver)] == [dz.NOOP, dz.NOOP]


def test_wont_delete_onto_uninitialized_or_settling_nodes(oracle, emu):
    """consolidation_test.go:3004-3048 (pods may not be counted on to land on an uninitialized node) and :3050-3119 (nor on
    a node still inside its consolidateAfter window); :3121-3190 consolidateAfter=Never nodes do take them."""
    its = fx.fake_default_instance_types()
    big = {t["name"]: t for t in its}["default-instance-type"]
    src = _node_with_pods("src", big, "test-zone-1", "on-demand", ["1"])
    for kw, decision in (({"initialized": False}, dz.NOOP), ({"under_consolidate_after": True}, dz.NOOP), ({}, dz.DELETE)):
        dst = _node_with_pods("dst", big, "test-zone-1", "on-demand", [], **kw)
        cluster = {"instanceTypes": [big], "nodePools": [fx.node_pool()], "nodes": [src, dst], "pendingPods": []}
        for solver in _solvers(oracle, emu):
            assert dz.compute_consolidation(cluster, [src], solver)["decision"] == decision, kw


def test_delete_with_a_permanently_pending_pod(oracle, emu):
    """consolidation_test.go:3390-3440: a pending pod that can never schedule does not block the deletion of a node whose
    own pods fit elsewhere (AllNonPendingPodsScheduled, scheduler.go:388-392)."""
    its = fx.fake_default_instance_types()
    big = {t["name"]: t for t in its}["default-instance-type"]
    n0 = _node_with_pods("node-0", big, "test-zone-1", "on-demand", ["1", "1"])
    n1 = _node_with_pods("node-1", big, "test-zone-1", "on-demand", ["1"])
    stuck = fx.pod(requests={"cpu": "1"}, node_selector={"non-existent": "node-label"})
    cluster = {"instanceTypes": its, "nodePools": [fx.node_pool()], "nodes": [n0, n1], "pendingPods": [stuck]}
    for solver in _solvers(oracle, emu):
        cmd = dz.compute_consolidation(cluster, [n1], solver)
        assert cmd["decision"] == dz.DELETE and stuck["uid"] in cmd["results"]["podErrors"]


def test_replace_keeps_zonal_spread(oracle, emu):
    """consolidation_test.go:4525-4597: three oversized nodes, one per zone, one spread pod each. A node can be replaced
    by a cheaper one, and the replacement is pinned to the zone the spread constraint leaves open — the candidate's own."""
    its = fx.fake_default_instance_types()
    by = {t["name"]: t for t in its}
    lab = {"app": "test"}
    tsc = [fx.spread(fx.ZONE, lab)]
    nodes = []
    for i, z in enumerate(("test-zone-1", "test-zone-2", "test-zone-3")):
        pod = fx.pod(labels=lab, requests={"cpu": "1"}, topology_spread=tsc, phase="Running", node_name=f"node-{i}")
        n = fx.state_node(f"node-{i}", by["arm-instance-type"], z, "on-demand", "default", used={"cpu": "15", "pods": "1"})   # no room for a second pod
        n["pods"] = [pod]
        nodes.append(n)
    cluster = {"instanceTypes": its, "nodePools": [fx.node_pool()], "nodes": nodes, "pendingPods": []}
    for solver in _solvers(oracle, emu):
        cmd = dz.compute_consolidation(cluster, [nodes[1]], solver)
        assert cmd["decision"] == dz.REPLACE and "arm-instance-type" not in cmd["replacement"]
        zone = [q["values"] for q in cmd["results"]["newNodeClaims"][0]["requirements"] if q["key"] == fx.ZONE][0]
        assert zone == ["test-zone-2"]


def test_merge_three_nodes_into_one(oracle, emu):
    """consolidation_test.go:3982-4028: three nodes of the most expensive on-demand type with one small pod each are
    replaced by a single cheaper node — the multi-node search ends on the whole prefix (multinodeconsolidation.go:117-207)."""
    its = fx.fake_instance_types_assorted()
    priciest, offering = max(((t, o) for t in its for o in t["offerings"] if dz._capacity_type(o) == "on-demand"), key=lambda x: x[1]["price"])
    zone = [r["values"][0] for r in offering["requirements"] if r["key"] == fx.ZONE][0]
    nodes = [_node_with_pods(f"node-{i}", priciest, zone, "on-demand", ["100m"]) for i in range(3)]
    cluster = {"instanceTypes": its, "nodePools": [fx.node_pool()], "nodes": nodes, "pendingPods": []}
    out = []
    for solver in _solvers(oracle, emu):
        cmd, probes = dz.first_n_consolidation_option(cluster, dz.sort_candidates(cluster, nodes), solver)
        assert cmd["decision"] == dz.REPLACE and sorted(cmd["candidates"]) == ["node-0", "node-1", "node-2"]
        assert [p[0] for p in probes] == [2, 3] and len(cmd["results"]["newNodeClaims"]) == 1
        assert priciest["name"] not in cmd["replacement"] and cmd["replacement"]
        out.append((cmd["replacement"], probes))
    assert out[0] == out[1]


def test_sweep_on_a_large_cluster(oracle, emu):
    """BASELINE configs[4] shape at test size: thousands of existing nodes with their bound pods, a batched single-node
    sweep over the best candidates (every probe re-solves against all the other nodes) — decisions and every probe's
    placements equal the oracle's."""
    from karpenter_amd.scheduling import SolveBatch
    cluster = dz.make_cluster(n_nodes=2500, pods_per_node=6, seed=7)
    cands = dz.sort_candidates(cluster, cluster["nodes"])[:5]
    got = dz.sweep_batched(cluster, cands, lambda ps: SolveBatch([NewScheduler(p, solver_lib=emu) for p in ps]))
    want = dz.sweep(cluster, cands, oracle.solve)
    keys = ("decision", "candidates", "replacement", "replacementCapacityType")
    assert [{k: c.get(k) for k in keys} for c in got] == [{k: c.get(k) for k in keys} for c in want]
    for g, w in zip(got, want):
        parity.assert_same_results(g["results"], w["results"])
    assert any(c["decision"] != dz.NOOP for c in got)


# ---- spot-to-spot consolidation: consolidation_test.go:1005-1250, consolidation.go:261-342 --------------------------

def _spot_cluster(its, node_type, n_nodes=1, gate=True):
    off = next(o for o in node_type["offerings"] if dz._capacity_type(o) == "spot")
    zone = [r["values"][0] for r in off["requirements"] if r["key"] == fx.ZONE][0]
    nodes = [_node_with_pods(f"spot-{i}", node_type, zone, "spot", ["100m"]) for i in range(n_nodes)]
    return {"instanceTypes": its, "nodePools": [fx.node_pool()], "nodes": nodes, "pendingPods": [], "options": {"spotToSpotConsolidation": gate}}, nodes


def _spot_price(t):
    return min([o["price"] for o in t["offerings"] if dz._capacity_type(o) == "spot"], default=None)


def test_spot_to_spot_replacement(oracle, emu):
    its = fx.fake_instance_types_assorted()
    spot_types = [t for t in its if _spot_price(t) is not None]
    priciest = max(spot_types, key=_spot_price)
    # :1005-1058 (spot entry) — the most expensive spot node is replaced by the 15 cheapest spot types
    cluster, nodes = _spot_cluster(its, priciest)
    for solver in _solvers(oracle, emu):
        cmd = dz.compute_consolidation(cluster, nodes, solver)
        assert cmd["decision"] == dz.REPLACE and len(cmd["replacement"]) == dz.MIN_INSTANCE_TYPES_FOR_SPOT_TO_SPOT and cmd["replacementCapacityType"] == "spot"
        assert priciest["name"] not in cmd["replacement"]
        prices = [_spot_price(next(t for t in its if t["name"] == n)) for n in cmd["replacementInPriceOrder"]]
        assert prices == sorted(prices) and max(prices) <= min(_spot_price(t) for t in spot_types if t["name"] not in cmd["replacement"])
    # :1136-1175 — the feature gate is off: a spot node is never replaced by a spot node
    cluster_off, nodes_off = _spot_cluster(its, priciest, gate=False)
    for solver in _solvers(oracle, emu):
        assert dz.compute_consolidation(cluster_off, nodes_off, solver)["decision"] == dz.NOOP
    # :3982-4028 (spot entry) — several spot nodes merge without the 15-type rule
    cluster3, nodes3 = _spot_cluster(its, priciest, n_nodes=3)
    for solver in _solvers(oracle, emu):
        cmd, _ = dz.first_n_consolidation_option(cluster3, nodes3, solver)
        assert cmd["decision"] == dz.REPLACE and len(cmd["candidates"]) == 3 and len(cmd["replacement"]) > dz.MIN_INSTANCE_TYPES_FOR_SPOT_TO_SPOT


def test_spot_to_spot_needs_fifteen_cheaper_types(oracle, emu):
    import copy
    # :1061-1134 — five instance types in all, one made very cheap: a single cheaper option is not enough flexibility
    its = copy.deepcopy(fx.fake_instance_types_assorted()[:5])
    its[0]["offerings"][0]["price"] = 0.001
    for r in its[0]["offerings"][0]["requirements"]:
        if r["key"] == fx.CAPACITY_TYPE:
            r["values"] = ["spot"]
    for r in its[0]["requirements"]:
        if r["key"] == fx.CAPACITY_TYPE and "spot" not in r["values"]:
            r["values"].append("spot")
    spot_types = sorted([t for t in its if _spot_price(t) is not None], key=_spot_price)
    cluster, nodes = _spot_cluster(its, spot_types[-1])
    for solver in _solvers(oracle, emu):
        cmd = dz.compute_consolidation(cluster, nodes, solver)
        assert cmd["decision"] == dz.NOOP and "requires 15 cheaper instance type options" in cmd.get("reason", "")
    # :1177-1245 — twenty types; the node's type is the second cheapest spot type, i.e. among the 15 cheapest: nothing to do
    its = copy.deepcopy(fx.fake_instance_types_assorted()[:20])
    its[0]["offerings"][0]["price"] = 0.001
    spot_types = sorted([t for t in its if _spot_price(t) is not None], key=_spot_price)
    if len(spot_types) >= 2:
        cluster, nodes = _spot_cluster(its, spot_types[1])
        for solver in _solvers(oracle, emu):
            assert dz.compute_consolidation(cluster, nodes, solver)["decision"] == dz.NOOP


def test_consolidation_ignoring_preferences(oracle, emu):
    """consolidation_test.go:4952-5062 — with PreferencePolicy=Ignore the simulation does not honour preferred
    anti-affinity (the pod moves next to its peers: DELETE) or a preferred instance type (the node is REPLACED by a
    cheaper one); with the default policy the same clusters stay as they are / replace onto the preferred type only."""
    its = fx.fake_instance_types_assorted()
    cheapest = min(its, key=lambda t: min(o["price"] for o in t["offerings"] if dz._capacity_type(o) == "on-demand") if any(dz._capacity_type(o) == "on-demand" for o in t["offerings"]) else 1e9)
    five = next(t for t in its if t["capacity"]["cpu"] == "8" and any(dz._capacity_type(o) == "on-demand" for o in t["offerings"]))
    zone = [r["values"][0] for o in five["offerings"] if dz._capacity_type(o) == "on-demand" for r in o["requirements"] if r["key"] == fx.ZONE][0]
    lab = {"app": "foo"}
    anti = [fx.weighted(1, fx.affinity_term(fx.HOSTNAME, lab))]

    def node(name, n_pods):
        pods = [fx.pod(labels=lab, requests={"cpu": "100m"}, pod_anti_preferences=anti, phase="Running", node_name=name) for _ in range(n_pods)]
        n = fx.state_node(name, five, zone, "on-demand", "default", used={"cpu": f"{100 * n_pods}m", "pods": str(n_pods)})
        n["pods"] = pods
        return n
    for policy, want in (("Ignore", dz.DELETE), ("Respect", dz.REPLACE)):
        n0, n1 = node("node-0", 2), node("node-1", 1)
        cluster = {"instanceTypes": its, "nodePools": [fx.node_pool()], "nodes": [n0, n1], "pendingPods": [], "options": {"preferencePolicy": policy}}
        for solver in _solvers(oracle, emu):
            cmd = dz.compute_consolidation(cluster, [n1], solver)
            # Ignore: the preference is dropped before scheduling and the pod joins node-0. Respect: the preferred
            # anti-affinity keeps it off node-0 first, so it gets a node of its own — a cheaper one
            assert cmd["decision"] == want, (policy, cmd["decision"])
    # :5017-5062 a pod that merely prefers the most expensive type does not pin the node to it
    priciest, off = max(((t, o) for t in its for o in t["offerings"] if dz._capacity_type(o) == "on-demand"), key=lambda x: x[1]["price"])
    pzone = [r["values"][0] for r in off["requirements"] if r["key"] == fx.ZONE][0]
    pod = fx.pod(labels=lab, requests={"cpu": "100m"}, node_preferences=[fx.req(fx.INSTANCE_TYPE, "In", priciest["name"])], phase="Running", node_name="big")
    big = fx.state_node("big", priciest, pzone, "on-demand", "default", used={"cpu": "100m", "pods": "1"})
    big["pods"] = [pod]
    cluster = {"instanceTypes": its, "nodePools": [fx.node_pool()], "nodes": [big], "pendingPods": [], "options": {"preferencePolicy": "Ignore"}}
    for solver in _solvers(oracle, emu):
        cmd = dz.compute_consolidation(cluster, [big], solver)
        assert cmd["decision"] == dz.REPLACE and priciest["name"] not in cmd["replacement"]
    cluster["options"] = {"preferencePolicy": "Respect"}
    for solver in _solvers(oracle, emu):
        assert dz.compute_consolidation(cluster, [big], solver)["decision"] == dz.NOOP      # the only acceptable type is the current one


def test_consolidation_into_reserved_capacity(oracle, emu):
    """consolidation_test.go:4778-4949 — with the ReservedCapacity gate a node moves into a (much cheaper) reserved
    offering: from on-demand / spot into the reservation of its own instance type, and from one reservation to the
    reservation of the cheapest type."""
    import copy
    base = fx.fake_instance_types_assorted()

    def with_reservation(its, t):
        t = next(x for x in its if x["name"] == t["name"])
        off = t["offerings"][0]
        zone = [r["values"][0] for r in off["requirements"] if r["key"] == fx.ZONE][0]
        for r in t["requirements"]:
            if r["key"] == fx.CAPACITY_TYPE and "reserved" not in r["values"]:
                r["values"].append("reserved")
        t["offerings"].append(fx.offering("reserved", zone, off["price"] / 1_000_000.0, reservation_id="r-" + t["name"], reservation_capacity=10))
        return zone

    def od_price(t):
        return min([o["price"] for o in t["offerings"] if dz._capacity_type(o) == "on-demand"], default=None)
    od_types = [t for t in base if od_price(t) is not None]
    priciest, cheapest = max(od_types, key=od_price), min(od_types, key=od_price)
    # from on-demand into the reservation of the same (most expensive) instance type
    its = copy.deepcopy(base)
    zone = with_reservation(its, priciest)
    p_it = next(t for t in its if t["name"] == priciest["name"])
    node = _node_with_pods("od-node", p_it, zone, "on-demand", ["100m"])
    cluster = {"instanceTypes": its, "nodePools": [fx.node_pool()], "nodes": [node], "pendingPods": [], "options": {"reservedCapacity": True},
               "wellKnownLabels": fx.FAKE_WELL_KNOWN}       # the fake provider registers the reservation id label as well-known (fake/cloudprovider.go:44)
    for solver in _solvers(oracle, emu):
        cmd = dz.compute_consolidation(cluster, [node], solver)
        claim = cmd["results"]["newNodeClaims"][0]
        assert cmd["decision"] == dz.REPLACE and claim["reservedOfferings"] == ["r-" + priciest["name"]]
        assert [q["values"] for q in claim["requirements"] if q["key"] == fx.CAPACITY_TYPE] == [["reserved"]]
        assert cmd["replacement"] == [priciest["name"]]
    # from that reservation into the reservation of the cheapest type
    its = copy.deepcopy(base)
    zone = with_reservation(its, priciest)
    with_reservation(its, cheapest)
    p_it = next(t for t in its if t["name"] == priciest["name"])
    node = _node_with_pods("reserved-node", p_it, zone, "reserved", ["100m"], extra_labels={"karpenter.sh/reservation-id": "r-" + priciest["name"]})
    cluster = {"instanceTypes": its, "nodePools": [fx.node_pool()], "nodes": [node], "pendingPods": [], "options": {"reservedCapacity": True},
               "wellKnownLabels": fx.FAKE_WELL_KNOWN}
    for solver in _solvers(oracle, emu):
        cmd = dz.compute_consolidation(cluster, [node], solver)
        assert cmd["decision"] == dz.REPLACE and cheapest["name"] in cmd["replacement"] and priciest["name"] not in cmd["replacement"]


def test_consolidation_respects_min_values_after_the_price_filter(oracle, emu):
    """consolidation_test.go:5064-5145 — two instance types, the NodePool wants three (BestEffort relaxes that to two for the
    simulation); dropping the current, more expensive type leaves one: the replacement would break minValues, so nothing
    happens (RemoveInstanceTypeOptionsByPriceAndMinValues, nodeclaim.go:411-420)."""
    its_all = fx.fake_instance_types_assorted()
    def od(t):
        return min([o["price"] for o in t["offerings"] if dz._capacity_type(o) == "on-demand"], default=None)
    od_types = [t for t in its_all if od(t) is not None]
    cheap, pricey = min(od_types, key=od), max(od_types, key=od)
    its = [cheap, pricey]
    pool = fx.node_pool(weight=100, requirements=[fx.req(fx.INSTANCE_TYPE, "In", cheap["name"], pricey["name"], "a-third-type", min_values=3)])
    zone = [r["values"][0] for o in pricey["offerings"] if dz._capacity_type(o) == "on-demand" for r in o["requirements"] if r["key"] == fx.ZONE][0]
    node = _node_with_pods("node-0", pricey, zone, "on-demand", ["100m"])
    for policy in ("BestEffort", "Strict"):
        cluster = {"instanceTypes": its, "nodePools": [pool], "nodes": [node], "pendingPods": [], "options": {"minValuesPolicy": policy}, "wellKnownLabels": fx.FAKE_WELL_KNOWN}
        for solver in _solvers(oracle, emu):
            cmd = dz.compute_consolidation(cluster, [node], solver)
            assert cmd["decision"] == dz.NOOP
            if policy == "BestEffort":
                assert len(cmd["results"]["newNodeClaims"]) == 1 and "minValues" in cmd.get("reason", "")
    # without minValues the same node is simply replaced by the cheap type
    cluster = {"instanceTypes": its, "nodePools": [fx.node_pool()], "nodes": [node], "pendingPods": [], "wellKnownLabels": fx.FAKE_WELL_KNOWN}
    for solver in _solvers(oracle, emu):
        cmd = dz.compute_consolidation(cluster, [node], solver)
        assert cmd["decision"] == dz.REPLACE and cmd["replacement"] == [cheap["name"]]


def test_simulation_truncates_to_600_instance_types_and_rechecks_min_values(oracle, emu):
    """helpers.go:131 — SimulateScheduling ends with results.TruncateInstanceTypes(ctx, MaxInstanceTypes): the replacement's
    options are the 600 cheapest, and a NodePool whose minValues needs more distinct instance types than the cut leaves
    loses the claim (its pods fail, scheduler.go:419-437), so the candidate is not consolidated. 700 kwok types."""
    its = fx.kwok_catalog(700)
    assert len(its) == 700
    pricey = max([t for t in its if "linux" in t["name"] and "amd64" in t["name"]], key=lambda t: t["offerings"][0]["price"])
    def cluster_with(min_values):
        pool = fx.node_pool(requirements=[fx.req(fx.INSTANCE_TYPE, "Exists", min_values=min_values)] if min_values else None)
        pool["nodeClassLabelKey"] = "karpenter.kwok.sh/kwoknodeclass"
        node = _node_with_pods("node-0", pricey, fx.KWOK_ZONES[0], "on-demand", ["100m"])
        return {"instanceTypes": its, "nodePools": [pool], "nodes": [node], "pendingPods": [], "wellKnownLabels": fx.KWOK_WELL_KNOWN}, node
    for solver in _solvers(oracle, emu):
        cluster, node = cluster_with(None)
        cmd = dz.compute_consolidation(cluster, [node], solver)
        assert cmd["decision"] == dz.REPLACE
        res = dz.simulate_scheduling(cluster, [node], solver)
        assert len(res["newNodeClaims"]) == 1 and len(res["newNodeClaims"][0]["instanceTypes"]) == 600   # 700 compatible types, capped
        # minValues = 650 distinct instance types: met by the 700 options Solve() leaves, broken by the cut to 600
        cluster, node = cluster_with(650)
        res = dz.simulate_scheduling(cluster, [node], solver)
        assert not res["newNodeClaims"] and not res["allNonPendingPodsScheduled"]
        assert dz.compute_consolidation(cluster, [node], solver)["decision"] == dz.NOOP
        # minValues = 500 survives the cut
        cluster, node = cluster_with(500)
        assert dz.compute_consolidation(cluster, [node], solver)["decision"] == dz.REPLACE


def test_single_node_candidate_order(oracle):
    """singlenodeconsolidation_test.go:104-170 — candidates sorted by savings ratio, then dealt out over the NodePools;
    a pool that an earlier, timed-out run did not reach comes first."""
    its = fx.fake_default_instance_types()
    it = {t["name"]: t for t in its}["default-instance-type"]
    pools = [fx.node_pool(f"nodepool-{i}") for i in (1, 2, 3)]
    nodes = []
    for pi, pool in enumerate(pools):
        for k, n_pods in enumerate((0, 4, 19)):          # disruption cost 1, 5, 20 -> three distinct savings ratios per pool
            nodes.append(_node_with_pods(f"{pool['name']}-n{k}", it, "test-zone-1", "on-demand", ["10m"] * n_pods, pool=pool["name"]))
    cluster = {"instanceTypes": its, "nodePools": pools, "nodes": nodes, "pendingPods": []}
    ranked = dz.sort_candidates(cluster, list(reversed(nodes)))
    costs = [dz.reschedule_disruption_cost(n["pods"]) for n in ranked]
    assert costs == [1.0] * 3 + [5.0] * 3 + [20.0] * 3                                        # :130-170
    ratios = [dz.savings_ratio(cluster, n) for n in ranked]
    assert ratios == sorted(ratios, reverse=True)                                             # :104-117
    woven = dz.interweave_by_nodepool(ranked)
    assert [dz._pool_name(n) for n in woven[:3]] == sorted({p["name"] for p in pools}) and len(woven) == 9
    assert [dz._pool_name(n) for n in woven] == [dz._pool_name(n) for n in woven[:3]] * 3      # round robin
    first = dz.interweave_by_nodepool(ranked, previously_unseen=["nodepool-2"])
    assert dz._pool_name(first[0]) == "nodepool-2" and len(first) == 9                        # :119-128


def _tight_cluster(seed, n_nodes=30, limits=None):
    """make_cluster with most nodes squeezed (what is still available on them shrunk to a sliver), so that displaced pods
    often need a replacement NodeClaim instead of fitting elsewhere; a few nodes uninitialized / inside consolidateAfter,
    one marked for deletion, one pending pod."""
    import random
    rng = random.Random(seed)
    cluster = dz.make_cluster(n_nodes=n_nodes, pods_per_node=5, n_types=60, seed=seed, utilisation=0.8)
    for i, n in enumerate(cluster["nodes"]):
        if rng.random() < 0.95:
            n["available"] = dict(n["available"], cpu=f"{rng.choice([0, 100, 300])}m")
        if i % 11 == 3:
            n["initialized"] = False
            n["labels"].pop("karpenter.sh/initialized", None)
        if i % 13 == 5:
            n["underConsolidateAfter"] = True
    cluster["nodes"][-1]["markedForDeletion"] = True
    cluster["pendingPods"] = [fx.pod(requests={"cpu": "250m", "memory": "128Mi"})]
    if limits:
        cluster["nodePools"][0]["limits"] = limits
    return cluster


@pytest.mark.parametrize("seed,limits", [(1, None), (2, {"cpu": "2000"}), (3, {"cpu": "150", "nodes": "31"}), (4, "volumes")])
def test_resident_cluster_probes_match_per_probe_rebuild(oracle, emu, seed, limits):
    """ksolve_probe_create (one ksolve_create for the cluster, a removed-node bitmap + displaced-pod rows per simulation, one
    batched launch) gives, probe by probe, the Results of SimulateScheduling assembled from scratch and solved by the
    oracle (helpers.go:53-155): single-node sweep, multi-node sets, NodePool limits handed back by the removed nodes
    (scheduler.go:835-842), uninitialized / consolidateAfter / deleting nodes, a pending pod."""
    import random
    volumes = limits == "volumes"
    cluster = _tight_cluster(seed, limits=None if volumes else limits)
    if volumes:
        # bound pods with volume requirement alternatives (volumeReqsByPod): the zone of the volume they mount — their node's
        # zone, or that one and a second one
        vr = random.Random(40 + seed)
        for n in cluster["nodes"]:
            for p in n["pods"]:
                if vr.random() < 0.4:
                    zs = [n["labels"][fx.ZONE]] + ([vr.choice(fx.KWOK_ZONES)] if vr.random() < 0.4 else [])
                    p["volumeRequirements"] = [[fx.req(fx.ZONE, "In", z)] for z in zs]
    cands = [n for n in dz.sort_candidates(cluster, cluster["nodes"]) if not n.get("markedForDeletion")][:14]
    got, rc = dz.sweep_resident(cluster, cands, solver_lib=emu)
    want = dz.sweep(cluster, cands, oracle.solve)
    assert [strip(c) for c in got] == [strip(c) for c in want]
    for g, w in zip(got, want):
        parity.assert_same_results(g["results"], w["results"])
        assert g["results"]["counters"]["referenceBinEvaluations"] == w["results"]["counters"]["binEvaluations"]
        assert g["results"]["allNonPendingPodsScheduled"] == w["results"]["allNonPendingPodsScheduled"]
    assert seed != 1 or {c["decision"] for c in got} == {dz.DELETE, dz.REPLACE, dz.NOOP}
    if volumes:   # the volume's zone ends up in the replacement's requirements
        assert any(tuple(r["values"]) in [(z,) for z in fx.KWOK_ZONES] for c in got for cl in c["results"]["newNodeClaims"] for r in cl["requirements"] if r["key"] == fx.ZONE)
    # multi-node sets through the same resident cluster: the binary search's probe sequence and command
    rng = random.Random(seed)
    sets = [rng.sample(cands, k) for k in (2, 3, 5)]
    rc.prefetch(sets)
    for cs in sets:
        g, w = dz.compute_consolidation(cluster, cs, rc), dz.compute_consolidation(cluster, cs, oracle.solve)
        assert strip(g) == strip(w)
        parity.assert_same_results(g["results"], w["results"])
    a, pa = dz.first_n_consolidation_option(cluster, cands, rc)
    b, pb = dz.first_n_consolidation_option(cluster, cands, oracle.solve)
    assert pa == pb and strip(a) == strip(b)
    assert strip(dz.single_node_consolidation(cluster, cands, rc)) == strip(dz.single_node_consolidation(cluster, cands, oracle.solve))
    rc.close()


@pytest.mark.parametrize("seed", [1, 3, 5, 8])
def test_multi_node_search_as_one_sweep(oracle, emu, seed):
    """MultiNodeConsolidation.firstNConsolidationOption (multinodeconsolidation.go:117-207) with EVERY prefix simulated in one
    launch (ResidentCluster.first_n: ksolve_sweep over all the prefixes, verdicts incl. filterOutSameInstanceType from the host
    library, then the binary search as a walk over them): the same probe sequence and command as the search that calls the
    oracle once per step, and every prefix's verdict equals computeConsolidation + the same-type filter on the oracle's Results."""
    cluster = _tight_cluster(seed)
    cands = [n for n in dz.sort_candidates(cluster, cluster["nodes"]) if not n.get("markedForDeletion")][:14]
    rc = dz.ResidentCluster(cluster, cands, solver_lib=emu)
    a, pa = rc.first_n(cands)
    b, pb = dz.first_n_consolidation_option(cluster, cands, oracle.solve)
    assert pa == pb and strip(a) == strip(b)
    sets = [cands[:k] for k in range(2, len(cands) + 1)]
    got = rc.decisions(sets, multi_node=True)
    assert [strip(c) for c in rc.decisions(sets, multi_node=True, library_prices=True)] == [strip(c) for c in got]   # prices / capacity types from the library's node table
    for cs, g, v in zip(sets, got, oracle.cluster_verdicts(cluster, sets, multi_node=True, well_known=fx.KWOK_WELL_KNOWN)):
        w = dz.compute_consolidation(cluster, cs, oracle.solve)
        if w["decision"] == dz.REPLACE and not dz.filter_out_same_instance_type(cluster, cs, w):
            w = {"decision": dz.NOOP, "replacement": None}
        assert (g["decision"], g["replacement"]) == (w["decision"], w.get("replacement")), (len(cs), g, strip(w))
        assert (g["decision"], g["replacement"]) == oracle.verdict_key(v)[:2], (len(cs), g, v)      # the oracle's own decision layer
    rc.close()


@pytest.mark.parametrize("seed,topology", [(2, False), (13, True), (21, True)])
def test_multi_node_sets_on_a_compact_cluster_fuzz(oracle, emu, seed, topology):
    """Prefixes of three windows of sortCandidates' order and arbitrary node sets of 1..24 nodes on a 500-node compact cluster
    (make_resident_cluster, with and without spread constraints on the bound pods): every verdict of the sweep — multiNode filter
    and the library's own candidate prices included — and every reference-equivalent evaluation count equals the oracle's
    simulation of the same set (an offline run of this loop over 24 seeds x 2 = 4608 sets was clean)."""
    import random
    cc = dz.make_resident_cluster(n_nodes=500, seed=seed, topology=topology)
    rc = dz.ResidentCluster.from_compact(cc, solver_lib=emu)
    full, rng, K = dz.compact_candidates(cc), random.Random(seed), 24
    sets, idxs = [], []
    for w in range(4):
        off = rng.randrange(0, len(full) - K)
        for k in range(1, K + 1):
            idx = full[off:off + k] if w < 3 else rng.sample(full, k)
            sets.append([cc["nodes"][i] for i in idx]); idxs.append(idx)
    cmds = rc.decisions(sets, multi_node=True, library_prices=True)
    refs = rc.last_sweep["referenceBinEvaluations"]
    json_form = rc.last_sweep
    # the binary form of the same call (ksched_sweep_arrays: a CSR of node positions in, arrays out — what a cgo caller uses)
    binary = rc.decisions(sets, multi_node=True, library_prices=True, arrays=True)
    strip = lambda c: {k: v for k, v in c.items() if k != "reason"}
    assert [strip(c) for c in binary] == [strip(c) for c in cmds]
    for k in ("decisions", "allNonPendingPodsScheduled", "claims", "status", "referenceBinEvaluations"):
        assert rc.last_sweep[k] == json_form[k], k
    base = dz.compact_problem(cc, pod_groups=[])
    if topology:
        base["clusterPods"] = dz.compact_cluster_pods(cc)
    probes, cand_sets = [], []
    for idx in idxs:
        pods = [dz.compact_node_pods(cc, i) for i in idx]
        probes.append({"removeNodes": [cc["nodes"][i]["name"] for i in idx], "pods": [p for ps in pods for p in ps]})
        cand_sets.append([dict(cc["nodes"][i], pods=ps) for i, ps in zip(idx, pods)])
    # the expectation is the oracle's from end to end: its simulation AND its restatement of computeConsolidation /
    # filterOutSameInstanceType (oracle/consolidation.hpp) — nothing of karpenter_amd.disruption judges the sweep here
    for j, (r, cs) in enumerate(zip(oracle.sweep(base, probes, threads=2, verdicts=True, multi_node=True), cand_sets)):
        assert (cmds[j]["decision"], cmds[j]["replacement"], cmds[j].get("replacementCapacityType")) == oracle.verdict_key(r["verdict"]), (j, len(cs), cmds[j], r["verdict"])
        assert refs[j] == r["counters"]["binEvaluations"], (j, len(cs))
    assert len({c["decision"] for c in cmds}) == 3
    rc.close()


@pytest.mark.parametrize("topology", [False, True])
def test_large_removed_node_lists(oracle, emu, topology):
    """A probe's removed nodes live in two vector registers when there are at most 128 of them (engine.h prm0_ / prm1_: the scan's mask of
    removed nodes and the evaluation count's subtraction are ballots), in the HBM list beyond: sets of 63..65 nodes (the first register's
    edge), 100 and 127..129 (the second's), 140 and 200 (the list), contiguous in sortCandidates' order and scattered over the cluster —
    verdict and reference-equivalent evaluation count against the oracle's simulation of the same set."""
    import random
    cc = dz.make_resident_cluster(n_nodes=700, seed=31, topology=topology)
    rc = dz.ResidentCluster.from_compact(cc, solver_lib=emu)
    full, rng = dz.compact_candidates(cc), random.Random(31)
    idxs = []
    for k in (63, 64, 65, 100, 127, 128, 129, 140, 200):
        off = rng.randrange(0, len(full) - k)
        idxs.append(full[off:off + k])
        idxs.append(sorted(rng.sample(full, k)))
    sets = [[cc["nodes"][i] for i in idx] for idx in idxs]
    cmds = rc.decisions(sets, multi_node=True, library_prices=True, arrays=True)
    refs = rc.last_sweep["referenceBinEvaluations"]
    base = dz.compact_problem(cc, pod_groups=[])
    if topology:
        base["clusterPods"] = dz.compact_cluster_pods(cc)
    probes = [{"removeNodes": [cc["nodes"][i]["name"] for i in idx], "pods": [p for i in idx for p in dz.compact_node_pods(cc, i)]} for idx in idxs]
    for j, r in enumerate(oracle.sweep(base, probes, threads=4, verdicts=True, multi_node=True)):
        assert (cmds[j]["decision"], cmds[j]["replacement"], cmds[j].get("replacementCapacityType")) == oracle.verdict_key(r["verdict"]), (j, len(idxs[j]), cmds[j], r["verdict"])
        assert refs[j] == r["counters"]["binEvaluations"], (j, len(idxs[j]))
    rc.close()


@pytest.mark.parametrize("topology", [False, True])
def test_compact_sweep_equals_the_one_wavefront_kernel(emu, monkeypatch, topology):
    """ksolve_sweep runs a cluster whose dictionaries fit it through the COMPACT form of the launch (ksolve_pack_sweep4: four
    wavefronts per workgroup share the read-only instance-type tables and the template records that wave 0 prepared once, every
    wavefront runs several probes one after the other on a ScratchSmall working set). It must give what the one-wavefront kernel
    with the general Scratch gives (KSOLVE_TEST_SWEEP_GENERAL, test builds only) for every probe: decisions, replacements, where
    every pod went, NodeClaims, evaluation counts — 240 single-node probes and 40 multi-node sets, so that each emulated wavefront
    goes through ~20 probes on the same working set. A probe handle (one simulation through ksolve_solve) gives the sweep's result too."""
    import random
    cc = dz.make_resident_cluster(n_nodes=600, seed=7, topology=topology)
    order = dz.compact_candidates(cc)
    rng = random.Random(3)
    singles = [[cc["nodes"][i]] for i in order[:240]]
    multis = [[cc["nodes"][i] for i in rng.sample(order, k)] for k in range(2, 42)]
    outs = []
    for general in (False, True):
        if general:
            monkeypatch.setenv("KSOLVE_TEST_SWEEP_GENERAL", "1")
        else:
            monkeypatch.delenv("KSOLVE_TEST_SWEEP_GENERAL", raising=False)
        rc = dz.ResidentCluster.from_compact(cc, solver_lib=emu)
        a = rc.decisions(singles, detail=True); la = rc.last_sweep
        b = rc.decisions(multis, detail=True, multi_node=True, library_prices=True); lb = rc.last_sweep
        keys = ("decisions", "allNonPendingPodsScheduled", "claims", "status", "referenceBinEvaluations")
        outs.append((a, b, [la[k] for k in keys], [lb[k] for k in keys]))
        if not general:
            # one simulation as a probe handle of the resident cluster: the pods a sweep schedules for this candidate
            j = next(i for i, c in enumerate(a) if c["decision"] == "replace")
            pr = rc.scheduler.Probe([singles[j][0]["name"]], pods_of_removed_nodes=True)
            res = pr.Solve()
            assert res["counters"]["referenceBinEvaluations"] == la["referenceBinEvaluations"][j]
            assert len(res["newNodeClaims"]) == la["claims"][j] == 1
            pr.close()
        rc.close()
    assert outs[0] == outs[1]
    assert len({c["decision"] for c in outs[0][0]}) == 3
    # ... and as several launches (an arena budget of 8 MB per launch: every chunk has its own hand-out order and counter)
    monkeypatch.delenv("KSOLVE_TEST_SWEEP_GENERAL", raising=False)
    monkeypatch.setenv("KSOLVE_SWEEP_ARENA_MB", "8")
    rc = dz.ResidentCluster.from_compact(cc, solver_lib=emu)
    a = rc.decisions(singles, detail=True)
    assert a == outs[0][0]
    rc.close()
    # ... and on a device that REFUSES the arena of the whole sweep (the refusal was sticky on the HIP backend and the
    # retry could never run; the emulation refuses above KSOLVE_TEST_ARENA_LIMIT_MB): the sweep goes on with half the probes per
    # launch until the arena fits, the handle stays usable, the verdicts are those of one launch
    monkeypatch.delenv("KSOLVE_SWEEP_ARENA_MB")
    monkeypatch.setenv("KSOLVE_TEST_ARENA_LIMIT_MB", "6")
    rc = dz.ResidentCluster.from_compact(cc, solver_lib=emu)
    a = rc.decisions(singles, detail=True)
    assert a == outs[0][0]
    assert rc.decisions(singles[:5], detail=True) == outs[0][0][:5]     # a later, smaller sweep of the same handle
    rc.close()


def test_same_instance_type_filter_in_the_sweep(oracle, emu):
    """filterOutSameInstanceType (multinodeconsolidation.go:209-246) inside the sweep's verdicts. [2 x the priciest type, 1 x a
    small type] -> one node: the small type is among the replacement options, so only options cheaper than the small node
    survive; when nothing is cheaper the prefix is no command at all, while plain computeConsolidation (single-node, validation)
    keeps its unfiltered list."""
    its = fx.fake_instance_types_assorted()
    od = lambda t: [o for o in t["offerings"] if dz._capacity_type(o) == "on-demand"]
    priciest, offering = max(((t, o) for t in its for o in od(t)), key=lambda x: x[1]["price"])
    zone = [r["values"][0] for r in offering["requirements"] if r["key"] == fx.ZONE][0]
    in_zone = lambda t: [o for o in od(t) if [r["values"][0] for r in o["requirements"] if r["key"] == fx.ZONE][0] == zone and o.get("available", True)]
    fits = sorted((t for t in its if in_zone(t) and int(t["capacity"]["cpu"]) >= 2), key=lambda t: in_zone(t)[0]["price"])
    changed = 0
    for small in (fits[0], fits[len(fits) // 3]):
        nodes = [_node_with_pods("node-0", priciest, zone, "on-demand", ["100m"]), _node_with_pods("node-1", priciest, zone, "on-demand", ["100m"]),
                 _node_with_pods("node-2", small, zone, "on-demand", ["100m"])]
        cluster = {"instanceTypes": its, "nodePools": [fx.node_pool()], "nodes": nodes, "pendingPods": []}
        cands = dz.sort_candidates(cluster, nodes)
        rc = dz.ResidentCluster(cluster, cands, solver_lib=emu)
        a, pa = rc.first_n(cands)
        b, pb = dz.first_n_consolidation_option(cluster, cands, oracle.solve)
        assert pa == pb and strip(a) == strip(b)
        plain, filtered = rc.decisions([cands], multi_node=False)[0], rc.decisions([cands], multi_node=True)[0]
        assert plain["decision"] == dz.REPLACE and small["name"] in plain["replacement"]
        assert filtered["decision"] == dz.NOOP or (small["name"] not in filtered["replacement"] and set(filtered["replacement"]) < set(plain["replacement"]))
        changed += filtered != plain
        rc.close()
    assert changed == 2


def test_probe_api_rejects_bad_descriptors(emu):
    cluster = dz.make_cluster(n_nodes=6, pods_per_node=2, seed=2)
    rc = dz.ResidentCluster(cluster, cluster["nodes"][:3], solver_lib=emu)
    uid = cluster["nodes"][0]["pods"][0]["uid"]
    with pytest.raises(RuntimeError):
        rc.scheduler.Probe(["no-such-node"], [uid])
    with pytest.raises(RuntimeError):
        rc.scheduler.Probe([cluster["nodes"][0]["name"]], [uid, uid])
    with pytest.raises(RuntimeError):
        rc.scheduler.Probe([], ["no-such-pod"])
    ok = rc.scheduler.Probe([cluster["nodes"][0]["name"]], [uid]).Solve()
    assert ok["counters"]["pods"] == 1 and ok["counters"]["existingNodes"] == 5
    rc.close()


def test_resident_cluster_probes_with_volume_limits(oracle, emu):
    """CSI attach limits of the cluster's nodes (VolumeUsage, existingnode.go:88, :179) inside probes of a resident cluster:
    the displaced pods' claims count against the limits of the nodes they move to, each probe with its own log of what it
    added (the nodes' tables are shared and pristine)."""
    import random
    rng = random.Random(77)
    cluster = _tight_cluster(5)
    csi = "ebs.csi"
    for n in cluster["nodes"]:
        vols = []
        for p in n["pods"]:
            if rng.random() < 0.7:
                p["volumes"] = [{"driver": csi, "pvc": f"default/pvc-{rng.randrange(40)}"} for _ in range(rng.choice([1, 2]))]
                vols += p["volumes"]
        n["volumeUsage"] = {"volumes": vols, "limits": {csi: len({v["pvc"] for v in vols}) + rng.choice([0, 0, 1])}}
    cands = [n for n in dz.sort_candidates(cluster, cluster["nodes"]) if not n.get("markedForDeletion")][:12]
    got, rc = dz.sweep_resident(cluster, cands, solver_lib=emu)
    want = dz.sweep(cluster, cands, oracle.solve)
    assert [strip(c) for c in got] == [strip(c) for c in want]
    for g, w in zip(got, want):
        parity.assert_same_results(g["results"], w["results"])
    fast = rc.decisions([[c] for c in cands])
    assert [(c["decision"], c["replacement"]) for c in fast] == [(c["decision"], c["replacement"]) for c in want]
    rc.close()
    # the limits decide something: without them some displaced pods land on other nodes
    for n in cluster["nodes"]:
        n["volumeUsage"]["limits"] = {}
    free = dz.sweep(cluster, cands, oracle.solve)
    placed = lambda cmds: [sorted((e["name"], tuple(e["pods"])) for e in c["results"]["existingNodes"] if e["pods"]) for c in cmds]
    assert placed(free) != placed(want)


def _topology_cluster(seed, n_nodes=40):
    """A cluster whose bound pods carry topology constraints: zonal and hostname spread, zonal self-affinity, hostname
    anti-affinity (a cluster pod's anti-affinity is an inverse group for everything that moves), one pod per node without any."""
    import random
    rng = random.Random(seed)
    cluster = dz.make_cluster(n_nodes=n_nodes, pods_per_node=4, seed=seed, utilisation=0.6)
    web, db, batch = {"app": "web"}, {"app": "db"}, {"app": "batch"}
    for n in cluster["nodes"]:
        for j, p in enumerate(n["pods"]):
            r = rng.random()
            if r < 0.25:
                p["labels"] = web; p["topologySpreadConstraints"] = [fx.spread(fx.ZONE, web, max_skew=rng.choice([1, 2]))]
            elif r < 0.45:
                p["labels"] = batch; p["topologySpreadConstraints"] = [fx.spread(fx.HOSTNAME, batch, max_skew=rng.choice([2, 3]))]
            elif r < 0.55:
                p["labels"] = db; p["podAntiAffinity"] = {"required": [fx.affinity_term(fx.HOSTNAME, db)]}
            elif r < 0.65:
                p["labels"] = {"app": "cache"}; p["podAffinity"] = {"required": [fx.affinity_term(fx.ZONE, {"app": "cache"})]}
            elif r < 0.7:
                p["labels"] = web      # counted by the web spread, constrains nothing itself
    return cluster


@pytest.mark.parametrize("seed", [11, 12, 13])
def test_resident_cluster_probes_with_topology(oracle, emu, seed):
    """Probes of a resident cluster whose pods carry topology constraints: the cluster is counted ONCE on the device (every bound
    pod, every node's domains) and each probe takes its candidates' share out — the removed nodes' per-node counters and domain
    registrations, the displaced pods' counts in the groups that select them and in the inverse anti-affinity groups they own —
    which is what NewTopology / countDomains give for the simulation (topology.go:68-103, :310-355, :361-459). Compared, probe by
    probe, with the oracle solving SimulateScheduling assembled from scratch (staying pods as cluster pods)."""
    import random
    cluster = _topology_cluster(seed)
    cands = [n for n in dz.sort_candidates(cluster, cluster["nodes"]) if not n.get("markedForDeletion")][:12]
    got, rc = dz.sweep_resident(cluster, cands, solver_lib=emu)
    want = dz.sweep(cluster, cands, oracle.solve)
    for g, w in zip(got, want):
        parity.assert_same_results(g["results"], w["results"])
        assert g["results"]["counters"]["referenceBinEvaluations"] == w["results"]["counters"]["binEvaluations"]
    assert [strip(c) for c in got] == [strip(c) for c in want]
    fast = rc.decisions([[c] for c in cands])
    assert [(c["decision"], c["replacement"]) for c in fast] == [(c["decision"], c["replacement"]) for c in want]
    rng = random.Random(seed)
    sets = [rng.sample(cands, k) for k in (2, 4)]
    rc.prefetch(sets)
    for cs in sets:
        g, w = dz.compute_consolidation(cluster, cs, rc), dz.compute_consolidation(cluster, cs, oracle.solve)
        parity.assert_same_results(g["results"], w["results"])
        assert strip(g) == strip(w)
    rc.close()


@pytest.mark.parametrize("seed", [100, 103, 107])
def test_probes_with_hostname_affinity_groups_and_shared_node_counters(oracle, emu, seed):
    """Probes of a resident cluster keep the cluster's per-node counters of hostname groups SHARED (round 4: what a probe's commits
    add lives in the overlay slot of the node, a removed node counts nothing) instead of copying a counter per node and group.
    Every kind of hostname group is in the cluster — spread, anti-affinity, self-affinity (with its bootstrap and
    anyCompatiblePodDomain, topologygroup.go:324-400) and affinity to another workload — and the verdict of every single-node probe
    and of a few multi-node prefixes equals the oracle's own (simulation and decision)."""
    import random
    rng = random.Random(seed)
    cluster = _topology_cluster(seed, n_nodes=30)
    for n in cluster["nodes"]:
        for p in n["pods"]:
            if "topologySpreadConstraints" in p or "podAntiAffinity" in p or "podAffinity" in p:
                continue
            r = rng.random()
            if r < 0.3:
                p["labels"] = {"app": "pair"}; p["podAffinity"] = {"required": [fx.affinity_term(fx.HOSTNAME, {"app": "pair"})]}
            elif r < 0.45:
                p["labels"] = {"app": "follower"}; p["podAffinity"] = {"required": [fx.affinity_term(fx.HOSTNAME, {"app": "batch"})]}
    cands = dz.sort_candidates(cluster, cluster["nodes"])[:12]
    rc = dz.ResidentCluster(cluster, cands, solver_lib=emu)
    sets = [[c] for c in cands] + [cands[:k] for k in (2, 3, 5)]
    cmds = rc.decisions(sets, multi_node=True)
    want = oracle.cluster_verdicts(cluster, sets, multi_node=True, well_known=fx.KWOK_WELL_KNOWN)
    for i, (c, v) in enumerate(zip(cmds, want)):
        assert (c["decision"], c["replacement"], c.get("replacementCapacityType")) == oracle.verdict_key(v), (seed, i, c, v)
    rc.close()
