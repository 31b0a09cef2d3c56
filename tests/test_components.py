"""BASELINE configs[3] shape (C4: many NodePools, every pod pinned to one): the whole batch and each NodePool component are
solved bit-exactly by the device algorithm; the split itself is only equal in quality to the whole-batch Solve(), not in
pod identities — the coupling through the unstable claim sort is pinned here so that nobody "optimises" the multi-GPU path
into sharding a single Solve() and calling it parity (karpenter_amd/components.py)."""
import pytest

import parity
from karpenter_amd import fixtures as fx
from karpenter_amd.components import split_by_nodepool, split_components
from karpenter_amd.scheduling import NewScheduler, SolveBatch
from test_device_algorithm import emu  # noqa: F401  (fixture)


def test_config4_whole_and_per_component(oracle, emu):
    prob = fx.config4(pods=20000, n_types=144, n_pools=16, seed=2)
    whole = oracle.solve(prob)
    parity.assert_same_results(NewScheduler(prob, solver_lib=emu).Solve(), whole)            # one Solve(), one wavefront: exact
    parts = split_by_nodepool(prob)
    assert [name for name, _ in parts] == [np_["name"] for np_ in prob["nodePools"]]
    # the components go to the device as one batched launch (or one per GPU rank); each is exact against the oracle
    got = SolveBatch([NewScheduler(sub, solver_lib=emu) for _, sub in parts])
    want = [oracle.solve(sub) for _, sub in parts]
    for g, w in zip(got, want):
        parity.assert_same_results(g, w)
    # quality: every pod placed, about the same number of NodeClaims, cost within half a percent of the whole-batch solve
    assert not whole["podErrors"] and not any(w["podErrors"] for w in want)
    n_split = sum(len(w["newNodeClaims"]) for w in want)
    assert abs(n_split - len(whole["newNodeClaims"])) <= 0.03 * len(whole["newNodeClaims"])
    split_cost = sum(w["packingCost"] for w in want)
    assert abs(split_cost - whole["packingCost"]) <= 0.005 * whole["packingCost"]
    # identity: NOT the same pods per claim — the pools are coupled through the position of each other's claims in the
    # array the reference re-sorts (scheduler.go:598); this is why exact scaling shards across Solve() calls instead
    differing = 0
    for (name, _), w in zip(parts, want):
        a = sorted(tuple(c["pods"]) for c in whole["newNodeClaims"] if c["nodePool"] == name)
        b = sorted(tuple(c["pods"]) for c in w["newNodeClaims"])
        differing += a != b
    assert differing > 0


def test_split_refuses_what_it_cannot_prove(oracle):
    base = fx.config4(pods=400, n_types=50, n_pools=2, seed=1)
    assert len(split_by_nodepool(base)) == 2
    lab = {"app": "x"}
    unpinned = dict(base, pods=[fx.pod()])
    spread = dict(base, pods=[fx.pod(labels=lab, node_selector={fx.NODEPOOL: "pool-00"}, topology_spread=[fx.spread(fx.ZONE, lab)])])
    unknown = dict(base, pods=[fx.pod(node_selector={fx.NODEPOOL: "no-such-pool"})])
    with_nodes = dict(base, stateNodes=[{"name": "n"}])
    reserved = dict(base, options={"reservedCapacity": True})
    for prob in (unpinned, spread, unknown, with_nodes, reserved):
        assert split_by_nodepool(prob) is None


def test_connected_components_of_pods_and_pools(oracle, emu):
    """Pods that may land on either of two NodePools (a required node-affinity term per pool, or one `In [a, b]`) tie those
    pools into one component; the other pools stay on their own. Every component is solved exactly as its own problem and
    the union places every pod at about the whole-batch cost."""
    prob = fx.config4(pods=6000, n_types=100, n_pools=6, seed=5)
    pin = lambda *pools: [fx.req(fx.NODEPOOL, "In", *pools)]
    extra = [fx.pod(requests={"cpu": "1", "memory": "1Gi"}, node_requirements=pin("pool-00", "pool-01")) for _ in range(40)]
    extra += [fx.pod(requests={"cpu": "2", "memory": "1Gi"}, node_requirements=[pin("pool-03"), pin("pool-04")]) for _ in range(40)]   # OR-ed terms
    extra += [fx.pod(requests={"cpu": "500m"}, node_selector={fx.NODEPOOL: "pool-04"}, node_requirements=pin("pool-04", "pool-05")) for _ in range(10)]   # selector AND affinity: pool-04 only
    prob = dict(prob, pods=prob.get("pods", []) + extra)
    parts = split_components(prob)
    assert [pools for pools, _ in parts] == [("pool-00", "pool-01"), ("pool-02",), ("pool-03", "pool-04"), ("pool-05",)]
    n_pods = lambda pr: len(pr.get("pods", [])) + sum(g["count"] for g in pr.get("podGroups", []))
    assert sum(n_pods(sub) for _, sub in parts) == n_pods(prob)
    got = SolveBatch([NewScheduler(sub, solver_lib=emu) for _, sub in parts])
    want = [oracle.solve(sub) for _, sub in parts]
    for g, w in zip(got, want):
        parity.assert_same_results(g, w)
    whole = oracle.solve(prob)
    assert not whole["podErrors"] and not any(w["podErrors"] for w in want)
    cost = sum(w["packingCost"] for w in want)
    assert abs(cost - whole["packingCost"]) <= 0.03 * whole["packingCost"]      # small batch: ~1% either way (here the split is cheaper)
    # a split by single NodePool must refuse this batch; the component split refuses what it cannot bound either
    assert split_by_nodepool(prob) is None
    loose = dict(prob, pods=prob["pods"] + [fx.pod(node_requirements=[pin("pool-00"), [fx.req(fx.ZONE, "In", "test-zone-1")]])])   # second term reaches any pool
    assert split_components(loose) is None
    assert split_components(dict(prob, pods=[fx.pod(node_requirements=pin("no-such-pool"))])) is None


def test_components_through_topology_groups(oracle, emu):
    """Pods with spread constraints / pod (anti-)affinity stay splittable as long as every topology group lives inside one
    component: a group ties its owner to every pod its namespaces + selector match (topologygroup.go:442), so pools whose
    pods are selected by one group merge, the others stay apart. Each component is solved exactly as its own problem."""
    prob = fx.config4(pods=3000, n_types=100, n_pools=5, seed=7)
    on = lambda pool: {fx.NODEPOOL: pool}
    web, db, cache = {"app": "web"}, {"app": "db"}, {"app": "cache"}
    extra = [fx.pod(labels=web, node_selector=on("pool-00"), requests={"cpu": "500m"}, topology_spread=[fx.spread(fx.HOSTNAME, web, max_skew=2)]) for _ in range(12)]       # spread inside pool-00
    extra += [fx.pod(labels=db, node_selector=on("pool-01"), requests={"cpu": "1"}, pod_anti_requirements=[fx.affinity_term(fx.HOSTNAME, db)]) for _ in range(6)]   # anti-affinity inside pool-01
    extra += [fx.pod(labels=cache, node_selector=on("pool-02"), requests={"cpu": "250m"}) for _ in range(8)]
    extra += [fx.pod(labels={"app": "api"}, node_selector=on("pool-03"), requests={"cpu": "250m"}, pod_anti_requirements=[fx.affinity_term(fx.HOSTNAME, cache)]) for _ in range(5)]   # pool-03's pods avoid pool-02's
    extra += [fx.pod(labels={"app": "batch"}, node_selector=on("pool-04"), pod_anti_preferences=[fx.weighted(5, fx.affinity_term(fx.HOSTNAME, {"app": "nothing"}))]) for _ in range(4)]  # selects nobody
    prob = dict(prob, pods=prob.get("pods", []) + extra)
    parts = split_components(prob)
    assert [pools for pools, _ in parts] == [("pool-00",), ("pool-01",), ("pool-02", "pool-03"), ("pool-04",)]
    got = SolveBatch([NewScheduler(sub, solver_lib=emu) for _, sub in parts])
    want = [oracle.solve(sub) for _, sub in parts]
    for g, w in zip(got, want):
        parity.assert_same_results(g, w)
    whole = oracle.solve(prob)
    assert not whole["podErrors"] and not any(w["podErrors"] for w in want)
    assert abs(sum(len(w["newNodeClaims"]) for w in want) - len(whole["newNodeClaims"])) <= max(2, 0.03 * len(whole["newNodeClaims"]))
    assert abs(sum(w["packingCost"] for w in want) - whole["packingCost"]) <= 0.02 * whole["packingCost"]
    # a selector that reaches pods of another pool merges the two; what cannot be evaluated here is refused
    reach = dict(prob, pods=prob["pods"] + [fx.pod(labels=web, node_selector=on("pool-04"))])
    assert [pools for pools, _ in split_components(reach)][0] == ("pool-00", "pool-04")
    expr = fx.pod(node_selector=on("pool-00"), topology_spread=[dict(fx.spread(fx.HOSTNAME, web), labelSelector={"matchExpressions": [{"key": "app", "operator": "Exists"}]})])
    assert split_components(dict(prob, pods=prob["pods"] + [expr])) is None
    nssel = fx.pod(node_selector=on("pool-00"), pod_anti_requirements=[fx.affinity_term(fx.HOSTNAME, web, namespace_selector={})])
    assert split_components(dict(prob, pods=prob["pods"] + [nssel])) is None
    # a group on any key but the hostname sees the domains of EVERY NodePool (below): refused
    zonal = fx.pod(labels=web, node_selector=on("pool-00"), topology_spread=[fx.spread(fx.ZONE, web)])
    assert split_components(dict(prob, pods=prob["pods"] + [zonal])) is None
    zaff = fx.pod(labels=web, node_selector=on("pool-00"), pod_requirements=[fx.affinity_term(fx.ZONE, web)])
    assert split_components(dict(prob, pods=prob["pods"] + [zaff])) is None


def test_a_zonal_group_sees_the_domains_of_every_nodepool(oracle):
    """the reference builds a group's domain universe over ALL NodePools (topology.go:104-142 buildDomainGroups) and
    domainMinCount takes the minimum over every domain of the key (topologygroup.go:300-322). Ten pods pinned to pool-a (zones 1-2)
    with a zonal spread of maxSkew 1, pool-b offering zone 3: the whole batch leaves eight of them unschedulable (zone 3 stays at
    zero and nothing may get more than one ahead of it) — a component that only knows pool-a's zones would schedule all ten. The
    split must refuse such a batch, in the host library and in the Python rule."""
    from karpenter_amd.components import split_components_reference
    lab = {"app": "web"}
    pools = [fx.node_pool("pool-a", requirements=[fx.req(fx.ZONE, "In", "test-zone-1", "test-zone-2")]),
             fx.node_pool("pool-b", requirements=[fx.req(fx.ZONE, "In", "test-zone-3")])]
    pods = [fx.pod(labels=lab, requests={"cpu": "500m"}, node_selector={fx.NODEPOOL: "pool-a"}, topology_spread=[fx.spread(fx.ZONE, lab)]) for _ in range(10)]
    pods += [fx.pod(requests={"cpu": "500m"}, node_selector={fx.NODEPOOL: "pool-b"}) for _ in range(3)]
    prob = fx.problem(fx.fake_instance_types(12), pools, pods)
    whole = oracle.solve(prob)
    assert len(whole["podErrors"]) == 8
    alone = oracle.solve(dict(prob, nodePools=pools[:1], pods=pods[:10]))
    assert not alone["podErrors"]          # what a component cut off from pool-b would answer
    assert split_components(prob) is None and split_components_reference(prob) is None


def test_the_host_library_split_equals_the_python_rule_and_deals_by_pod_count():
    """ksched_split_components (karpenter_amd/host/ksched.cpp: what a Go controller calls, round-4 review item 7) against the same
    rule written in Python (components.split_components_reference) on batches with pins, OR-ed affinity terms and topology
    selectors — same components, same sub-problems — and its deal of the components over N devices: every component on exactly one
    device, by pod count, largest first, never worse than 4/3 of the best possible load (the LPT bound)."""
    from karpenter_amd.components import split_components_reference
    prob = fx.config4(pods=6000, n_types=100, n_pools=6, seed=5)
    pin = lambda *pools: [fx.req(fx.NODEPOOL, "In", *pools)]
    on = lambda pool: {fx.NODEPOOL: pool}
    web = {"app": "web"}
    extra = [fx.pod(requests={"cpu": "1"}, node_requirements=pin("pool-00", "pool-01")) for _ in range(10)]
    extra += [fx.pod(labels=web, node_selector=on("pool-02"), topology_spread=[fx.spread(fx.HOSTNAME, web)]) for _ in range(6)]
    extra += [fx.pod(labels={"app": "api"}, node_selector=on("pool-03"), pod_anti_requirements=[fx.affinity_term(fx.HOSTNAME, web)]) for _ in range(3)]
    for batch in (prob, dict(prob, pods=extra), fx.config4(pods=20000, n_types=60, n_pools=16, seed=42)):
        lib, ref = split_components(batch), split_components_reference(batch)
        assert [pools for pools, _ in lib] == [pools for pools, _ in ref]
        for (_, a), (_, b) in zip(lib, ref):
            assert a == b
    # refusals agree too
    for bad in (dict(prob, pods=[fx.pod()]), dict(prob, stateNodes=[{"name": "n"}])):
        assert split_components(bad) is None and split_components_reference(bad) is None
    batch = fx.config4(pods=50000, n_types=60, n_pools=16, seed=42)
    n_pods = lambda pr: len(pr.get("pods", [])) + sum(g["count"] for g in pr.get("podGroups", []))
    for n in (1, 2, 3, 8):
        parts, bins = split_components(batch, bins=n)
        assert sorted(i for b in bins for i in b) == list(range(len(parts))) and len(bins) == n
        loads = [sum(n_pods(parts[i][1]) for i in b) for b in bins]
        assert sum(loads) == 50000
        assert max(loads) <= (4 / 3) * max(50000 / n, max(n_pods(sub) for _, sub in parts)) + 1
