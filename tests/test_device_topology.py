"""CPU tests of the device topology path (topology spread, pod affinity / anti-affinity, inverse
anti-affinity, TopologyNodeFilter, relaxation of preferred terms): the product's engine compiled for the host (test
infrastructure, tests/emu) behind the real C ABI and host flattener, compared with the oracle claim by claim. Scenario
shapes follow pkg/controllers/provisioning/scheduling/topology_test.go; the GPU run is tests/test_gpu_parity.py."""
import random

import pytest

import parity
from karpenter_amd import fixtures as fx
from karpenter_amd.scheduling import NewScheduler
from test_device_algorithm import check, emu  # noqa: F401  (fixture)

LAB = {"test": "test"}


def solve_both(oracle, emu, pods, its=None, pools=None, **kw):
    prob = fx.problem(its if its is not None else fx.fake_default_instance_types(), pools or [fx.node_pool()], pods, **kw)
    return check(oracle, emu, prob)


def test_zonal_spread(oracle, emu):
    # topology_test.go:110-124
    got, _ = solve_both(oracle, emu, [fx.pod(labels=LAB, topology_spread=[fx.spread(fx.ZONE, LAB)]) for _ in range(4)])
    assert not got["podErrors"]
    # maxSkew 2, min domains, an unsatisfiable ScheduleAnyway constraint that gets relaxed away
    solve_both(oracle, emu, [fx.pod(labels=LAB, topology_spread=[fx.spread(fx.ZONE, LAB, max_skew=2)]) for _ in range(11)])
    solve_both(oracle, emu, [fx.pod(labels=LAB, topology_spread=[fx.spread(fx.ZONE, LAB, min_domains=5)]) for _ in range(6)])
    solve_both(oracle, emu, [fx.pod(labels=LAB, node_selector={fx.ZONE: "test-zone-1"},
                                    topology_spread=[fx.spread(fx.ZONE, LAB, when="ScheduleAnyway", affinity_policy="Ignore")]) for _ in range(5)])


def test_zonal_spread_limited_by_selector_and_nodepool(oracle, emu):
    # topology_test.go:188-260: pods restricted to a subset of zones only count / use those zones
    pods = [fx.pod(labels=LAB, topology_spread=[fx.spread(fx.ZONE, LAB)],
                   node_requirements=[fx.req(fx.ZONE, "In", ["test-zone-1", "test-zone-2"])]) for _ in range(6)]
    solve_both(oracle, emu, pods)
    pools = [fx.node_pool(requirements=[fx.req(fx.ZONE, "In", ["test-zone-2", "test-zone-3"])])]
    solve_both(oracle, emu, [fx.pod(labels=LAB, topology_spread=[fx.spread(fx.ZONE, LAB)]) for _ in range(7)], pools=pools)
    # capacity-type spread (topology_test.go:655-760)
    solve_both(oracle, emu, [fx.pod(labels=LAB, topology_spread=[fx.spread(fx.CAPACITY_TYPE, LAB)]) for _ in range(5)])


def test_hostname_spread_and_anti_affinity(oracle, emu):
    got, _ = solve_both(oracle, emu, [fx.pod(labels=LAB, topology_spread=[fx.spread(fx.HOSTNAME, LAB)]) for _ in range(4)])
    assert sorted(len(c["pods"]) for c in got["newNodeClaims"]) == [1, 1, 1, 1]
    solve_both(oracle, emu, [fx.pod(labels=LAB, topology_spread=[fx.spread(fx.HOSTNAME, LAB, max_skew=3)]) for _ in range(10)])
    got, _ = solve_both(oracle, emu, [fx.pod(labels=LAB, pod_anti_requirements=[fx.affinity_term(fx.HOSTNAME, LAB)]) for _ in range(3)])
    assert len(got["newNodeClaims"]) == 3
    # Schrödinger (topology_test.go:2502-2531) and inverse anti-affinity (:2466-2500)
    got, _ = solve_both(oracle, emu, [fx.pod(labels=LAB, pod_anti_requirements=[fx.affinity_term(fx.ZONE, LAB)]) for _ in range(5)])
    assert len(got["newNodeClaims"]) == 1 and len(got["podErrors"]) == 4
    aff = {"security": "s2"}
    anti = [fx.affinity_term(fx.ZONE, aff)]
    zp = [fx.pod(requests={"cpu": "2"}, pod_anti_requirements=anti, node_selector={fx.ZONE: f"test-zone-{i}"}) for i in (1, 2, 3)]
    victim = fx.pod(labels=aff)
    got, _ = solve_both(oracle, emu, zp + [victim])
    assert list(got["podErrors"]) == [victim["uid"]]


def test_pod_affinity(oracle, emu):
    lab = {"app": "a"}
    got, _ = solve_both(oracle, emu, [fx.pod(labels=lab, requests={"cpu": "1"}, pod_requirements=[fx.affinity_term(fx.ZONE, lab)]) for _ in range(6)])
    assert not got["podErrors"]
    # hostname self-affinity: everything lands on one claim until it is full
    solve_both(oracle, emu, [fx.pod(labels=lab, requests={"cpu": "1"}, pod_requirements=[fx.affinity_term(fx.HOSTNAME, lab)]) for _ in range(9)])
    # affinity to another deployment that is pinned to a zone (topology_test.go:1900-1960)
    target = [fx.pod(labels={"app": "db"}, node_selector={fx.ZONE: "test-zone-2"}, requests={"cpu": "2"}) for _ in range(2)]
    followers = [fx.pod(labels={"app": "web"}, pod_requirements=[fx.affinity_term(fx.ZONE, {"app": "db"})]) for _ in range(5)]
    solve_both(oracle, emu, target + followers)
    # affinity to something that does not exist: unschedulable
    got, _ = solve_both(oracle, emu, [fx.pod(pod_requirements=[fx.affinity_term(fx.ZONE, {"app": "nope"})]) for _ in range(2)])
    assert len(got["podErrors"]) == 2
    # preferred affinity / anti-affinity terms are relaxed away one by one
    solve_both(oracle, emu, [fx.pod(labels=lab, pod_preferences=[fx.weighted(10, fx.affinity_term(fx.ZONE, {"app": "nope"}))],
                                    pod_anti_preferences=[fx.weighted(5, fx.affinity_term(fx.HOSTNAME, lab))]) for _ in range(4)])


def test_existing_nodes_and_cluster_pods(oracle, emu):
    # pods already running in the cluster seed the domain counts (countDomains, topology.go:361-459) and inverse
    # anti-affinity groups (topology.go:310-324)
    its = fx.fake_default_instance_types()
    nodes, cluster = [], []
    for i, zone in enumerate(["test-zone-1", "test-zone-1", "test-zone-2"]):
        n = fx.state_node(f"node-{i}", its[2], zone, "on-demand", "default", used={"cpu": "1", "pods": "1"})
        nodes.append(n)
        cluster.append(fx.pod(labels=LAB, phase="Running", node_name=f"node-{i}", requests={"cpu": "1"}))
    cluster.append(fx.pod(labels={"role": "guard"}, phase="Running", node_name="node-2",
                          pod_anti_requirements=[fx.affinity_term(fx.ZONE, {"role": "intruder"})]))
    pods = [fx.pod(labels=LAB, topology_spread=[fx.spread(fx.ZONE, LAB)]) for _ in range(5)]
    pods += [fx.pod(labels=LAB, topology_spread=[fx.spread(fx.HOSTNAME, LAB, max_skew=2)]) for _ in range(4)]
    pods += [fx.pod(labels={"role": "intruder"}) for _ in range(2)]
    solve_both(oracle, emu, pods, its=its, state_nodes=nodes, cluster_pods=cluster)


def test_taint_and_affinity_policies(oracle, emu):
    # nodeTaintsPolicy=Honor hides the domains a pod cannot tolerate; nodeAffinityPolicy=Ignore counts every node
    pools = [fx.node_pool("tainted", requirements=[fx.req(fx.ZONE, "In", ["test-zone-3"])], taints=[{"key": "x", "value": "y", "effect": "NoSchedule"}], weight=5),
             fx.node_pool("open", requirements=[fx.req(fx.ZONE, "In", ["test-zone-1", "test-zone-2"])])]
    for policy in (None, "Honor", "Ignore"):
        pods = [fx.pod(labels=LAB, topology_spread=[fx.spread(fx.ZONE, LAB, taints_policy=policy)]) for _ in range(5)]
        solve_both(oracle, emu, pods, pools=pools)
    pods = [fx.pod(labels=LAB, node_selector={fx.ZONE: "test-zone-1"}, topology_spread=[fx.spread(fx.ZONE, LAB, affinity_policy="Ignore")]) for _ in range(3)]
    solve_both(oracle, emu, pods)


@pytest.mark.parametrize("seed", range(12))
def test_topology_fuzz(oracle, emu, seed):
    """Random mixes of the benchmark's constraint kinds (scheduling_benchmark_test.go:240-330) on a small catalogue, so
    that constraints interact: shared selectors across kinds, limited zones, preferred terms, several NodePools."""
    rng = random.Random(1000 + seed)
    labels = [{"my-label": c} for c in "abc"]
    res = lambda: {"cpu": f"{rng.choice([100, 250, 500, 1000, 1500])}m", "memory": f"{rng.choice([100, 256, 512, 1024])}Mi"}
    pods = []
    for _ in range(rng.randrange(30, 90)):
        kind = rng.randrange(8)
        lab = rng.choice(labels)
        sel = rng.choice(labels)
        kw = dict(labels=lab, requests=res())
        if kind == 0:
            kw["topology_spread"] = [fx.spread(fx.ZONE, sel, max_skew=rng.choice([1, 1, 2]))]
        elif kind == 1:
            kw["topology_spread"] = [fx.spread(fx.HOSTNAME, sel, max_skew=rng.choice([1, 2, 4]))]
        elif kind == 2:
            kw["pod_requirements"] = [fx.affinity_term(rng.choice([fx.ZONE, fx.HOSTNAME]), sel)]
        elif kind == 3:
            kw["pod_anti_requirements"] = [fx.affinity_term(rng.choice([fx.HOSTNAME, fx.HOSTNAME, fx.ZONE]), sel)]
        elif kind == 4:
            kw["topology_spread"] = [fx.spread(fx.ZONE, sel), fx.spread(fx.HOSTNAME, sel, max_skew=3, when=rng.choice(["DoNotSchedule", "ScheduleAnyway"]))]
        elif kind == 5:
            kw["pod_preferences"] = [fx.weighted(rng.randrange(1, 100), fx.affinity_term(fx.ZONE, sel))]
            kw["pod_anti_preferences"] = [fx.weighted(rng.randrange(1, 100), fx.affinity_term(fx.HOSTNAME, sel))]
        elif kind == 6:
            kw["node_selector"] = {fx.ZONE: rng.choice(["test-zone-1", "test-zone-2", "test-zone-3"])}
            if rng.random() < 0.5:
                kw["topology_spread"] = [fx.spread(fx.CAPACITY_TYPE, sel)]
        pods.append(fx.pod(**kw))
    pools = [fx.node_pool()]
    if seed % 3 == 1:
        pools = [fx.node_pool("a", requirements=[fx.req(fx.ZONE, "In", ["test-zone-1", "test-zone-2"])], weight=10), fx.node_pool("b")]
    its = fx.fake_instance_types(12) if seed % 2 else fx.fake_default_instance_types()
    check(oracle, emu, fx.problem(its, pools, pods))


def test_same_hash_groups_created_by_relaxation(oracle, emu):
    """TopologyGroup.Hash() covers the node filter's KEYS, not its values (topologygroup.go:188-222), and Topology.Update
    reuses whatever group it finds under the hash (topology.go:162-194). Two workloads whose first node-affinity term is
    unsatisfiable relax to filters over the same key with different zones: whichever relaxes first creates the group and
    the other joins it, so the outcome depends on the queue order. Both orders must match the oracle."""
    def workload(cpu, zones, n):
        terms = [[fx.req("example.com/unknown", "In", "x")], [fx.req(fx.ZONE, "In", *zones)]]
        return [fx.pod(labels=LAB, requests={"cpu": cpu}, node_requirements=terms, topology_spread=[fx.spread(fx.ZONE, LAB)]) for _ in range(n)]
    seen = []
    for first, second in ((["test-zone-1", "test-zone-2"], ["test-zone-2", "test-zone-3"]), (["test-zone-2", "test-zone-3"], ["test-zone-1", "test-zone-2"])):
        pods = workload("1", first, 5) + workload("500m", second, 5)     # the larger pods pop (and relax) first
        got, _ = solve_both(oracle, emu, pods)
        assert got["counters"]["topologyAliasClasses"] == 1 and got["counters"]["relaxations"] >= 2
        seen.append(sorted((c["requirements"] and [r["values"] for r in c["requirements"] if r["key"] == fx.ZONE][0], len(c["pods"])) for c in got["newNodeClaims"]))
    assert seen[0] != seen[1]
    # three creators, one of them joining an identical existing member
    pods = workload("2", ["test-zone-1"], 3) + workload("1", ["test-zone-3"], 3) + workload("500m", ["test-zone-1"], 3) + workload("250m", ["test-zone-2", "test-zone-3"], 3)
    got, _ = solve_both(oracle, emu, pods)
    assert got["counters"]["topologyAliasClasses"] == 1
