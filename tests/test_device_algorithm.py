"""CPU tests of the DEVICE ALGORITHM: the product's engine / Go-pdqsort emulation / flat requirement algebra compiled
for the host (tests/emu/ksolve_emu.cpp, test infrastructure only) and driven through the real C ABI and the real host
flattener (karpenter_amd/libksched.so), compared with the oracle claim by claim (L1-strict: same claims in the same
order with the same pod identities, instance types, requirements and requests). The GPU run of the same comparisons
is tests/test_gpu_parity.py."""
import copy
import ctypes
import random

import pytest

import parity
from karpenter_amd import fixtures as fx
from karpenter_amd.scheduling import NewScheduler, Unsupported

AMD = {fx.ARCH: "amd64"}


@pytest.fixture(scope="module")
def emu():
    import __graft_entry__  # builds libksched.so (host flattener); the HIP library is not needed for these tests
    import os, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ksched = os.path.join(root, "karpenter_amd", "libksched.so")
    host = os.path.join(root, "karpenter_amd", "host", "ksched.cpp")
    deps = [host, os.path.join(root, "include", "ksolve.h"), os.path.join(root, "karpenter_amd", "csrc", "reqalg.h")]
    if not os.path.exists(ksched) or any(os.path.getmtime(d) > os.path.getmtime(ksched) for d in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", ksched, host, "-ldl"])
    return parity.build_emu()


def check(oracle, emu, prob):
    want = oracle.solve(prob)
    got = NewScheduler(prob, solver_lib=emu).Solve()
    parity.assert_same_results(got, want)
    assert got["counters"]["referenceBinEvaluations"] == want["counters"]["binEvaluations"]  # V
    assert abs(got["packingCost"] - want["packingCost"]) < 1e-9 * max(1.0, want["packingCost"])
    return got, want


def test_config1_kwok_5000_pods(oracle, emu):
    got, _ = check(oracle, emu, fx.config1())
    assert got["scheduledPods"] == 5000 and not got["podErrors"]


@pytest.mark.parametrize("pods,types,seed", [(3000, 100, 7), (12000, 500, 42), (800, 144, 1)])
def test_config2_selectors_taints(oracle, emu, pods, types, seed):
    got, _ = check(oracle, emu, fx.config2(pods=pods, n_types=types, seed=seed))
    assert got["scheduledPods"] == pods


def test_reference_binpacking_cases(oracle, emu):
    its = fx.fake_default_instance_types()
    pods = [fx.pod(requests={"memory": "1.8G"}, node_selector=AMD) for _ in range(40)] + [fx.pod(requests={"memory": "400M"}, node_selector=AMD) for _ in range(20)]
    got, _ = check(oracle, emu, fx.problem(its, [fx.node_pool()], pods))
    assert len(got["newNodeClaims"]) == 20
    check(oracle, emu, fx.problem(its, [fx.node_pool()], [fx.pod(requests={"cpu": "1m", "memory": "1m"}, node_selector=AMD) for _ in range(25)]))
    check(oracle, emu, fx.problem(fx.fake_instance_types(5), [fx.node_pool()], [fx.pod(requests={"cpu": "4.5"}), fx.pod(requests={"cpu": "1"})]))


def test_unschedulable_pods_requeue_and_error_codes(oracle, emu):
    its = fx.fake_default_instance_types()
    pods = [fx.pod(requests={"memory": "2Ti"}), fx.pod(requests={"cpu": "1"}), fx.pod(node_selector={fx.ZONE: "nowhere"}),
            fx.pod(node_requirements=[fx.req("undefined-key", "In", "x")]), fx.pod(requests={"cpu": "100"}), fx.pod(requests={"cpu": "2"})]
    got, want = check(oracle, emu, fx.problem(its, [fx.node_pool()], pods))
    assert len(got["podErrors"]) == 4
    codes = {e["code"] for e in got["podErrors"].values()}
    assert codes == {2, 4}  # incompatible requirements / InstanceTypeFilterError


def test_custom_label_operators_and_gt_lt(oracle, emu):
    key = "test-key"
    for expr in (fx.req(key, "In", "test-value"), fx.req(key, "NotIn", "test-value"), fx.req(key, "Exists"), fx.req(key, "DoesNotExist"),
                 fx.req(key, "In", "another-value"), fx.req(key, "NotIn", "another-value")):
        for labels in ({}, {key: "test-value"}):
            check(oracle, emu, fx.problem(fx.fake_default_instance_types(), [fx.node_pool(labels=labels)], [fx.pod(node_requirements=[expr])]))
    its = fx.fake_instance_types(8)
    for expr in (fx.req(fx.FAKE_INTEGER_LABEL, "Gt", "6"), fx.req(fx.FAKE_INTEGER_LABEL, "Lt", "3"), fx.req(fx.FAKE_INTEGER_LABEL, "Gt", "100"),
                 fx.req(fx.FAKE_INTEGER_LABEL, "NotIn", "2", "3"), fx.req(fx.FAKE_EXOTIC_LABEL, "Exists"), fx.req(fx.FAKE_EXOTIC_LABEL, "DoesNotExist")):
        check(oracle, emu, fx.problem(its, [fx.node_pool()], [fx.pod(node_requirements=[expr]), fx.pod(requests={"cpu": "1"})]))
    # NodePool with a Gt requirement narrowing the integer label, pods with Lt: bounds intersect on the claim
    pool = fx.node_pool(requirements=[fx.req(fx.FAKE_INTEGER_LABEL, "Gt", "2")])
    pods = [fx.pod(node_requirements=[fx.req(fx.FAKE_INTEGER_LABEL, "Lt", "6")]), fx.pod(node_requirements=[fx.req(fx.FAKE_INTEGER_LABEL, "Lt", "3")])]
    check(oracle, emu, fx.problem(its, [pool], pods))


def test_preference_relaxation_ladder(oracle, emu):
    its = fx.fake_default_instance_types()
    pods = [fx.pod(node_requirements=[fx.req(fx.ZONE, "In", "test-zone-3")], node_preferences=[fx.req(fx.ZONE, "In", "invalid")]),
            fx.pod(node_requirements=[[fx.req(fx.ZONE, "In", "invalid")], [fx.req(fx.ZONE, "In", "test-zone-2")]]),
            fx.pod(node_preferences=[{"weight": 1, "matchExpressions": [fx.req(fx.ARCH, "In", "arm64")]}, {"weight": 5, "matchExpressions": [fx.req(fx.ARCH, "In", "sparc")]}]),
            fx.pod(node_requirements=[[fx.req(fx.ZONE, "In", "invalid")], [fx.req(fx.ZONE, "In", "invalid-2")]])]
    got, want = check(oracle, emu, fx.problem(its, [fx.node_pool()], pods))
    assert got["counters"]["relaxations"] == want["counters"]["relaxations"] > 0
    check(oracle, emu, fx.problem(its, [fx.node_pool()], pods, options={"preferencePolicy": "Ignore"}))
    # PreferNoSchedule taints are tolerated only after relaxation (preferences.go:133-146)
    pool = fx.node_pool(taints=[{"key": "soft", "value": "x", "effect": "PreferNoSchedule"}])
    check(oracle, emu, fx.problem(its, [pool], [fx.pod(), fx.pod(requests={"cpu": "1"})]))


def test_taints_weights_and_limits(oracle, emu):
    its = fx.fake_default_instance_types()
    pools = [fx.node_pool("tainted", weight=10, taints=[{"key": "dedicated", "value": "x", "effect": "NoSchedule"}]), fx.node_pool("plain", weight=1)]
    tol = [{"key": "dedicated", "operator": "Exists"}]
    pods = [fx.pod(requests={"cpu": "1"}, tolerations=tol if i % 3 == 0 else None) for i in range(30)]
    got, _ = check(oracle, emu, fx.problem(its, pools, pods))
    assert {c["nodePool"] for c in got["newNodeClaims"]} == {"tainted", "plain"}
    pools = [fx.node_pool("low", weight=1), fx.node_pool("high", weight=10, limits={"cpu": "20"})]
    pods = [fx.pod(requests={"cpu": "3"}, node_selector=AMD) for _ in range(12)]
    check(oracle, emu, fx.problem(its, pools, pods))
    pools = [fx.node_pool("only", limits={"cpu": "8", "memory": "1Ti"})]
    got, _ = check(oracle, emu, fx.problem(its, pools, pods))
    assert any(e["code"] == 7 for e in got["podErrors"].values())  # nodepool limits
    check(oracle, emu, fx.problem(its, [fx.node_pool("n", limits={"nodes": "0"})], pods[:2]))


def _cluster(its, rng, n_nodes, pools=("default",)):
    nodes = []
    zones = sorted({z for t in its for r in t["requirements"] if r["key"] == fx.ZONE for z in r["values"]})
    for i in range(n_nodes):
        t = rng.choice(its)
        tz = [z for r in t["requirements"] if r["key"] == fx.ZONE for z in r["values"]] or zones
        used = {"cpu": f"{rng.choice([0, 100, 500, 1500])}m", "pods": str(rng.choice([0, 1, 3]))}
        taints = [{"key": "team", "value": "0", "effect": "NoSchedule"}] if rng.random() < 0.2 else None
        nodes.append(fx.state_node(f"node-{rng.randrange(10**6):06d}-{i}", t, rng.choice(tz), rng.choice(["spot", "on-demand"]), rng.choice(pools), used=used,
                                   taints=taints, initialized=rng.random() < 0.8, under_consolidate_after=rng.random() < 0.2))
    return nodes


def test_existing_nodes_first_fit(oracle, emu):
    # suite_test.go "Existing Nodes" :2606-2727 / "In-Flight Nodes" :1829-2016: existing capacity is used before new claims,
    # initialized nodes first then by name (scheduler.go:845-858), strict requirement compatibility (existingnode.go:100)
    its = fx.fake_default_instance_types()
    by = {t["name"]: t for t in its}
    nodes = [fx.state_node("node-b", by["default-instance-type"], "test-zone-1", used={"cpu": "1"}),
             fx.state_node("node-a", by["small-instance-type"], "test-zone-2"),
             fx.state_node("node-c", by["default-instance-type"], "test-zone-3", initialized=False),
             fx.state_node("node-t", by["arm-instance-type"], "test-zone-1", taints=[{"key": "k", "value": "v", "effect": "NoSchedule"}])]
    pods = ([fx.pod(requests={"cpu": "1"}) for _ in range(8)] + [fx.pod(requests={"cpu": "500m"}, node_selector={fx.ZONE: "test-zone-3"}) for _ in range(3)]
            + [fx.pod(requests={"cpu": "3"}, tolerations=[{"key": "k", "operator": "Exists"}]) for _ in range(4)] + [fx.pod(requests={"cpu": "1"}, node_selector={"custom": "x"})]
            + [fx.pod(node_selector={fx.HOSTNAME: "node-a"}), fx.pod(node_requirements=[fx.req(fx.HOSTNAME, "NotIn", "node-a", "node-b")])])
    got, want = check(oracle, emu, fx.problem(its, [fx.node_pool(limits={"cpu": "100"})], pods, state_nodes=nodes))
    assert [e["name"] for e in got["existingNodes"]] == ["node-a", "node-b", "node-t", "node-c"]
    assert sum(len(e["pods"]) for e in got["existingNodes"]) > 0 and len(got["newNodeClaims"]) >= 1


def test_existing_nodes_consolidation_simulation_fuzz(oracle, emu):
    # disruption.SimulateScheduling (helpers.go:53-155) shapes: non-pending pods from candidate nodes, nodes under
    # consolidateAfter are skipped unless the pod is pending or comes from a deleting node (scheduler.go:628)
    rng = random.Random(77)
    for trial in range(12):
        its = fx.kwok_catalog(rng.choice([12, 40, 144]))
        nodes = _cluster(its, rng, rng.choice([3, 40, 150]))
        pods = []
        for j in range(rng.choice([10, 120])):
            sel = {fx.ZONE: rng.choice(fx.KWOK_ZONES)} if rng.random() < 0.3 else {}
            pods.append(fx.pod(requests={"cpu": f"{rng.choice([100, 500, 2000])}m", "memory": f"{rng.choice([256, 2048])}Mi"}, node_selector=sel,
                               tolerations=[{"key": "team", "operator": "Exists"}] if rng.random() < 0.4 else None,
                               phase=rng.choice(["Pending", "Running"]), node_name=rng.choice(["", "gone-node"])))
        np_ = fx.node_pool("default", limits={"cpu": str(rng.choice([8, 200, 100000]))})
        np_["nodeClassLabelKey"] = "karpenter.kwok.sh/kwoknodeclass"
        prob = fx.problem(its, [np_], pods, well_known=fx.KWOK_WELL_KNOWN, state_nodes=nodes, options={"consolidationSimulation": True},
                          deleting_node_names=["gone-node"] if rng.random() < 0.5 else [])
        check(oracle, emu, prob)


def test_empty_and_degenerate_inputs(oracle, emu):
    its = fx.fake_default_instance_types()
    check(oracle, emu, fx.problem(its, [fx.node_pool()], []))
    check(oracle, emu, fx.problem(its, [fx.node_pool(requirements=[fx.req(fx.ARCH, "In", "sparc")])], [fx.pod()]))   # template filtered out: no templates
    check(oracle, emu, fx.problem(its, [fx.node_pool()], [fx.pod(uid="not-a-uuid-b"), fx.pod(uid="not-a-uuid-a"), fx.pod(uid="Z")]))  # rank fallback for uid order


def test_unsupported_is_loud_not_cpu(emu):
    with pytest.raises(Unsupported):   # more requirement keys than the device's key mask has bits
        pods = [fx.pod(node_requirements=[fx.req(f"example.com/key-{i}", "Exists") for i in range(40)])]
        NewScheduler(fx.problem(fx.fake_default_instance_types(), [fx.node_pool()], pods), solver_lib=emu)
    with pytest.raises(Unsupported):   # override groups are solved on the device, except on reserved offerings
        its = fx.fake_default_instance_types()
        its[0]["offerings"].append(dict(fx.offering("reserved", "test-zone-1", 0.01, reservation_id="r-1", reservation_capacity=1), capacityOverride={"memory": "1Gi"}))
        NewScheduler(fx.problem(its, [fx.node_pool()], [fx.pod()], options={"reservedCapacity": True}), solver_lib=emu)


def _with_override_offerings(it, available=True, capacity=None, overhead=None, zones=None):
    """suite_test.go:5532-5545: base offerings cloned with CapacityOverride / OverheadOverride."""
    clones = []
    for o in list(it["offerings"]):
        if zones is not None and not any(q["key"] == fx.ZONE and q["values"][0] in zones for q in o["requirements"]):
            continue
        c = dict(o, available=available)
        if capacity is not None:
            c["capacityOverride"] = dict(capacity)
        if overhead is not None:
            c["overheadOverride"] = dict(overhead)
        clones.append(c)
    it["offerings"] = it["offerings"] + clones
    return it


def test_offering_override_groups(oracle, emu):
    """Offering CapacityOverride / OverheadOverride (types.go:202-269): every distinct override pair is one more allocatable
    group of the type; fits() passes the type when SOME group holds the requests and has a compatible offering
    (nodeclaim.go:624-638). The reference's two scenarios (suite_test.go:5524-5607) and the group semantics, on the device."""
    ext = "test.com/extended-slots"
    res = {"cpu": "4", "memory": "8Gi"}
    mk = lambda name="t", **kw: _with_override_offerings(fx.fake_instance_type(name, dict(res)), **kw)
    # suite_test.go:5524-5566: only the type whose override offerings carry the extended resource is selected
    got, _ = check(oracle, emu, fx.problem([mk("override-capable", capacity={ext: "4"}, overhead={"memory": "1Gi"}), fx.fake_instance_type("normal", dict(res))],
                                           [fx.node_pool()], [fx.pod(requests={ext: "1"})]))
    assert not got["podErrors"] and got["newNodeClaims"][0]["instanceTypes"] == ["override-capable"]
    # suite_test.go:5568-5607: the override allocatable would fit, but its offerings are unavailable
    got, _ = check(oracle, emu, fx.problem([mk("override-capable", available=False, capacity={ext: "4"}, overhead={"memory": "1Gi"})], [fx.node_pool()], [fx.pod(requests={ext: "1"})]))
    assert len(got["podErrors"]) == 1 and not got["newNodeClaims"]
    # one group trades memory for slots: the two pods fit different groups of the same type and cannot share a claim
    ov = mk(capacity={ext: "4"}, overhead={"memory": "1Gi"})
    got, _ = check(oracle, emu, fx.problem([ov], [fx.node_pool()], [fx.pod(requests={"memory": "7680Mi"}), fx.pod(requests={ext: "1"})]))
    assert sorted(len(c["pods"]) for c in got["newNodeClaims"]) == [1, 1]
    got, _ = check(oracle, emu, fx.problem([ov], [fx.node_pool()], [fx.pod(requests={ext: "1", "memory": "1Gi"}) for _ in range(4)] + [fx.pod(requests={ext: "1"})]))
    assert sorted(len(c["pods"]) for c in got["newNodeClaims"]) == [1, 4]
    # the fitting group has no compatible offering (overrides only in zone 3, pod pinned to zone 1): error flags included
    z3 = mk(capacity={ext: "2"}, zones=["test-zone-3"])
    got, _ = check(oracle, emu, fx.problem([z3], [fx.node_pool()], [fx.pod(requests={ext: "1"}, node_selector={fx.ZONE: "test-zone-1"})]))
    assert len(got["podErrors"]) == 1
    got, _ = check(oracle, emu, fx.problem([z3], [fx.node_pool()], [fx.pod(requests={ext: "1"}), fx.pod(requests={ext: "1"}, node_selector={fx.ZONE: "test-zone-3"})]))
    assert not got["podErrors"]
    # a capacity override replaces the whole key; with the base offerings unavailable only the shrunken group can launch
    t = mk(capacity={"memory": "2Gi"})
    for o in t["offerings"][:5]:
        o["available"] = False
    got, _ = check(oracle, emu, fx.problem([t], [fx.node_pool()], [fx.pod(requests={"memory": "3Gi"}), fx.pod(requests={"memory": "1Gi"})]))
    assert len(got["podErrors"]) == 1 and len(got["newNodeClaims"]) == 1
    t = mk(capacity={})                                  # an empty override map without an overhead override is the base group
    for o in t["offerings"][:5]:
        o["available"] = False
    got, _ = check(oracle, emu, fx.problem([t], [fx.node_pool()], [fx.pod(requests={"memory": "3Gi"})]))
    assert not got["podErrors"]
    # an overhead override that drives a dimension negative: resources.Fits refuses that group (resources.go:190)
    t = mk(overhead={"memory": "9Gi"}, zones=["test-zone-1"])
    check(oracle, emu, fx.problem([t], [fx.node_pool()], [fx.pod(node_selector={fx.ZONE: "test-zone-1"}), fx.pod()]))


def test_host_ports(oracle, emu):
    """HostPortUsage (hostportusage.go:39-117) in NodeClaim.CanAdd (per daemon-overhead group, nodeclaim.go:256-259 and
    :562-565) and ExistingNode.CanAdd (existingnode.go:87-93, :178). The reference's three known answers and the matching
    rule (protocol, port, equal or unspecified IP)."""
    its = fx.fake_default_instance_types()
    # scheduling/suite_test.go:2593-2603: the same host port twice -> two nodes
    got, _ = check(oracle, emu, fx.problem(its, [fx.node_pool()], [fx.pod(host_ports=[8080]), fx.pod(host_ports=[8080])]))
    assert len(got["newNodeClaims"]) == 2 and not got["podErrors"]
    # provisioning/suite_test.go:956-973: a compatible daemonset holds the port on every node that could be launched
    ds = fx.pod(requests={"cpu": "2", "memory": "2Gi"}, host_ports=[8080])
    got, _ = check(oracle, emu, fx.problem(its, [fx.node_pool()], [fx.pod(requests={"cpu": "1", "memory": "1Gi"}, host_ports=[8080])], daemonset_pods=[ds]))
    assert len(got["podErrors"]) == 1 and not got["newNodeClaims"]
    # provisioning/suite_test.go:1496-1530: the daemonset with the port only runs on large types, the pod asks for small ones
    size = lambda v: [fx.req(fx.FAKE_LABEL_INSTANCE_SIZE, "In", v)]
    dss = [fx.pod(requests={"cpu": "4", "memory": "4Gi"}, node_requirements=size("large"), host_ports=[8080]),
           fx.pod(requests={"cpu": "2", "memory": "2Gi"}, node_requirements=size("small"), host_ports=[8081])]
    got, _ = check(oracle, emu, fx.problem(its, [fx.node_pool()], [fx.pod(requests={"cpu": "1", "memory": "1Gi"}, node_requirements=size("small"), host_ports=[8080])], daemonset_pods=dss))
    assert not got["podErrors"] and len(got["newNodeClaims"]) == 1
    # without the size requirement the pod's claim keeps only the group whose daemons leave the port free
    got, _ = check(oracle, emu, fx.problem(its, [fx.node_pool()], [fx.pod(requests={"cpu": "1", "memory": "1Gi"}, host_ports=[8080]),
                                                                   fx.pod(requests={"cpu": "1"}, host_ports=[8081])], daemonset_pods=dss))
    # HostPort.Matches: protocol and port must agree; IPs must be equal unless one side is unspecified (0.0.0.0, "" or ::)
    hp = lambda port, ip="", proto="TCP": {"port": port, "ip": ip, "protocol": proto}
    pods = [fx.pod(host_ports=[hp(80, "10.0.0.1")]), fx.pod(host_ports=[hp(80, "10.0.0.2")]), fx.pod(host_ports=[hp(80, "10.0.0.1", "UDP")]),
            fx.pod(host_ports=[hp(80)]), fx.pod(host_ports=[hp(80, "::")]), fx.pod(host_ports=[hp(81, "10.0.0.1"), hp(82)]), fx.pod(host_ports=[hp(82, "10.0.0.9")]),
            fx.pod(host_ports=[hp(0)]), fx.pod()]
    got, _ = check(oracle, emu, fx.problem(its, [fx.node_pool()], pods))
    assert not got["podErrors"]
    # existing nodes: ports of bound pods (StateNode.HostPortUsage) and of pods added during this Solve
    node = fx.state_node("node-a", its[2], "test-zone-1", host_ports=[8080])
    free = fx.state_node("node-b", its[2], "test-zone-2")
    pods = [fx.pod(host_ports=[8080]), fx.pod(host_ports=[8080]), fx.pod(host_ports=[8080]), fx.pod(host_ports=[9090]), fx.pod(host_ports=[9090])]
    got, _ = check(oracle, emu, fx.problem(its, [fx.node_pool()], pods, state_nodes=[node, free]))
    on_nodes = {e["name"]: len(e["pods"]) for e in got["existingNodes"]}
    assert on_nodes == {"node-a": 1, "node-b": 2} and len(got["newNodeClaims"]) == 2
    # two passes: what the first pass launched keeps its ports in the second
    first = fx.problem(its, [fx.node_pool()], [fx.pod(host_ports=[8080]), fx.pod(host_ports=[8443])])
    got, _ = check(oracle, emu, first)
    nodes, bound = fx.launch(got, its, first["pods"])
    assert sum(len(n.get("hostPorts", [])) for n in nodes) == 2
    got, _ = check(oracle, emu, fx.problem(its, [fx.node_pool()], [fx.pod(host_ports=[8080, 8443]), fx.pod(host_ports=[8081])], state_nodes=nodes, cluster_pods=bound))
    assert len(got["newNodeClaims"]) == 1


def test_host_ports_fuzz(oracle, emu):
    for seed in range(8):
        rng = random.Random(700 + seed)
        its = fx.fake_instance_types(rng.choice([6, 20]))
        ports = [8080, 8081, 9000, 9001, 53]
        pick = lambda: [{"port": rng.choice(ports), "ip": rng.choice(["", "", "10.0.0.1", "10.0.0.2"]), "protocol": rng.choice(["TCP", "TCP", "UDP"])}
                        for _ in range(rng.choice([0, 0, 1, 1, 2]))]
        pods = [fx.pod(requests={"cpu": rng.choice(["100m", "500m", "1"])}, host_ports=pick(),
                       node_selector=({fx.ZONE: rng.choice(["test-zone-1", "test-zone-2"])} if rng.random() < 0.3 else None)) for _ in range(rng.choice([15, 40]))]
        kw = {}
        if seed % 2 == 0:
            kw["daemonset_pods"] = [fx.pod(requests={"cpu": "100m"}, host_ports=[{"port": 9000, "ip": "", "protocol": "TCP"}],
                                           node_requirements=[fx.req(fx.ZONE, "In", "test-zone-1")]), fx.pod(requests={"memory": "64Mi"}, host_ports=pick())]
        if seed % 3 == 0:
            kw["state_nodes"] = [fx.state_node(f"n-{i}", its[-1], rng.choice(["test-zone-1", "test-zone-2"]), host_ports=pick()) for i in range(3)]
        check(oracle, emu, fx.problem(its, [fx.node_pool()], pods, **kw))


def test_offering_override_groups_fuzz(oracle, emu):
    """Random catalogues where some types carry one or two override groups in some zones, pods with zone selectors and
    extended-resource requests, daemonset overhead, existing claims that keep growing: claim by claim against the oracle."""
    ext = "test.com/extended-slots"
    for seed in range(8):
        rng = random.Random(900 + seed)
        its = fx.fake_instance_types(rng.choice([6, 12, 70]))
        for it in its:
            if rng.random() < 0.5:
                _with_override_offerings(it, capacity={ext: str(rng.choice([2, 4, 8]))}, overhead=rng.choice([None, {"memory": "1Gi"}, {"cpu": "1"}]),
                                         zones=rng.choice([None, ["test-zone-1"], ["test-zone-2", "test-zone-3"]]))
            if rng.random() < 0.25:
                _with_override_offerings(it, capacity={"memory": rng.choice(["2Gi", "64Gi"])}, zones=[rng.choice(["test-zone-1", "test-zone-2", "test-zone-3"])],
                                         available=rng.random() < 0.8)
        pods = []
        for _ in range(rng.choice([20, 60])):
            req = {"cpu": rng.choice(["100m", "500m", "1", "2"]), "memory": rng.choice(["128Mi", "1Gi", "3Gi"])}
            if rng.random() < 0.5:
                req[ext] = str(rng.choice([1, 2, 3]))
            sel = {fx.ZONE: rng.choice(["test-zone-1", "test-zone-2", "test-zone-3"])} if rng.random() < 0.4 else None
            pods.append(fx.pod(requests=req, node_selector=sel))
        kw = {}
        if seed % 3 == 0:
            kw["daemonset_pods"] = [fx.pod(requests={"cpu": "100m", "memory": "64Mi"})]
        check(oracle, emu, fx.problem(its, [fx.node_pool()], pods, **kw))


def test_hugepages(oracle, emu):
    """Hugepage capacity is carved out of the allocatable memory (computeAllocatable, types.go:281-291) and is a resource
    dimension of its own for pods that request it."""
    its = fx.fake_instance_types(6)
    for i, it in enumerate(its):
        it["capacity"]["hugepages-2Mi"] = f"{512 * (i + 1)}Mi"
    its[0]["capacity"]["hugepages-1Gi"] = "4Gi"      # more than the type's memory: clamps at zero, nothing fits there
    pods = [fx.pod(requests={"cpu": "500m", "memory": "1Gi"}) for _ in range(9)]
    pods += [fx.pod(requests={"cpu": "250m", "memory": "256Mi", "hugepages-2Mi": "1Gi"}) for _ in range(5)]
    pods += [fx.pod(requests={"memory": "3Gi", "hugepages-2Mi": "2560Mi"}) for _ in range(2)]
    pods += [fx.pod(requests={"hugepages-1Gi": "8Gi"})]                                   # no type has that much
    got, want = check(oracle, emu, fx.problem(its, [fx.node_pool()], pods))
    assert len(got["podErrors"]) == 1 and got["newNodeClaims"]
    # without the carve-out a 1Gi pod would fit the smallest type: it must not be among any claim's options
    assert all("fake-it-0" not in c["instanceTypes"] for c in got["newNodeClaims"])


def sorted_its(res):
    """Daemon-overhead groups are visited in Go map order by the reference (scheduler.go:1001), so the order of
    InstanceTypeOptions across groups is not defined: compare them as sets."""
    for c in res["newNodeClaims"]:
        c["instanceTypes"] = sorted(c["instanceTypes"])
    return res


def check_daemons(oracle, emu, prob):
    want = sorted_its(oracle.solve(prob))
    got = sorted_its(NewScheduler(prob, solver_lib=emu).Solve())
    parity.assert_same_results(got, want)
    assert got["counters"]["referenceBinEvaluations"] == want["counters"]["binEvaluations"]
    return got


def test_daemonset_overhead(oracle, emu):
    """suite_test.go "Daemonsets" (:2143-2460): overhead is added per instance type when fitting, and the smallest overhead
    is added to the NodeClaim's requests by FinalizeScheduling (nodeclaim.go:353-377)."""
    its = fx.fake_default_instance_types()
    ds = [fx.pod(requests={"cpu": "1", "memory": "1Gi"})]
    got = check_daemons(oracle, emu, fx.problem(its, [fx.node_pool()], [fx.pod(requests={"cpu": "1", "memory": "1Gi"})], daemonset_pods=ds))
    req = got["newNodeClaims"][0]["requests"]
    assert int(req["cpu"]) == 2 * 10**9 and int(req["pods"]) == 2 * 10**9                 # suite_test.go:2155-2172
    # a daemonset restricted to arm64 splits a template's instance types into two overhead groups
    ds2 = ds + [fx.pod(requests={"cpu": "2"}, node_selector={fx.ARCH: "arm64"})]
    pods = [fx.pod(requests={"cpu": f"{c}m"}) for c in (500, 900, 1500, 2500, 3500) for _ in range(4)]
    check_daemons(oracle, emu, fx.problem(its, [fx.node_pool()], pods, daemonset_pods=ds2))
    check_daemons(oracle, emu, fx.problem(its, [fx.node_pool()], pods + [fx.pod(node_selector={fx.ARCH: "arm64"}, requests={"cpu": "3"})], daemonset_pods=ds2))
    # daemonset that does not tolerate the NodePool's taint is not counted; one with a node affinity term that has to relax is
    pools = [fx.node_pool(taints=[{"key": "k", "value": "v", "effect": "NoSchedule"}])]
    tol = [{"key": "k", "operator": "Exists"}]
    ds3 = [fx.pod(requests={"cpu": "1"}), fx.pod(requests={"cpu": "500m"}, tolerations=tol,
                                                   node_requirements=[[fx.req(fx.ZONE, "In", "nowhere")], [fx.req(fx.ZONE, "In", "test-zone-2")]])]
    check_daemons(oracle, emu, fx.problem(its, pools, [fx.pod(requests={"cpu": "1"}, tolerations=tol) for _ in range(6)], daemonset_pods=ds3))
    # existing nodes: expected daemons minus what already runs there (existingnode.go:50-64)
    nodes = [fx.state_node(f"node-{i}", its[2], "test-zone-1", "on-demand", "default", used={"cpu": "500m", "pods": "1"}) for i in range(3)]
    nodes[0]["daemonSetRequests"] = {"cpu": "1", "memory": "1Gi", "pods": "1"}
    check_daemons(oracle, emu, fx.problem(its, [fx.node_pool()], [fx.pod(requests={"cpu": "900m"}) for _ in range(8)], daemonset_pods=ds, state_nodes=nodes))


def test_daemonset_fuzz(oracle, emu):
    rng = random.Random(77)
    for trial in range(20):
        its = fx.kwok_catalog(rng.choice([20, 70, 144])) if trial % 2 else fx.fake_instance_types(rng.choice([6, 20]))
        wk = fx.KWOK_WELL_KNOWN if trial % 2 else fx.FAKE_WELL_KNOWN
        zones = fx.KWOK_ZONES if trial % 2 else ["test-zone-1", "test-zone-2", "test-zone-3"]
        ds = []
        for _ in range(rng.randrange(1, 4)):
            kw = dict(requests={"cpu": f"{rng.choice([50, 100, 250])}m", "memory": f"{rng.choice([64, 128])}Mi"})
            pick = rng.random()
            if pick < 0.3:
                kw["node_selector"] = {fx.ARCH: rng.choice(["amd64", "arm64"])}
            elif pick < 0.5:
                kw["node_selector"] = {fx.ZONE: rng.choice(zones)}
            elif pick < 0.6:
                kw["node_requirements"] = [[fx.req(fx.OS, "In", "plan9")], [fx.req(fx.OS, "In", "linux")]]
            ds.append(fx.pod(**kw))
        pods = [fx.pod(requests={"cpu": f"{rng.choice(fx.BENCH_CPU_M)}m", "memory": f"{rng.choice(fx.BENCH_MEM_MI)}Mi"},
                       node_selector=rng.choice([None, None, {fx.ARCH: "arm64"}, {fx.ZONE: rng.choice(zones)}])) for _ in range(rng.randrange(20, 150))]
        np_ = fx.node_pool()
        if trial % 2:
            np_["nodeClassLabelKey"] = "karpenter.kwok.sh/kwoknodeclass"
        check_daemons(oracle, emu, fx.problem(its, [np_], pods, daemonset_pods=ds, well_known=wk))


def _mv_types():
    off = [fx.offering("spot", "test-zone-1-spot", 0.52)]
    small = fx.fake_instance_type("instance-type-1", resources={"cpu": "1", "memory": "1Gi"}, offerings=off, architecture="arm64", operating_systems=["linux"])
    big = fx.fake_instance_type("instance-type-2", resources={"cpu": "4", "memory": "4Gi"}, offerings=[fx.offering("spot", "test-zone-1-spot", 1.0)], architecture="arm64", operating_systems=["linux"])
    return [small, big]


def test_min_values(oracle, emu):
    """instance_selection_test.go "MinValues" (:620-1500): a claim must keep at least minValues distinct values of the key
    among its instance types, so two pods that only fit together on the big type go to two nodes."""
    its = _mv_types()
    two = [fx.pod(requests={"cpu": "0.9", "memory": "0.9Gi"}) for _ in range(2)]
    pool = fx.node_pool(requirements=[fx.req(fx.INSTANCE_TYPE, "In", "instance-type-1", "instance-type-2", min_values=2)])
    got, _ = check(oracle, emu, fx.problem(its, [pool], two))
    assert len(got["newNodeClaims"]) == 2 and all(len(c["instanceTypes"]) == 2 for c in got["newNodeClaims"])      # :621-691
    # without minValues both share the big type
    got, _ = check(oracle, emu, fx.problem(its, [fx.node_pool(requirements=[fx.req(fx.INSTANCE_TYPE, "In", "instance-type-1", "instance-type-2")])], two))
    assert len(got["newNodeClaims"]) == 1
    # more values required than exist: strict policy fails the pods (:1234-1259), BestEffort relaxes and annotates
    pool3 = fx.node_pool(requirements=[fx.req(fx.INSTANCE_TYPE, "In", "instance-type-1", "instance-type-2", min_values=3)])
    got, _ = check(oracle, emu, fx.problem(its, [pool3], two))
    assert len(got["podErrors"]) == 2 and not got["newNodeClaims"]
    got, _ = check(oracle, emu, fx.problem(its, [pool3], two, options={"minValuesPolicy": "BestEffort"}))
    assert not got["podErrors"]
    assert all(c["annotations"]["karpenter.sh/nodeclaim-min-values-relaxed"] == "true" for c in got["newNodeClaims"])
    # a pod that only fits the big type cannot keep two values: strict = error code MIN_VALUES on the new claim
    big_pod = [fx.pod(requests={"cpu": "3"})]
    got, _ = check(oracle, emu, fx.problem(its, [pool], big_pod))
    assert list(got["podErrors"].values())[0]["code"] == 10
    got, _ = check(oracle, emu, fx.problem(its, [pool], big_pod, options={"minValuesPolicy": "BestEffort"}))
    assert not got["podErrors"]
    # minValues on a label other than the instance type, several keys, and Exists with minValues
    kw = fx.kwok_catalog(144)
    np_ = fx.node_pool(requirements=[fx.req("karpenter.kwok.sh/instance-family", "Exists", min_values=3), fx.req(fx.INSTANCE_TYPE, "Exists", min_values=10)])
    np_["nodeClassLabelKey"] = "karpenter.kwok.sh/kwoknodeclass"
    pods = [fx.pod(requests={"cpu": f"{c}m", "memory": f"{m}Mi"}) for c in (500, 4000, 30000, 120000) for m in (512, 8192, 65536) for _ in range(3)]
    for policy in ("Strict", "BestEffort"):
        check(oracle, emu, fx.problem(kw, [np_], pods, well_known=fx.KWOK_WELL_KNOWN, options={"minValuesPolicy": policy}))


def reserved_types(capacity=1):
    """suite_test.go "Reserved Instance Types" BeforeEach (:4677-4714): medium and small carry a reserved offering each."""
    its = [fx.fake_instance_type(n, resources={"cpu": str(c), "memory": f"{c}Gi"}) for n, c in (("large-instance-type", 6), ("medium-instance-type", 3), ("small-instance-type", 2))]
    for it in its[1:]:
        for r in it["requirements"]:
            if r["key"] == fx.CAPACITY_TYPE:
                r["values"].append("reserved")
        it["offerings"].append(fx.offering("reserved", "test-zone-1", fx.fake_price(it["capacity"]) / 100000.0, reservation_id="r-" + it["name"], reservation_capacity=capacity))
    return its


def test_reserved_offerings(oracle, emu):
    strict = {"reservedCapacity": True, "reservedOfferingMode": "Strict"}
    fallback = {"reservedCapacity": True, "reservedOfferingMode": "Fallback"}
    pods = lambda n, cpu="1800m": [fx.pod(requests={"cpu": cpu}) for _ in range(n)]
    # suite_test.go:4715-4765: the first claim reserves both offerings, so only one pod schedules per loop
    got, _ = check(oracle, emu, fx.problem(reserved_types(), [fx.node_pool()], pods(3), options=strict))
    assert len(got["newNodeClaims"]) == 1 and len(got["newNodeClaims"][0]["pods"]) == 1
    assert sorted(e["code"] for e in got["podErrors"].values()) == [8, 8]
    claim = got["newNodeClaims"][0]
    assert sorted(claim["reservedOfferings"]) == ["r-medium-instance-type", "r-small-instance-type"]
    assert [q["values"] for q in claim["requirements"] if q["key"] == fx.CAPACITY_TYPE] == [["reserved"]]
    # Fallback mode (the consolidation simulator's, helpers.go:106-112) may use on-demand / spot instead
    got, _ = check(oracle, emu, fx.problem(reserved_types(), [fx.node_pool()], pods(3), options=fallback))
    assert not got["podErrors"]
    # feature gate off: reserved offerings are ordinary offerings
    check(oracle, emu, fx.problem(reserved_types(), [fx.node_pool()], pods(3)))
    # more capacity, more pods, several NodePools sharing the reservations (suite_test.go:4767-4822)
    for opts in (strict, fallback):
        check(oracle, emu, fx.problem(reserved_types(3), [fx.node_pool("np-1", weight=2), fx.node_pool("np-2")], pods(7, "900m") + pods(4, "2500m"), options=opts))
        sel = [fx.pod(requests={"cpu": "1"}, node_selector={fx.CAPACITY_TYPE: ct}) for ct in ("reserved", "reserved", "on-demand", "spot", "reserved")]
        check(oracle, emu, fx.problem(reserved_types(2), [fx.node_pool()], sel + pods(5, "600m"), options=opts))
        check(oracle, emu, fx.problem(reserved_types(2), [fx.node_pool()], [fx.pod(requests={"cpu": "500m"}, node_selector={fx.ZONE: z}) for z in ("test-zone-1", "test-zone-2", "test-zone-1", "test-zone-3")], options=opts))


def test_reserved_offerings_fuzz(oracle, emu):
    rng = random.Random(4242)
    for trial in range(25):
        its = fx.fake_instance_types(rng.choice([4, 8, 12]))
        n_res = rng.randrange(1, 5)
        for i in range(n_res):
            for it in rng.sample(its, rng.randrange(1, 3)):
                for r in it["requirements"]:
                    if r["key"] == fx.CAPACITY_TYPE and "reserved" not in r["values"]:
                        r["values"].append("reserved")
                it["offerings"].append(fx.offering("reserved", rng.choice(["test-zone-1", "test-zone-2"]), 0.001 * (i + 1), reservation_id=f"cr-{i}",
                                                   reservation_capacity=rng.randrange(1, 4), available=rng.random() < 0.9))
        pods = []
        for _ in range(rng.randrange(5, 40)):
            kw = dict(requests={"cpu": f"{rng.choice([250, 500, 1000, 2000])}m", "memory": f"{rng.choice([128, 512, 1024])}Mi"})
            pick = rng.random()
            if pick < 0.2:
                kw["node_selector"] = {fx.CAPACITY_TYPE: rng.choice(["reserved", "on-demand", "spot"])}
            elif pick < 0.35:
                kw["node_selector"] = {fx.ZONE: rng.choice(["test-zone-1", "test-zone-2", "test-zone-3"])}
            elif pick < 0.45:
                kw["node_preferences"] = [fx.req(fx.CAPACITY_TYPE, "In", "reserved")]
            pods.append(fx.pod(**kw))
        opts = {"reservedCapacity": True, "reservedOfferingMode": rng.choice(["Strict", "Fallback"])}
        pools = [fx.node_pool()] if trial % 3 else [fx.node_pool("a", weight=5, limits={"cpu": "20"}), fx.node_pool("b")]
        check(oracle, emu, fx.problem(its, pools, pods, options=opts))


def test_random_problems_fuzz(oracle, emu):
    rng = random.Random(2024)
    for trial in range(25):
        n_types = rng.choice([6, 20, 70, 144])
        its = fx.kwok_catalog(n_types) if rng.random() < 0.6 else fx.fake_instance_types(n_types)
        wk = fx.KWOK_WELL_KNOWN if "kwok" in str(its[0]["requirements"]) else fx.FAKE_WELL_KNOWN
        zones = sorted({z for t in its for r in t["requirements"] if r["key"] == fx.ZONE for z in r["values"]})
        pools = []
        for i in range(rng.choice([1, 1, 2, 3])):
            taints = [{"key": "team", "value": str(i), "effect": "NoSchedule"}] if rng.random() < 0.4 else None
            reqs = [fx.req(fx.ZONE, rng.choice(["In", "NotIn"]), rng.choice(zones))] if rng.random() < 0.3 else None
            limits = {"cpu": str(rng.choice([4, 64, 1000]))} if rng.random() < 0.3 else None
            pools.append(fx.node_pool(f"np-{i}", weight=rng.choice([0, 5, 5, 9]), taints=taints, requirements=reqs, limits=limits, labels={"team": str(i)} if rng.random() < 0.5 else None))
        pods = []
        for j in range(rng.choice([5, 40, 300])):
            sel = {}
            if rng.random() < 0.3: sel[fx.ZONE] = rng.choice(zones + ["nowhere"])
            if rng.random() < 0.2: sel[fx.ARCH] = rng.choice(["amd64", "arm64"])
            if rng.random() < 0.1: sel["team"] = rng.choice(["0", "1", "9"])
            tol = [{"key": "team", "operator": "Exists"}] if rng.random() < 0.5 else None
            nreq = [fx.req(fx.CAPACITY_TYPE, rng.choice(["In", "NotIn"]), rng.choice(["spot", "on-demand"]))] if rng.random() < 0.2 else None
            pods.append(fx.pod(requests={"cpu": f"{rng.choice([100, 250, 1000, 3500])}m", "memory": f"{rng.choice([128, 512, 4096])}Mi"}, node_selector=sel, tolerations=tol,
                               node_requirements=nreq, creation=rng.choice([0, 0, 5])))
        check(oracle, emu, fx.problem(its, pools, pods, well_known=wk))


def test_claim_order_emulation_matches_go_pdqsort(oracle, emu):
    """pdq_emul.h against the oracle's literal pdqsort port on random commit traces (ties everywhere, n up to 4000)."""
    lib = ctypes.CDLL(emu)
    rng = random.Random(11)
    for trial in range(40):
        ops, n = [], 0
        for _ in range(rng.choice([15, 49, 51, 130, 700, 4000])):
            if n == 0 or rng.random() < rng.choice([0.02, 0.2, 0.7]):
                ops.append(-1); n += 1
            else:
                ops.append(rng.randrange(n) if rng.random() < 0.5 else max(0, n - 1 - int(rng.expovariate(0.2))))
        arr = (ctypes.c_int * len(ops))(*ops)
        out = (ctypes.c_int * (len(ops) + 1))()
        got_n = lib.ksolve_emu_order_trace(arr, len(ops), out, None)
        assert list(out[:got_n]) == oracle.evaluate({"fn": "order_trace", "ops": ops}), trial


def test_more_than_4096_claims(oracle, emu):
    """The staged first-fit scan keeps two dead-row words per lane: up to 8192 in-flight NodeClaims per problem."""
    its = fx.fake_default_instance_types()
    pods = [fx.pod(requests={"cpu": "9"}) for _ in range(4300)] + [fx.pod(requests={"cpu": "200m"}) for _ in range(900)]
    got, _ = check(oracle, emu, fx.problem(its, [fx.node_pool()], pods))
    assert len(got["newNodeClaims"]) > 4096
    lab = {"app": "nginx"}
    pods = [fx.pod(labels=lab, requests={"cpu": "100m"}, pod_anti_requirements=[fx.affinity_term(fx.HOSTNAME, lab)]) for _ in range(4200)]
    got, _ = check(oracle, emu, fx.problem(its, [fx.node_pool()], pods + [fx.pod(requests={"cpu": "1"}) for _ in range(300)]))
    assert len(got["newNodeClaims"]) > 4096


def test_more_claims_than_the_lds_order_holds(oracle, emu):
    """When a solve needs more in-flight NodeClaims than the LDS-resident claim order holds (8192 by default, lowered here
    with ldsClaimCap so that the case stays small) it is re-run on the BIG engine (claim order in HBM): same results."""
    its = fx.fake_default_instance_types()
    lab = {"app": "nginx"}
    opts = {"ldsClaimCap": 128}
    pods = [fx.pod(labels=lab, requests={"cpu": "100m"}, pod_anti_requirements=[fx.affinity_term(fx.HOSTNAME, lab)]) for _ in range(700)]
    pods += [fx.pod(requests={"cpu": "1"}) for _ in range(200)] + [fx.pod(labels={"x": "y"}, topology_spread=[fx.spread(fx.ZONE, {"x": "y"})]) for _ in range(30)]
    pods += [fx.pod(labels={"h": "s"}, topology_spread=[fx.spread(fx.HOSTNAME, {"h": "s"}, max_skew=2)]) for _ in range(90)]
    got, _ = check(oracle, emu, fx.problem(its, [fx.node_pool()], pods, options=opts))
    assert len(got["newNodeClaims"]) > 700
    # a problem without topology: big pods, one per node; and the C2 mix with a tiny cap
    pods = [fx.pod(requests={"cpu": "9"}) for _ in range(650)] + [fx.pod(requests={"cpu": "300m"}) for _ in range(500)]
    got, _ = check(oracle, emu, fx.problem(its, [fx.node_pool()], pods, options=opts))
    assert len(got["newNodeClaims"]) >= 650
    prob = fx.config2(pods=60000, n_types=144, seed=11)
    prob["options"]["ldsClaimCap"] = 64
    got, _ = check(oracle, emu, prob)
    assert len(got["newNodeClaims"]) > 64


def test_batch_with_an_overflowing_problem(oracle, emu):
    from karpenter_amd.scheduling import SolveBatch
    its = fx.fake_default_instance_types()
    probs = [fx.problem(its, [fx.node_pool()], [fx.pod(requests={"cpu": "9"}) for _ in range(n)] + [fx.pod(requests={"cpu": "500m"}) for _ in range(40)], options={"ldsClaimCap": 128})
             for n in (50, 400, 90)]
    for got, prob in zip(SolveBatch([NewScheduler(p, solver_lib=emu) for p in probs]), probs):
        parity.assert_same_results(got, oracle.solve(prob))


def test_batch_reports_a_failed_problem_instead_of_crashing(oracle, emu):
    """A problem that overflows maxClaims on a handle that cannot move to the BIG engine fails with 'capacity'; batched
    with healthy problems it must report that status per problem (solve_batch lost it and the caller
    dereferenced null result arrays)."""
    from karpenter_amd.scheduling import SolveBatch
    its = fx.fake_default_instance_types()
    bad = fx.problem(its, [fx.node_pool()], [fx.pod(requests={"cpu": "9"}) for _ in range(100)], options={"maxClaims": 64})
    ok = fx.problem(its, [fx.node_pool()], [fx.pod(requests={"cpu": "500m"}) for _ in range(40)], options={"maxClaims": 64})
    with pytest.raises(RuntimeError, match="capacity"):
        NewScheduler(bad, solver_lib=emu).Solve()
    with pytest.raises(RuntimeError, match="capacity"):
        SolveBatch([NewScheduler(bad, solver_lib=emu), NewScheduler(ok, solver_lib=emu)])
    with pytest.raises(RuntimeError, match="capacity"):
        SolveBatch([NewScheduler(ok, solver_lib=emu), NewScheduler(bad, solver_lib=emu)])
    got = SolveBatch([NewScheduler(ok, solver_lib=emu)])[0]
    parity.assert_same_results(got, oracle.solve(ok))


def test_size_limits_are_enforced_loudly(oracle, emu):
    """The fixed capacities of the flat format: at the limit the problem solves (and matches the oracle), past it the
    product reports Unsupported — it never degrades silently."""
    # instance types: the format allows 2048 (32 mask words); what actually bounds a problem is that the per-type tables
    # must fit one CU's LDS — 1000 KWOK types (BASELINE configs[3]) do, 2049 of anything do not
    np_ = fx.node_pool()
    np_["nodeClassLabelKey"] = "karpenter.kwok.sh/kwoknodeclass"
    pods = [fx.pod(requests={"cpu": f"{c}"}) for c in (1, 3, 17, 100, 250)]
    check(oracle, emu, fx.problem(fx.kwok_catalog(1000), [np_], pods, well_known=fx.KWOK_WELL_KNOWN))
    check(oracle, emu, fx.problem(fx.fake_instance_types(800), [fx.node_pool()], pods))
    with pytest.raises(Unsupported):
        NewScheduler(fx.problem(fx.fake_instance_types(2049), [fx.node_pool()], [fx.pod()]), solver_lib=emu)
    # requirement keys: the well-known / template keys plus custom pod keys, up to 32 in total
    base = fx.fake_default_instance_types()
    def with_keys(n):
        return [fx.pod(node_requirements=[fx.req(f"custom-{i}", "NotIn", "x") for i in range(n)])]
    check(oracle, emu, fx.problem(base, [fx.node_pool()], with_keys(8)))
    with pytest.raises(Unsupported):
        NewScheduler(fx.problem(base, [fx.node_pool()], with_keys(40)), solver_lib=emu)
    # resource dimensions: cpu, memory, pods + five extended resources = 8
    its8 = fx.fake_instance_types(4)
    for i in range(5):
        its8[0]["capacity"][f"vendor.com/res-{i}"] = "4"
    check(oracle, emu, fx.problem(its8, [fx.node_pool()], [fx.pod(requests={f"vendor.com/res-{i}": "1" for i in range(5)})]))
    its9 = fx.fake_instance_types(4)
    for i in range(6):
        its9[0]["capacity"][f"vendor.com/res-{i}"] = "4"
    with pytest.raises(Unsupported):
        NewScheduler(fx.problem(its9, [fx.node_pool()], [fx.pod()]), solver_lib=emu)
    # NodePools: 32 templates
    pools = [fx.node_pool(f"pool-{i:02d}", weight=i) for i in range(32)]
    check(oracle, emu, fx.problem(base, pools, [fx.pod() for _ in range(5)]))
    with pytest.raises(Unsupported):
        NewScheduler(fx.problem(base, pools + [fx.node_pool("one-too-many")], [fx.pod()]), solver_lib=emu)


def test_product_go_sort_matches_oracle_go_sort(oracle, emu):
    """csrc/go_sort.h (the finalize kernel's OrderByPrice) and oracle/pdqsort.hpp are two independent restatements of
    Go's sort.Slice; on inputs full of ties they must leave the same permutation (every pdqsort path: insertion sort,
    ninther, partial insertion sort, partitionEqual, breakPatterns, heapsort)."""
    lib = ctypes.CDLL(emu)
    rng = random.Random(99)
    shapes = []
    for n in (0, 1, 2, 12, 13, 49, 50, 51, 100, 257, 1000, 3000):
        shapes += [[rng.randrange(0, max(1, n // 8)) for _ in range(n)],             # many ties
                   sorted(rng.randrange(0, 50) for _ in range(n)),                    # already sorted
                   sorted((rng.randrange(0, 50) for _ in range(n)), reverse=True),    # reversed
                   [rng.randrange(0, 3) for _ in range(n)],                           # three distinct keys
                   [i % 7 for i in range(n)],                                         # periodic pattern
                   list(range(n // 2)) + list(range(n - n // 2))]                     # two sorted runs
    # the organ-pipe / killer patterns that push pdqsort into breakPatterns and heapsort
    shapes.append([min(i, 2000 - i) for i in range(2000)])
    shapes.append([(i * 7919) % 13 for i in range(4000)])
    for keys in shapes:
        n = len(keys)
        arr = (ctypes.c_longlong * max(1, n))(*keys)
        out = (ctypes.c_int * max(1, n))()
        lib.ksolve_emu_go_sort(arr, n, out)
        assert list(out[:n]) == oracle.evaluate({"fn": "sort_by_key", "keys": keys}), (n, keys[:20])


def test_cancel_from_another_thread(oracle, emu):
    """ksolve_cancel is the ctx deadline of Solve (scheduler.go:477-480): raised from another thread at any moment of the
    call it stops the pack loop at the next queue block; what was placed so far is a prefix of the full solve, every
    other pod is reported unscheduled, and the next Solve starts with a fresh context."""
    import threading
    import time
    n = 300000
    s = NewScheduler(fx.config2(pods=n), solver_lib=emu)
    t0 = time.time()
    full = s.Solve(want_results=False)
    t_full = time.time() - t0
    assert not full["timedOut"] and full["scheduledPods"] == n

    def cancelled_after(delay):
        out = {}
        th = threading.Thread(target=lambda: out.update(r=s.Solve()))
        th.start()
        time.sleep(delay)
        s.Cancel()
        th.join()
        return out["r"]

    def check_prefix(r):
        placed = sum(len(c["pods"]) for c in r["newNodeClaims"])
        assert placed == r["scheduledPods"] < n and placed + len(r["podErrors"]) == n and r["counters"]["pops"] == placed
        prob = fx.config2(pods=n)
        prob.setdefault("options", {})["maxSteps"] = placed          # the cancelled run is the full run stopped after `placed` pops
        parity.assert_same_results(r, NewScheduler(prob, solver_lib=emu).Solve())
        return placed

    # the moment the cancel lands is timing-dependent: try a few delays, early (classification / queue sort: the pack
    # loop then stops before its first pod) and in the middle of the pack loop
    landed = []
    # (every cancel that lands checks that the claim order reported is the one of the last sort the reference would have run:
    # the cursor engine places the entries before a poll boundary without the eager re-sort of the group path)
    for frac in (0.03, 0.6, 0.75, 0.5, 0.85, 0.4, 0.9, 0.55, 0.65, 0.7, 0.8, 0.45):
        r = cancelled_after(frac * t_full)
        if r["timedOut"]:
            landed.append(check_prefix(r))
        if len([x for x in landed if x > 0]) >= 4:
            break
    assert landed and max(landed) > 0, f"no cancel landed inside the pack loop: {landed}"
    again = s.Solve(want_results=False)
    assert not again["timedOut"] and again["scheduledPods"] == n


def test_malformed_problems_are_rejected_not_crashed(emu):
    """The flattener and ksolve_create validate what they are given: a problem document with fields removed, nulled or
    replaced by garbage either still solves, or is refused as invalid / unsupported — it never takes the process down
    (a segfault here would kill this test run)."""
    lab = {"app": "a"}
    its = fx.fake_default_instance_types()
    pods = [fx.pod(labels=lab, requests={"cpu": "500m", "memory": "128Mi"}, topology_spread=[fx.spread(fx.ZONE, lab)], tolerations=[{"key": "k", "operator": "Exists"}],
                   node_requirements=[[fx.req(fx.ZONE, "In", "test-zone-1")], [fx.req(fx.ARCH, "In", "amd64")]],
                   pod_anti_preferences=[fx.weighted(2, fx.affinity_term(fx.HOSTNAME, lab))]) for _ in range(3)]
    base = fx.problem(its, [fx.node_pool(limits={"cpu": "100"}, taints=[{"key": "k", "value": "v", "effect": "NoSchedule"}])], pods,
                      state_nodes=[fx.state_node("n1", its[0], "test-zone-1")], cluster_pods=[fx.pod(labels=lab, phase="Running", node_name="n1")],
                      daemonset_pods=[fx.pod(requests={"cpu": "100m"})], options={"reservedCapacity": True})

    def paths(obj, prefix=()):
        out = []
        items = obj.items() if isinstance(obj, dict) else enumerate(obj[:6]) if isinstance(obj, list) else ()
        for k, v in items:
            out.append(prefix + (k,))
            out += paths(v, prefix + (k,))
        return out

    outcomes = {"solved": 0, "refused": 0}
    for seed in range(200):
        rng = random.Random(seed)
        bad = copy.deepcopy(base)
        for _ in range(rng.randrange(1, 4)):
            path = rng.choice(paths(bad))
            cur = bad
            try:
                for k in path[:-1]:
                    cur = cur[k]
                r = rng.random()
                if r < 0.3:
                    cur.pop(path[-1])
                else:
                    cur[path[-1]] = rng.choice([None, "garbage", -1, [], {}, 1e308, "9" * 40])
            except (KeyError, IndexError, TypeError):
                pass
        try:
            NewScheduler(bad, solver_lib=emu).Solve()
            outcomes["solved"] += 1
        except (Unsupported, RuntimeError):
            outcomes["refused"] += 1
    assert outcomes["solved"] and outcomes["refused"]


def test_quantities_out_of_range_are_refused_not_wrapped(emu):
    """cpu "1e400000000" used to wrap to 0 in the flattener (the pod was then SCHEDULED as if it asked for nothing)
    after a 2^31-step loop; resource.Quantity saturates such a value, so the reference leaves the pod unschedulable. The
    flattener now refuses it loudly, quickly."""
    import time
    its = fx.fake_default_instance_types()
    for q in ("1e400000000", "1e41", "9" * 60, "99999999999999999999999999999999Ei", "1e-41"):
        t0 = time.time()
        with pytest.raises((Unsupported, RuntimeError)):
            NewScheduler(fx.problem(its, [fx.node_pool()], [fx.pod(requests={"cpu": q})]), solver_lib=emu).Solve()
        assert time.time() - t0 < 5
    # large but representable quantities still solve (and fail to fit, like in the reference)
    got = NewScheduler(fx.problem(its, [fx.node_pool()], [fx.pod(requests={"cpu": "1e12"}), fx.pod(requests={"memory": "8Ei"})]), solver_lib=emu).Solve()
    assert len(got["podErrors"]) == 2


def test_hostile_sizes_and_stray_resources(oracle, emu):
    # an overhead entry for a resource the capacity does not have is ignored (resources.Subtract keeps capacity's keys)
    its = fx.fake_default_instance_types()
    its[3]["overhead"] = {"example.com/widget": 1}
    check(oracle, emu, fx.problem(its, [fx.node_pool()], [fx.pod() for _ in range(3)]))
    # pod group counts: negative is invalid, absurd is refused before anything is allocated
    prob = fx.config2(pods=1000)
    prob["podGroups"][0]["count"] = -5
    with pytest.raises(RuntimeError, match="negative"):
        NewScheduler(prob, solver_lib=emu)
    prob["podGroups"][0]["count"] = 10**12
    with pytest.raises(Unsupported, match="pods in one problem"):
        NewScheduler(prob, solver_lib=emu)
    # out-of-range options fall back to their defaults
    prob = fx.config2(pods=1000)
    prob["options"].update({"maxClaims": -3, "ldsClaimCap": 10**9, "truncateInstanceTypes": -2})
    assert NewScheduler(prob, solver_lib=emu).Solve(want_results=False)["scheduledPods"] == 1000


def test_repeated_solves_on_one_handle_are_identical(oracle, emu):
    """A handle is reused for every Solve() of a scheduler (bench.py re-solves the resident inputs): whatever one solve
    leaves in the workspace — claims, dead rows, topology counters, reservation counters, the claim order, a cancelled
    run's partial state — the next solve must start from scratch."""
    its = reserved_types(2)
    lab = {"app": "x"}
    nodes = [fx.state_node(f"node-{i}", its[0], "test-zone-1", used={"cpu": "1", "pods": "1"}) for i in range(2)]
    cluster = [fx.pod(labels=lab, phase="Running", node_name="node-0", pod_anti_requirements=[fx.affinity_term(fx.HOSTNAME, lab)])]
    pods = [fx.pod(labels=lab, requests={"cpu": "700m"}, topology_spread=[fx.spread(fx.ZONE, lab)]) for _ in range(25)]
    pods += [fx.pod(labels=lab, requests={"cpu": "300m"}, pod_anti_requirements=[fx.affinity_term(fx.HOSTNAME, lab)]) for _ in range(6)]
    pods += [fx.pod(requests={"cpu": "1800m"}, node_preferences=[fx.req(fx.ZONE, "In", "test-zone-9")]) for _ in range(8)]
    prob = fx.problem(its, [fx.node_pool(limits={"cpu": "60"})], pods, state_nodes=nodes, cluster_pods=cluster, daemonset_pods=[fx.pod(requests={"cpu": "100m"})],
                      options={"reservedCapacity": True, "reservedOfferingMode": "Fallback", "ldsClaimCap": 16})
    want = oracle.solve(prob)
    s = NewScheduler(prob, solver_lib=emu)
    first = s.Solve()
    parity.assert_same_results(first, want)
    for _ in range(3):
        parity.assert_same_results(s.Solve(), first)
    # a cancelled run in between leaves nothing behind either
    import threading
    th = threading.Thread(target=s.Solve)
    th.start(); s.Cancel(); th.join()
    parity.assert_same_results(s.Solve(), first)


def test_row_hash_collisions_are_reported_never_merged(oracle, emu, monkeypatch):
    """Pod classing (csrc/kernels.h: row_hash_body → row_table_insert + row_diff_far). With the row hash narrowed to a few
    bits, distinct rows share a hash; the verification against the slot's representative has to report every one of them
    (the host re-seeds, and gives up after four attempts) — a merged class would be a wrong packing with no error."""
    prob = fx.config2(pods=600, n_types=50, seed=11)
    monkeypatch.setenv("KSOLVE_TEST_HASH_KEEP", "0x7")
    with pytest.raises(RuntimeError, match="row hash collisions persist"):
        NewScheduler(prob, solver_lib=emu).Solve()
    # all bits but one: a single class table slot chain per half, still exact
    monkeypatch.setenv("KSOLVE_TEST_HASH_KEEP", "0xFFFFFFFFFFFFFFFF")
    check(oracle, emu, prob)
    # identical rows only (one class): a narrowed hash cannot collide, and must not report
    monkeypatch.setenv("KSOLVE_TEST_HASH_KEEP", "0x1")
    same = fx.problem(fx.fake_default_instance_types(), [fx.node_pool()], [fx.pod(requests={"cpu": "1"}, node_selector=AMD) for _ in range(200)])
    check(oracle, emu, same)


def _complement_family_types():
    """Eight instance types whose requirement on a custom `family` key uses every operator: In, NotIn (Values() = the
    excluded ones), Exists (Values() = none), DoesNotExist, and no requirement at all."""
    fam = "example.com/family"
    shapes = [("a", fx.req(fam, "In", "m5")), ("b", fx.req(fam, "In", "c5", "c6")), ("c", fx.req(fam, "NotIn", "r5")), ("d", fx.req(fam, "NotIn", "m5", "x1")),
              ("e", fx.req(fam, "Exists")), ("f", None), ("g", fx.req(fam, "In", "m5")), ("h", fx.req(fam, "NotIn", "z9"))]
    its = []
    for i, (n, r) in enumerate(shapes):
        its.append(fx.fake_instance_type(f"fam-{n}", {"cpu": str(2 + 2 * (i % 4)), "memory": f"{4 + 4 * (i % 4)}Gi", "pods": "20"}, requirements=[r] if r else None))
    return fam, its


def test_min_values_on_a_key_instance_types_constrain_with_notin_exists(oracle, emu):
    """InstanceTypes.SatisfiesMinValues (types.go:399-433) unions `it.Requirements.Get(key).Values()`: for a NotIn requirement
    these are the EXCLUDED values, for Exists / DoesNotExist / an absent requirement none. The device counts a value when an In
    type has it or a complement type does not (engine.h distinct_values)."""
    fam, its = _complement_family_types()
    pods = [fx.pod(requests={"cpu": f"{c}m", "memory": f"{m}Mi"}) for c in (300, 1500, 2500, 5000) for m in (256, 3000, 9000) for _ in range(3)]
    seen_err, seen_ok = False, False
    for mv in (1, 2, 3, 4, 5, 6, 7, 9):
        for policy in ("Strict", "BestEffort"):
            for op, vals in (("Exists", ()), ("NotIn", ("q1",)), ("In", ("m5", "c5", "c6", "r5", "x1", "z9"))):
                pool = fx.node_pool(requirements=[fx.req(fam, op, *vals, min_values=mv)])
                got, _ = check(oracle, emu, fx.problem(its, [pool], pods, options={"minValuesPolicy": policy}))
                seen_err |= bool(got["podErrors"])
                seen_ok |= bool(got["newNodeClaims"])
    assert seen_err and seen_ok
    # pods that narrow the key themselves
    sel = [fx.pod(requests={"cpu": "500m"}, node_selector={fam: v}) for v in ("m5", "c5", "r5", "m5", "q1")] + pods[:6]
    for mv in (1, 2, 4):
        pool = fx.node_pool(requirements=[fx.req(fam, "Exists", min_values=mv)])
        check(oracle, emu, fx.problem(its, [pool], sel))
    # launch shaping: Truncate's re-check (scheduler.go:419-437) on the same catalogue
    pool = fx.node_pool(requirements=[fx.req(fam, "Exists", min_values=3)])
    check(oracle, emu, fx.problem(its, [pool], pods, options={"truncateInstanceTypes": 3}))


def _zone_of(claim):
    return next(tuple(sorted(r["values"])) for r in claim["requirements"] if r["key"] == fx.ZONE)


def test_volume_requirement_alternatives(oracle, emu):
    """PodData.VolumeRequirements (volumeReqsByPod, scheduler.go:138, :222, :572) in NodeClaim.CanAdd (nodeclaim.go:138-157,
    tryVolumeAlternative :164-242) and ExistingNode.CanAdd (existingnode.go:108-139, :143-168): the alternatives narrow the
    bin's requirements, not the pod's; they are tried in order. Known answers of provisioning/suite_test.go "Volume Topology
    Requirements" (:1987-2400) with the requirement sets VolumeTopology.GetRequirements derives from the fixtures there
    (storage class zones test-zone-2/3 = one alternative; a bound PersistentVolume in test-zone-3 = one alternative)."""
    its = fx.fake_default_instance_types()
    zone = lambda *z: fx.req(fx.ZONE, "In", *z)
    sc = [[zone("test-zone-2", "test-zone-3")]]
    pv3 = [[zone("test-zone-3")]]
    # :2103-2115 storage class zones ∩ the pod's own zones
    got, _ = check(oracle, emu, fx.problem(its, [fx.node_pool()], [fx.pod(node_requirements=[zone("test-zone-1", "test-zone-3")], volume_requirements=sc)]))
    assert [_zone_of(c) for c in got["newNodeClaims"]] == [("test-zone-3",)] and not got["podErrors"]
    # :2171-2181 the zone of a bound volume
    got, _ = check(oracle, emu, fx.problem(its, [fx.node_pool()], [fx.pod(volume_requirements=pv3)]))
    assert [_zone_of(c) for c in got["newNodeClaims"]] == [("test-zone-3",)]
    # :2317-2329 the pod wants another zone than its volume
    got, _ = check(oracle, emu, fx.problem(its, [fx.node_pool()], [fx.pod(node_requirements=[zone("test-zone-1")], volume_requirements=pv3)]))
    assert len(got["podErrors"]) == 1 and not got["newNodeClaims"]
    # :2290-2316 a volume pinned to a hostname: no NodeClaim carries that name
    got, _ = check(oracle, emu, fx.problem(its, [fx.node_pool()], [fx.pod(volume_requirements=[[fx.req(fx.HOSTNAME, "In", "some-node")]])]))
    assert len(got["podErrors"]) == 1 and not got["newNodeClaims"]
    # ... but an existing node with that name takes it, and Exists / NotIn on the hostname pass on a new claim
    by = {t["name"]: t for t in its}
    nodes = [fx.state_node("some-node", by["default-instance-type"], "test-zone-1"), fx.state_node("other-node", by["default-instance-type"], "test-zone-2")]
    got, _ = check(oracle, emu, fx.problem(its, [fx.node_pool()], [fx.pod(volume_requirements=[[fx.req(fx.HOSTNAME, "In", "some-node")]]),
                                                                   fx.pod(volume_requirements=[[fx.req(fx.HOSTNAME, "NotIn", "some-node", "other-node")]]),
                                                                   fx.pod(volume_requirements=[[fx.req(fx.HOSTNAME, "In", "gone")], [fx.req(fx.HOSTNAME, "In", "other-node")]])], state_nodes=nodes))
    assert {e["name"]: len(e["pods"]) for e in got["existingNodes"]} == {"some-node": 1, "other-node": 1} and len(got["newNodeClaims"]) == 1
    # :2353-2400 the volume's zone survives relaxation: the first required term cannot be met, the second can
    p = fx.pod(node_requirements=[[fx.req("example.com/label", "In", "unsupported")], [fx.req(fx.CAPACITY_TYPE, "In", "on-demand")]], volume_requirements=pv3)
    got, _ = check(oracle, emu, fx.problem(its, [fx.node_pool()], [p]))
    assert [_zone_of(c) for c in got["newNodeClaims"]] == [("test-zone-3",)] and not got["podErrors"]
    # alternatives in order: the first one the NodePool admits wins; pods with different lists do not share a class
    pool = fx.node_pool(requirements=[zone("test-zone-2", "test-zone-3")])
    alts = [[zone("test-zone-1")], [zone("test-zone-3")], [zone("test-zone-2")]]
    got, _ = check(oracle, emu, fx.problem(its, [pool], [fx.pod(requests={"cpu": "1"}, volume_requirements=alts) for _ in range(3)]
                                                        + [fx.pod(requests={"cpu": "1"}, volume_requirements=alts[::-1]) for _ in range(3)] + [fx.pod(requests={"cpu": "1"})]))
    assert sorted(_zone_of(c) for c in got["newNodeClaims"]) == [("test-zone-2",), ("test-zone-3",)]
    # an alternative can fail late: no instance type in its zone holds the pod (the filter, nodeclaim.go:213), the next one does
    big3 = [fx.fake_instance_type("big-z3", {"cpu": "32", "memory": "64Gi"}, offerings=[fx.offering("on-demand", "test-zone-3", 3.0)]),
            fx.fake_instance_type("small-all", {"cpu": "2", "memory": "4Gi"})]
    got, _ = check(oracle, emu, fx.problem(big3, [fx.node_pool()], [fx.pod(requests={"cpu": "10"}, volume_requirements=[[zone("test-zone-1")], [zone("test-zone-3")]]),
                                                                    fx.pod(requests={"cpu": "10"}, volume_requirements=[[zone("test-zone-1")], [zone("test-zone-2")]])]))
    assert [_zone_of(c) for c in got["newNodeClaims"]] == [("test-zone-3",)] and len(got["podErrors"]) == 1
    # topology counts with the pod's own requirements while the claim takes the volume's zone (nodeclaim.go:197-201)
    lab = {"app": "db"}
    pods = [fx.pod(labels=lab, requests={"cpu": "1"}, topology_spread=[fx.spread(fx.ZONE, lab)], volume_requirements=[[zone(z)], [zone("test-zone-2")]]) for z in ("test-zone-1", "test-zone-1", "test-zone-3", "test-zone-3", "test-zone-1")]
    check(oracle, emu, fx.problem(its, [fx.node_pool()], pods))
    # a custom label that only the volume mentions (undefined on the claim: allowed for well-known labels only, nodeclaim.go:171)
    got, _ = check(oracle, emu, fx.problem(its, [fx.node_pool()], [fx.pod(volume_requirements=[[fx.req("example.com/rack", "In", "r1")]]),
                                                                   fx.pod(volume_requirements=[[fx.req("example.com/rack", "NotIn", "r1")]]),
                                                                   fx.pod(volume_requirements=[[fx.req("example.com/rack", "In", "r1")], [zone("test-zone-2")]])]))
    assert len(got["podErrors"]) == 1
    check(oracle, emu, fx.problem(its, [fx.node_pool(labels={"example.com/rack": "r1"})], [fx.pod(volume_requirements=[[fx.req("example.com/rack", "In", "r1")]]),
                                                                                          fx.pod(volume_requirements=[[fx.req("example.com/rack", "In", "r2")]])]))


def test_volume_requirement_alternatives_fuzz(oracle, emu):
    zones = ["test-zone-1", "test-zone-2", "test-zone-3"]
    for seed in range(14):
        rng = random.Random(9100 + seed)
        its = fx.fake_instance_types(rng.choice([5, 12])) if seed % 2 else fx.fake_default_instance_types()
        by = {t["name"]: t for t in its}
        names = sorted(by)

        def alt():
            reqs = []
            if rng.random() < 0.8: reqs.append(fx.req(fx.ZONE, rng.choice(["In", "In", "NotIn"]), *rng.sample(zones, rng.choice([1, 1, 2]))))
            if rng.random() < 0.25: reqs.append(fx.req(fx.CAPACITY_TYPE, "In", rng.choice(["spot", "on-demand"])))
            if rng.random() < 0.2: reqs.append(fx.req("example.com/rack", rng.choice(["In", "NotIn", "Exists", "DoesNotExist"]), *([] if rng.random() < 0.3 else [rng.choice(["r1", "r2"])])))
            if rng.random() < 0.15: reqs.append(fx.req(fx.HOSTNAME, rng.choice(["In", "NotIn"]), rng.choice(["node-0", "node-1", "nowhere"])))
            for r in reqs:
                if r["operator"] in ("Exists", "DoesNotExist"): r["values"] = []
                elif not r["values"]: r["values"] = ["r1"]
            return reqs or [fx.req(fx.ZONE, "In", rng.choice(zones))]
        lists = [[alt() for _ in range(rng.choice([1, 1, 2, 3]))] for _ in range(4)]
        lab = {"app": "a"}
        pods = []
        for j in range(rng.choice([12, 40])):
            kw = {}
            if rng.random() < 0.3: kw["node_selector"] = {fx.ZONE: rng.choice(zones)}
            if rng.random() < 0.2: kw["node_preferences"] = [fx.req(fx.ZONE, "In", rng.choice(zones))]
            if rng.random() < 0.25: kw.update(labels=lab, topology_spread=[fx.spread(rng.choice([fx.ZONE, fx.HOSTNAME]), lab, when=rng.choice(["DoNotSchedule", "ScheduleAnyway"]))])
            pods.append(fx.pod(requests={"cpu": rng.choice(["100m", "500m", "1", "3"])}, volume_requirements=rng.choice(lists) if rng.random() < 0.7 else None, **kw))
        nodes = [fx.state_node(f"node-{i}", by[rng.choice(names)], rng.choice(zones), used={"cpu": rng.choice(["0", "1"])}, extra_labels=({"example.com/rack": rng.choice(["r1", "r2"])} if rng.random() < 0.5 else None))
                 for i in range(rng.choice([0, 0, 2, 5]))]
        pool = fx.node_pool(requirements=[fx.req(fx.ZONE, "In", *rng.sample(zones, rng.choice([2, 3])))] if rng.random() < 0.5 else None,
                            labels={"example.com/rack": "r1"} if rng.random() < 0.3 else None)
        got, _ = check(oracle, emu, fx.problem(its, [pool], pods, state_nodes=nodes))


@pytest.mark.parametrize("at", [1024, 5 * 1024, 37 * 1024, 100 * 1024, 150 * 1024])
def test_cancel_at_a_poll_boundary_is_the_full_run_stopped_there(emu, monkeypatch, at):
    """The cancel flag is polled at queue-block boundaries (every 1024 pods on the cursor engine, every 64 on the general one).
    KSOLVE_TEST_CANCEL_AT makes it land deterministically once `at` pods are placed: the Results must be the full run stopped
    after exactly that many pops — in particular the claim order is the one of the last sort the reference would have run
    (scheduler.go:598 sorts at the start of an add), not one re-sort further (the cursor engine's group path re-sorts eagerly)."""
    n = 200000
    for engine in ("auto", "general"):
        prob = fx.config2(pods=n, n_types=500, seed=42)
        prob["options"] = dict(prob.get("options") or {}, engine=engine)
        monkeypatch.setenv("KSOLVE_TEST_CANCEL_AT", str(at))
        got = NewScheduler(prob, solver_lib=emu).Solve()
        monkeypatch.delenv("KSOLVE_TEST_CANCEL_AT")
        assert got["timedOut"] and got["scheduledPods"] == at and got["counters"]["engine"] == ("cursor" if engine == "auto" else "general")
        prob["options"]["maxSteps"] = at
        parity.assert_same_results(got, NewScheduler(prob, solver_lib=emu).Solve())


def test_volume_requirement_alternatives_at_scale(oracle, emu):
    """A few thousand pods on the KWOK catalogue: a third of them with volume requirement alternatives over zones, some with a
    zonal spread constraint, a cluster of existing nodes in front — long enough for claims to fill up, be re-sorted and be
    skipped through the dead bits while alternatives are being tried (the pruning must stay exact with them)."""
    rng = random.Random(4242)
    its = fx.kwok_catalog(144)
    nodes = _cluster(its, rng, 40)
    lab = {"app": "store"}
    pods = []
    for j in range(5000):
        kw = {}
        if rng.random() < 0.33:
            zs = rng.sample(fx.KWOK_ZONES, rng.choice([1, 1, 2, 3]))
            kw["volume_requirements"] = [[fx.req(fx.ZONE, "In", z)] for z in zs] if rng.random() < 0.7 else [[fx.req(fx.ZONE, "In", *zs)]]
        if rng.random() < 0.1:
            kw.update(labels=lab, topology_spread=[fx.spread(fx.ZONE, lab, max_skew=rng.choice([1, 3]))])
        if rng.random() < 0.15:
            kw["node_selector"] = {fx.ZONE: rng.choice(fx.KWOK_ZONES)}
        pods.append(fx.pod(requests={"cpu": f"{rng.choice([500, 2000, 7000, 15000, 30000])}m", "memory": f"{rng.choice([256, 1024, 8192])}Mi"}, **kw))
    np_ = fx.node_pool("default")
    np_["nodeClassLabelKey"] = "karpenter.kwok.sh/kwoknodeclass"
    got, _ = check(oracle, emu, fx.problem(its, [np_], pods, well_known=fx.KWOK_WELL_KNOWN, state_nodes=nodes))
    assert len(got["newNodeClaims"]) > 100 and sum(len(e["pods"]) for e in got["existingNodes"]) > 0


def test_volume_requirement_alternatives_edge_cases(oracle, emu):
    """An alternative without requirements (admits everything), Exists / DoesNotExist and Gt / Lt inside an alternative (the
    bounds go through the working-set path of CanAdd), an alternative list that ends in the empty one."""
    its = fx.fake_default_instance_types()
    zone = lambda *z: fx.req(fx.ZONE, "In", *z)
    pods = [fx.pod(volume_requirements=[[]]), fx.pod(volume_requirements=[[], [zone("test-zone-1")]]),
            fx.pod(node_selector={fx.ZONE: "test-zone-2"}, volume_requirements=[[zone("test-zone-1")], []]),
            fx.pod(volume_requirements=[[fx.req(fx.ZONE, "Exists")]]), fx.pod(volume_requirements=[[fx.req(fx.ZONE, "DoesNotExist")]]),
            fx.pod(volume_requirements=[[fx.req(fx.FAKE_INTEGER_LABEL, "Gt", "2")]]),
            fx.pod(volume_requirements=[[fx.req(fx.FAKE_INTEGER_LABEL, "Lt", "1")], [fx.req(fx.FAKE_INTEGER_LABEL, "Gt", "3")]])]
    got, _ = check(oracle, emu, fx.problem(its, [fx.node_pool()], pods))
    assert len(got["podErrors"]) == 1 and pods[4]["uid"] in got["podErrors"]      # a zone that must not exist: every offering has one
    by = {t["name"]: t for t in its}
    nodes = [fx.state_node("n1", by["default-instance-type"], "test-zone-1"), fx.state_node("n2", by["arm-instance-type"], "test-zone-2")]
    check(oracle, emu, fx.problem(its, [fx.node_pool()], pods, state_nodes=nodes))


def test_a_rejected_class_comes_back_when_its_key_becomes_defined(oracle, emu):
    """The per-(class, bin) rejections the scan keeps are final while a bin only narrows — except for the undefined-key rule of
    Requirements.Compatible (requirements.go:185-193): `rack In [r1]` on a custom label the NodeClaim does not carry is
    rejected, a later pod's `rack NotIn [r2]` is not (negative operators may meet an undefined key) and DEFINES the key on the
    claim (Add, requirements.go:133-140), after which `In [r1]` intersects it. The pod that failed is retried by the queue
    (queue.go:52-66) and must find the bin open again; the same on an existing node (existingnode.go:100-106)."""
    its = fx.fake_default_instance_types()
    rack = "example.com/rack"
    first = fx.pod(requests={"cpu": "2"})
    wants = [fx.pod(requests={"cpu": "1500m"}, node_requirements=[fx.req(rack, "In", "r1")]) for _ in range(2)]
    opens = fx.pod(requests={"cpu": "1"}, node_requirements=[fx.req(rack, "NotIn", "r2")])
    exists = fx.pod(requests={"cpu": "900m"}, node_requirements=[fx.req(rack, "Exists")])
    got, _ = check(oracle, emu, fx.problem(its, [fx.node_pool()], [first] + wants + [opens, exists]))
    on = {u: c["hostname"] for c in got["newNodeClaims"] for u in c["pods"]}
    assert not got["podErrors"] and on[wants[0]["uid"]] == on[opens["uid"]]
    # without the NotIn pod the key never appears: the In / Exists pods cannot be scheduled at all
    got, _ = check(oracle, emu, fx.problem(its, [fx.node_pool()], [first] + wants + [exists]))
    assert len(got["podErrors"]) == 3
    # an existing node: the key is not among its labels
    by = {t["name"]: t for t in its}
    node = fx.state_node("node-a", by["arm-instance-type"], "test-zone-1")
    got, _ = check(oracle, emu, fx.problem(its, [fx.node_pool()], [first] + wants + [opens, exists], state_nodes=[node]))
    assert not got["podErrors"] and sum(len(e["pods"]) for e in got["existingNodes"]) >= 3
    # many of them, with zonal spread narrowing the claims in between (a change that defines no key)
    lab = {"app": "z"}
    pods = []
    for i in range(60):
        pods.append(fx.pod(requests={"cpu": f"{2000 - 10 * i}m"}, node_requirements=[fx.req(rack, "In", f"r{i % 3}")]))
        pods.append(fx.pod(requests={"cpu": f"{900 - 5 * i}m"}, node_requirements=[fx.req(rack, "NotIn", f"r{(i + 1) % 3}")]))
        pods.append(fx.pod(labels=lab, requests={"cpu": f"{700 - 5 * i}m"}, topology_spread=[fx.spread(fx.ZONE, lab)]))
    check(oracle, emu, fx.problem(its, [fx.node_pool()], pods))
    check(oracle, emu, fx.problem(its, [fx.node_pool()], pods, state_nodes=[node, fx.state_node("node-b", by["arm-instance-type"], "test-zone-2")]))


def test_a_rejected_class_comes_back_when_exists_becomes_notin(oracle, emu):
    """The second way a rejected class comes back (round-2 advisor finding): Intersects lets a NotIn / DoesNotExist pair through
    whatever the values (requirements.go:258-265), and the claim's Operator() depends on its current value set
    (requirement.go:290-301). A claim holding `rack Exists` (from the NodePool) rejects a `rack DoesNotExist` pod; a
    `rack NotIn [x]` pod narrows the claim to NotIn [x]; the DoesNotExist pod, retried by the queue (queue.go:52-66), is now
    compatible with that claim — the defined-key mask never changed, so only the operator-class rule drops the stale verdict."""
    its = fx.fake_default_instance_types()
    rack = "example.com/rack"
    for op, vals in (("Exists", []), ("Gt", ["1"]), ("Lt", ["9"])):
        pool = fx.node_pool(requirements=[fx.req(rack, op, *vals)])
        first = fx.pod(requests={"cpu": "900m"})
        absent = fx.pod(requests={"cpu": "700m"}, node_requirements=[fx.req(rack, "DoesNotExist")])
        notin = fx.pod(requests={"cpu": "300m"}, node_requirements=[fx.req(rack, "NotIn", "5")])
        got, _ = check(oracle, emu, fx.problem(its, [pool], [first, absent, notin]))
        on = {u: c["hostname"] for c in got["newNodeClaims"] for u in c["pods"]}
        assert not got["podErrors"] and on[absent["uid"]] == on[first["uid"]] == on[notin["uid"]], op
    # many claims, with In pods and zonal spread narrowing them in between
    lab = {"app": "z"}
    pods = []
    for i in range(50):
        pods.append(fx.pod(requests={"cpu": f"{2000 - 10 * i}m"}, node_requirements=[fx.req(rack, "DoesNotExist")]))
        pods.append(fx.pod(requests={"cpu": f"{900 - 5 * i}m"}, node_requirements=[fx.req(rack, "NotIn", f"r{i % 3}")]))
        pods.append(fx.pod(requests={"cpu": f"{800 - 5 * i}m"}, node_requirements=[fx.req(rack, "In", f"r{(i + 1) % 3}")]))
        pods.append(fx.pod(labels=lab, requests={"cpu": f"{700 - 5 * i}m"}, topology_spread=[fx.spread(fx.ZONE, lab)]))
    check(oracle, emu, fx.problem(its, [fx.node_pool(requirements=[fx.req(rack, "Exists")])], pods))


def test_undefined_key_revival_fuzz(oracle, emu):
    """Custom labels that no NodePool or instance type defines, under every operator: In / Exists / Gt fail on a bin until a
    NotIn / DoesNotExist pod has defined the key there, the queue retries the failed pods (queue.go:52-66) — the rejections
    the scan remembers must be dropped exactly when a key becomes defined. Claims and existing nodes, with zonal spread
    narrowing the bins in between."""
    keys = ["example.com/rack", "example.com/tier"]
    revived = 0
    for seed in range(24):
        rng = random.Random(5200 + seed)
        its = fx.fake_default_instance_types() if seed % 2 else fx.fake_instance_types(rng.choice([6, 12]))
        by = {t["name"]: t for t in its}
        lab = {"app": "s"}
        pods = []
        for i in range(rng.choice([30, 70])):
            kw = {}
            r = rng.random()
            if r < 0.55:
                k = rng.choice(keys)
                op = rng.choice(["In", "In", "NotIn", "NotIn", "Exists", "DoesNotExist", "Gt"])
                vals = [] if op in ("Exists", "DoesNotExist") else (["2"] if op == "Gt" else rng.sample(["1", "3", "5", "x"], rng.choice([1, 2])))
                kw["node_requirements"] = [fx.req(k, op, *vals)]
            elif r < 0.7:
                kw.update(labels=lab, topology_spread=[fx.spread(fx.ZONE, lab)])
            pods.append(fx.pod(requests={"cpu": f"{rng.choice([100, 300, 700, 1500])}m"}, **kw))
        nodes = [fx.state_node(f"node-{i}", by[rng.choice(sorted(by))], rng.choice(["test-zone-1", "test-zone-2"]),
                               extra_labels=({keys[0]: "3"} if rng.random() < 0.3 else None)) for i in range(rng.choice([0, 2]))]
        pool = fx.node_pool(labels={keys[1]: "1"} if rng.random() < 0.25 else None)
        if seed % 3 == 2:   # the key is on the claim from the start under an Exists-class operator: revival through Exists -> NotIn
            pool = fx.node_pool(requirements=[fx.req(keys[0], *rng.choice([("Exists",), ("Gt", "0"), ("Lt", "7")]))])
        got, _ = check(oracle, emu, fx.problem(its, [pool], pods, state_nodes=nodes))
        placed = {u for c in got["newNodeClaims"] for u in c["pods"]} | {u for e in got["existingNodes"] for u in e["pods"]}
        revived += sum(1 for p in pods if p["uid"] in placed and any(r["operator"] in ("In", "Exists", "Gt") and r["key"] in keys for term in (p.get("nodeAffinity") or {}).get("required", []) for r in term))
    assert revived > 20     # positive operators on an undefined custom key got a bin: only possible through a key that became defined


def test_rehydration_by_position_equals_the_results_document(oracle, emu):
    """Scheduler.Assignment / PodsByClaim (ksched_assignment): the flat pod_assignment / pod_slot arrays of a Solve(want_results=
    "claims") put every pod on the same NodeClaim, in the same slot, as the Results document with its uid lists — with existing
    nodes and an unschedulable pod in the problem."""
    import numpy as np
    prob = fx.config2(pods=6000, n_types=144, seed=11)
    prob["pods"] = [fx.pod(uid="00000000-0000-0000-0000-00000000dead", requests={"cpu": "4000"})]      # fits nothing
    s = NewScheduler(prob, solver_lib=emu)
    full = s.Solve()
    lean = s.Solve(want_results="claims")
    assert [c["podCount"] for c in lean["newNodeClaims"]] == [len(c["pods"]) for c in full["newNodeClaims"]] and not any(c["pods"] for c in lean["newNodeClaims"])
    uids = [p["uid"] for p in prob["pods"]] + [fx.group_pod_uid(g["uidSeed"], i) for g in prob["podGroups"] for i in range(g["count"])]
    by = s.PodsByClaim(len(lean["newNodeClaims"]))
    assert [[uids[i] for i in pods] for pods in by] == [c["pods"] for c in full["newNodeClaims"]]
    assign, _ = s.Assignment()
    assert assign[0] == -1 and list(full["podErrors"]) == [uids[0]] and int((assign == -1).sum()) == 1
    s.close()
    parity.assert_same_results(full, oracle.solve(prob))
