"""Known-answer scheduling tests for the CPU oracle, transcribed from the reference's functional tests (which need a
kube-apiserver in the reference; here only the scheduling decision is checked):
  pkg/controllers/provisioning/scheduling/suite_test.go        Binpacking :1521-1827, custom constraints :953-1110
  pkg/controllers/provisioning/scheduling/topology_test.go     zonal / hostname spread :110-653, anti-affinity
  pkg/controllers/provisioning/scheduling/instance_selection_test.go :40-  (cheapest instance type)
"""
import pytest
import collections

from karpenter_amd import fixtures as fx

AMD = {fx.ARCH: "amd64"}


def cheapest_type(claim, catalog):
    price = {t["name"]: min(o["price"] for o in t["offerings"]) for t in catalog}
    return min(claim["instanceTypes"], key=lambda n: (price[n], n))


def solve(oracle, pods, its=None, pools=None, **kw):
    its = its or fx.fake_default_instance_types()
    return oracle.solve(fx.problem(its, pools or [fx.node_pool()], pods, **kw)), its


def test_binpacking_small_pod_smallest_instance(oracle):
    r, its = solve(oracle, [fx.pod(requests={"memory": "100M"})])
    assert len(r["newNodeClaims"]) == 1 and cheapest_type(r["newNodeClaims"][0], its) == "small-instance-type"  # suite_test.go:1522-1533
    r, its = solve(oracle, [fx.pod(requests={"memory": "2000M"})])
    assert cheapest_type(r["newNodeClaims"][0], its) == "small-instance-type"                                   # :1534-1545


def test_binpacking_new_nodes_at_capacity(oracle):
    pods = [fx.pod(requests={"memory": "1.8G"}, node_selector=AMD) for _ in range(40)]
    r, its = solve(oracle, pods)
    assert len(r["newNodeClaims"]) == 20                                                                         # :1593-1612
    assert all(len(c["pods"]) == 2 and cheapest_type(c, its) == "default-instance-type" for c in r["newNodeClaims"])
    pods += [fx.pod(requests={"memory": "400M"}, node_selector=AMD) for _ in range(20)]
    r, its = solve(oracle, pods)
    assert len(r["newNodeClaims"]) == 20 and not r["podErrors"]                                                  # :1613-1644
    assert all(cheapest_type(c, its) == "default-instance-type" for c in r["newNodeClaims"])


def test_binpacking_pod_limit_per_node(oracle):
    pods = [fx.pod(requests={"cpu": "1m", "memory": "1m"}, node_selector=AMD) for _ in range(25)]
    r, its = solve(oracle, pods)
    assert len(r["newNodeClaims"]) == 5 and all(cheapest_type(c, its) == "small-instance-type" for c in r["newNodeClaims"])  # :1694-1715


def test_binpacking_pack_tightly_and_oversized(oracle):
    its = fx.fake_instance_types(5)
    r, _ = solve(oracle, [fx.pod(requests={"cpu": "4.5"}), fx.pod(requests={"cpu": "1"})], its=its)
    assert len(r["newNodeClaims"]) == 2                                                                          # :1645-1669
    assert cheapest_type(r["newNodeClaims"][0], its) != cheapest_type(r["newNodeClaims"][1], its)
    r, _ = solve(oracle, [fx.pod(requests={"memory": "2Ti"})])
    assert len(r["newNodeClaims"]) == 0 and len(r["podErrors"]) == 1                                            # :1681-1692
    r, _ = solve(oracle, [fx.pod(requests={"foo.com/weird-resources": "0"})])
    assert len(r["newNodeClaims"]) == 1 and not r["podErrors"]                                                  # :1670-1680


def test_custom_label_operator_matrix(oracle):
    # suite_test.go:953-1068 — operators on a custom key that the NodePool does / does not define
    def one(expr, pool_labels=None):
        r, _ = solve(oracle, [fx.pod(node_requirements=[expr])], pools=[fx.node_pool(labels=pool_labels or {})])
        return not r["podErrors"]
    key = "test-key"
    assert not one(fx.req(key, "In", "test-value"))            # undefined key, In      -> does not schedule
    assert one(fx.req(key, "NotIn", "test-value"))             # undefined key, NotIn   -> schedules
    assert not one(fx.req(key, "Exists"))                      # undefined key, Exists  -> does not schedule
    assert one(fx.req(key, "DoesNotExist"))                    # undefined key, DNE     -> schedules
    lab = {key: "test-value"}
    assert one(fx.req(key, "In", "test-value"), lab)
    assert not one(fx.req(key, "In", "another-value"), lab)
    assert not one(fx.req(key, "NotIn", "test-value"), lab)
    assert one(fx.req(key, "NotIn", "another-value"), lab)
    assert one(fx.req(key, "Exists"), lab)
    assert not one(fx.req(key, "DoesNotExist"), lab)


def test_compatible_pods_share_a_node(oracle):
    # suite_test.go:1070-1110
    key = "test-key"
    pools = [fx.node_pool(requirements=[fx.req(key, "In", "test-value", "another-value")])]
    r, _ = solve(oracle, [fx.pod(node_requirements=[fx.req(key, "In", "test-value")]), fx.pod(node_requirements=[fx.req(key, "NotIn", "another-value")])], pools=pools)
    assert len(r["newNodeClaims"]) == 1
    r, _ = solve(oracle, [fx.pod(node_requirements=[fx.req(key, "In", "test-value")]), fx.pod(node_requirements=[fx.req(key, "In", "another-value")])], pools=pools)
    assert len(r["newNodeClaims"]) == 2


def test_gt_lt_node_affinity(oracle):
    # fake instance types carry the integer label = cpu count (fake/instancetype.go:163)
    its = fx.fake_instance_types(8)
    r, _ = solve(oracle, [fx.pod(node_requirements=[fx.req(fx.FAKE_INTEGER_LABEL, "Gt", "6")])], its=its)
    assert sorted(r["newNodeClaims"][0]["instanceTypes"]) == ["fake-it-6", "fake-it-7"]
    r, _ = solve(oracle, [fx.pod(node_requirements=[fx.req(fx.FAKE_INTEGER_LABEL, "Lt", "3")])], its=its)
    assert sorted(r["newNodeClaims"][0]["instanceTypes"]) == ["fake-it-0", "fake-it-1"]
    r, _ = solve(oracle, [fx.pod(node_requirements=[fx.req(fx.FAKE_INTEGER_LABEL, "Gt", "100")])], its=its)
    assert r["podErrors"]


def test_preference_relaxation_order(oracle):
    # suite_test.go:1126-1246: an unsatisfiable preferred node affinity is dropped, the required one is kept
    p = fx.pod(node_requirements=[fx.req(fx.ZONE, "In", "test-zone-3")], node_preferences=[fx.req(fx.ZONE, "In", "invalid")])
    r, _ = solve(oracle, [p])
    assert not r["podErrors"] and r["counters"]["relaxations"] == 1
    zone = [q for q in r["newNodeClaims"][0]["requirements"] if q["key"] == fx.ZONE][0]
    assert zone["values"] == ["test-zone-3"]
    # multiple required terms are OR'd: first term impossible -> relaxed to the second
    p = fx.pod(node_requirements=[[fx.req(fx.ZONE, "In", "invalid")], [fx.req(fx.ZONE, "In", "test-zone-2")]])
    r, _ = solve(oracle, [p])
    zone = [q for q in r["newNodeClaims"][0]["requirements"] if q["key"] == fx.ZONE][0]
    assert not r["podErrors"] and zone["values"] == ["test-zone-2"]


def test_taints_and_tolerations(oracle):
    pools = [fx.node_pool(taints=[{"key": "dedicated", "value": "x", "effect": "NoSchedule"}])]
    r, _ = solve(oracle, [fx.pod()], pools=pools)
    assert r["podErrors"]
    for tol in ({"key": "dedicated", "operator": "Exists"}, {"key": "dedicated", "operator": "Equal", "value": "x", "effect": "NoSchedule"}, {"operator": "Exists"}):
        r, _ = solve(oracle, [fx.pod(tolerations=[tol])], pools=pools)
        assert not r["podErrors"], tol
    r, _ = solve(oracle, [fx.pod(tolerations=[{"key": "dedicated", "operator": "Equal", "value": "y"}])], pools=pools)
    assert r["podErrors"]


def test_nodepool_weight_order_and_limits(oracle):
    pools = [fx.node_pool("low", weight=1), fx.node_pool("high", weight=10)]
    r, _ = solve(oracle, [fx.pod()], pools=pools)
    assert r["newNodeClaims"][0]["nodePool"] == "high"            # OrderByWeight — nodepool.go:161-171
    pools = [fx.node_pool("low", weight=1), fx.node_pool("high", weight=10, limits={"cpu": "4"})]
    pods = [fx.pod(requests={"cpu": "3"}, node_selector=AMD) for _ in range(3)]
    r, _ = solve(oracle, pods, pools=pools)
    # the first claim pessimistically consumes the largest surviving type's capacity (subtractMax, scheduler.go:1049)
    assert [c["nodePool"] for c in r["newNodeClaims"]].count("high") == 1 and not r["podErrors"]


def test_zonal_topology_spread(oracle):
    # topology_test.go:110-124: 4 pods, maxSkew 1 over three zones -> skew multiset (1,1,2)
    lab = {"test": "test"}
    pods = [fx.pod(labels=lab, topology_spread=[fx.spread(fx.ZONE, lab)]) for _ in range(4)]
    r, _ = solve(oracle, pods)
    assert not r["podErrors"]
    cnt = collections.Counter()
    for c in r["newNodeClaims"]:
        zone = [q for q in c["requirements"] if q["key"] == fx.ZONE][0]
        assert len(zone["values"]) == 1
        cnt[zone["values"][0]] += len(c["pods"])
    assert sorted(cnt.values()) == [1, 1, 2]


def test_hostname_spread_and_anti_affinity(oracle):
    lab = {"test": "test"}
    pods = [fx.pod(labels=lab, topology_spread=[fx.spread(fx.HOSTNAME, lab)]) for _ in range(4)]
    r, _ = solve(oracle, pods)
    assert sorted(len(c["pods"]) for c in r["newNodeClaims"]) == [1, 1, 1, 1]        # topology_test.go:547-560
    pods = [fx.pod(labels=lab, pod_anti_requirements=[fx.affinity_term(fx.HOSTNAME, lab)]) for _ in range(3)]
    r, _ = solve(oracle, pods)
    assert len(r["newNodeClaims"]) == 3 and not r["podErrors"]
    # topology_test.go:2502-2531 (Schrödinger): a zonal anti-affinity pod without a zone could be in ANY zone, so a second
    # pod it selects cannot schedule in the same batch
    pods = [fx.pod(labels=lab, pod_anti_requirements=[fx.affinity_term(fx.ZONE, lab)]) for _ in range(5)]
    r, _ = solve(oracle, pods)
    assert len(r["newNodeClaims"]) == 1 and len(r["podErrors"]) == 4
    # topology_test.go:2466-2500 (inverse): three zone-pinned anti-affinity pods schedule; the pod they select cannot
    aff = {"security": "s2"}
    anti = [fx.affinity_term(fx.ZONE, aff)]
    zp = [fx.pod(requests={"cpu": "2"}, pod_anti_requirements=anti, node_selector={fx.ZONE: f"test-zone-{i}"}) for i in (1, 2, 3)]
    victim = fx.pod(labels=aff)
    r, _ = solve(oracle, zp + [victim])
    assert len(r["newNodeClaims"]) == 3 and list(r["podErrors"]) == [victim["uid"]]


def test_self_affinity_colocates(oracle):
    lab = {"app": "a"}
    pods = [fx.pod(labels=lab, requests={"cpu": "1"}, pod_requirements=[fx.affinity_term(fx.ZONE, lab)]) for _ in range(6)]
    r, _ = solve(oracle, pods)
    zones = set()
    for c in r["newNodeClaims"]:
        zones.update([q for q in c["requirements"] if q["key"] == fx.ZONE][0]["values"])
    assert not r["podErrors"] and len(zones) == 1


def test_benchmark_diverse_pods_all_schedule(oracle):
    # scheduling_benchmark_test.go:176-181: every pod of the diverse mix must schedule against 400 fake types
    import random
    rng = random.Random(42)
    rl = lambda: {"my-label": rng.choice("abcdefg")}
    res = lambda: {"cpu": f"{rng.choice(fx.BENCH_CPU_M)}m", "memory": f"{rng.choice(fx.BENCH_MEM_MI)}Mi"}
    pods = [fx.pod(labels=rl(), requests=res()) for _ in range(60)]
    for key in (fx.ZONE, fx.HOSTNAME):
        pods += [fx.pod(labels=rl(), requests=res(), topology_spread=[fx.spread(key, rl())]) for _ in range(60)]
    for _ in range(60):
        lab = {"my-affininity": rng.choice("abcdefg")}
        pods.append(fx.pod(labels=lab, requests=res(), pod_requirements=[fx.affinity_term(fx.ZONE, lab)]))
    pods += [fx.pod(labels={"app": "nginx"}, requests=res(), pod_anti_requirements=[fx.affinity_term(fx.HOSTNAME, {"app": "nginx"})]) for _ in range(60)]
    r = oracle.solve(fx.problem(fx.fake_instance_types(400), [fx.node_pool(limits={"cpu": "10000000", "memory": "10000000Gi"})], pods))
    assert not r["podErrors"] and len(r["newNodeClaims"]) >= 60


def test_reserved_instance_types(oracle):
    # suite_test.go:4715-4765: a claim compatible with both reserved types reserves both, so one pod schedules per loop;
    # once both reservations are used up the last pod falls back to on-demand/spot
    def types(cap_medium, cap_small):
        its = [fx.fake_instance_type(n, resources={"cpu": str(c), "memory": f"{c}Gi"}) for n, c in (("large-instance-type", 6), ("medium-instance-type", 3), ("small-instance-type", 2))]
        for it, cap in ((its[1], cap_medium), (its[2], cap_small)):
            for r in it["requirements"]:
                if r["key"] == fx.CAPACITY_TYPE:
                    r["values"].append("reserved")
            it["offerings"].append(fx.offering("reserved", "test-zone-1", fx.fake_price(it["capacity"]) / 100000.0, reservation_id="r-" + it["name"],
                                               reservation_capacity=max(cap, 0), available=cap > 0))
        return its
    opts = {"reservedCapacity": True, "reservedOfferingMode": "Strict"}
    pods = [fx.pod(requests={"cpu": "1800m"}) for _ in range(3)]
    r = oracle.solve(fx.problem(types(1, 1), [fx.node_pool()], pods, options=opts))
    assert len(r["newNodeClaims"]) == 1 and len(r["newNodeClaims"][0]["pods"]) == 1 and len(r["podErrors"]) == 2
    assert sorted(r["newNodeClaims"][0]["reservedOfferings"]) == ["r-medium-instance-type", "r-small-instance-type"]
    # second loop: the small reservation is gone (unavailable), the claim reserves the medium one
    r = oracle.solve(fx.problem(types(1, 0), [fx.node_pool()], pods[1:], options=opts))
    assert len(r["newNodeClaims"]) == 1 and r["newNodeClaims"][0]["reservedOfferings"] == ["r-medium-instance-type"] and len(r["podErrors"]) == 1
    # third loop: no reservation left, falls back to on-demand / spot
    r = oracle.solve(fx.problem(types(0, 0), [fx.node_pool()], pods[2:], options=opts))
    assert len(r["newNodeClaims"]) == 1 and not r["newNodeClaims"][0]["reservedOfferings"] and not r["podErrors"]


# ---- offering capacity / overhead overrides (types.go:202-269; the device answers KSOLVE_ERR_UNSUPPORTED for these) ----

def _with_override_offerings(it, available=True, capacity=None, overhead=None):
    """suite_test.go:5532-5545: every base offering cloned with CapacityOverride / OverheadOverride."""
    clones = []
    for o in list(it["offerings"]):
        c = dict(o, available=available)
        if capacity is not None:
            c["capacityOverride"] = dict(capacity)
        if overhead is not None:
            c["overheadOverride"] = dict(overhead)       # InstanceTypeOverhead.Total() of the override
        clones.append(c)
    it["offerings"] = it["offerings"] + clones
    return it


def test_offering_overrides_known_answers(oracle):
    ext = "test.com/extended-slots"
    res = {"cpu": "4", "memory": "8Gi"}
    # suite_test.go:5524-5566: only the type whose override offerings carry the extended resource is selected
    ov = _with_override_offerings(fx.fake_instance_type("override-capable", res), capacity={ext: "4"}, overhead={"memory": "1Gi"})
    normal = fx.fake_instance_type("normal", res)
    r, _ = solve(oracle, [fx.pod(requests={ext: "1"})], its=[ov, normal])
    assert not r["podErrors"] and len(r["newNodeClaims"]) == 1
    assert r["newNodeClaims"][0]["instanceTypes"] == ["override-capable"]
    # suite_test.go:5568-5607: the override allocatable would fit, but its offerings are unavailable: no NodeClaim
    ov = _with_override_offerings(fx.fake_instance_type("override-capable", res), available=False, capacity={ext: "4"}, overhead={"memory": "1Gi"})
    r, _ = solve(oracle, [fx.pod(requests={ext: "1"})], its=[ov])
    assert len(r["podErrors"]) == 1 and not r["newNodeClaims"]


def test_offering_override_groups_semantics(oracle):
    """groupOfferingsByOverride + fits (types.go:224-269, nodeclaim.go:624-638): a type fits when SOME group both fits the
    requests and has an offering compatible with the requirements; the base group keeps the base allocatable."""
    ext = "test.com/extended-slots"
    res = {"cpu": "4", "memory": "8Gi"}
    # the override group trades memory for slots (overhead 1Gi -> 7Gi allocatable): a 7.5Gi pod only fits the base group,
    # a slot pod only the override group; both groups live on one type, so each pod gets a claim of that type, but they
    # cannot share one (requests 7.5Gi + slots fit neither group)
    ov = _with_override_offerings(fx.fake_instance_type("t", res), capacity={ext: "4"}, overhead={"memory": "1Gi"})
    big, slot = fx.pod(requests={"memory": "7680Mi"}), fx.pod(requests={ext: "1"})
    r, _ = solve(oracle, [big, slot], its=[ov])
    assert not r["podErrors"] and sorted(len(c["pods"]) for c in r["newNodeClaims"]) == [1, 1]
    r, _ = solve(oracle, [fx.pod(requests={ext: "1", "memory": "1Gi"}) for _ in range(4)] + [fx.pod(requests={ext: "1"})], its=[ov])
    assert not r["podErrors"] and sorted(len(c["pods"]) for c in r["newNodeClaims"]) == [1, 4]          # four slots per node
    # override offerings only in test-zone-3: a pod that needs slots AND zone 1 has a fitting group without a compatible
    # offering and a compatible offering in a group that does not fit -> unschedulable; zone 3 works
    t = fx.fake_instance_type("t", res)
    z3 = [dict(o, capacityOverride={ext: "2"}) for o in t["offerings"] if any(q["key"] == fx.ZONE and q["values"] == ["test-zone-3"] for q in o["requirements"])]
    t["offerings"] += z3
    r, _ = solve(oracle, [fx.pod(requests={ext: "1"}, node_selector={fx.ZONE: "test-zone-1"})], its=[t])
    assert len(r["podErrors"]) == 1
    r, _ = solve(oracle, [fx.pod(requests={ext: "1"})], its=[t])
    assert not r["podErrors"]
    # fits() does not narrow the claim's zone requirement (the launch picks the offering): no zone requirement appears
    assert not [q for q in r["newNodeClaims"][0]["requirements"] if q["key"] == fx.ZONE]
    # a capacity override replaces the whole key (lo.Assign): memory 2Gi instead of 8Gi in that group; an empty override map
    # with no overhead override is the base group
    t = _with_override_offerings(fx.fake_instance_type("t", res), capacity={"memory": "2Gi"})
    for o in t["offerings"][:5]:
        o["available"] = False                                            # only the shrunken group can launch
    r, _ = solve(oracle, [fx.pod(requests={"memory": "3Gi"})], its=[t])
    assert len(r["podErrors"]) == 1
    t = _with_override_offerings(fx.fake_instance_type("t", res), capacity={})
    for o in t["offerings"][:5]:
        o["available"] = False
    r, _ = solve(oracle, [fx.pod(requests={"memory": "3Gi"})], its=[t])
    assert not r["podErrors"]


@pytest.mark.parametrize("shape", ["config2", "config3", "config4"])
def test_parallel_in_flight_scan_is_the_sequential_scan(oracle, monkeypatch, shape):
    """ORACLE_THREADS fans the candidates of addToInflightNode out over a worker pool as the reference's parallelizeUntil does
    (scheduler.go:939-961, :667-686: the lowest index that succeeds wins) — what the offline pins of the largest configurations are
    made with. Same Results document, same counters as the sequential scan, with the fan-out forced on from the fourth claim."""
    prob = {"config2": lambda: fx.config2(pods=3000, n_types=40, seed=5), "config3": lambda: fx.config3(pods=1500, n_types=30, seed=5),
            "config4": lambda: fx.config4(pods=4000, n_types=60, n_pools=5, seed=5)}[shape]()
    want = oracle.solve(prob)
    monkeypatch.setenv("ORACLE_THREADS", "4")
    monkeypatch.setenv("ORACLE_PAR_MIN", "4")
    got = oracle.solve(prob)
    assert len(want["newNodeClaims"]) > 8
    for k in ("newNodeClaims", "podErrors", "packingCost"):
        assert got[k] == want[k], k
    for k in ("binEvaluations", "instanceTypeEvaluations"):
        if k in want["counters"]:
            assert got["counters"][k] == want["counters"][k], k
