"""GPU parity tests (run with -m gpu on an MI355X): the product path — karpenter_amd/libksolve.so (HIP) through the C ABI
— against the oracle on the same seeded inputs, bit-exact (L1-strict), plus size-independent properties at full size."""
import collections
import os

import pytest

import parity
from karpenter_amd import fixtures as fx
from karpenter_amd.scheduling import NewScheduler, device_available

pytestmark = pytest.mark.gpu
AMD = {fx.ARCH: "amd64"}


@pytest.fixture(scope="module", autouse=True)
def built():
    import __graft_entry__
    __graft_entry__.build()
    assert device_available(), "GPU tests need a usable gfx950 device and karpenter_amd/libksolve.so (no CPU fallback)"


def check(oracle, prob, solver_lib=None):
    """The engine the library picks on its own (the cursor engine for purely positive provisioning batches, the general
    engine otherwise) AND — when that was the cursor engine — the general engine too: both against the oracle."""
    want = oracle.solve(prob)
    got = NewScheduler(prob, solver_lib=solver_lib).Solve()
    parity.assert_same_results(got, want)
    assert got["counters"]["referenceBinEvaluations"] == want["counters"]["binEvaluations"]
    if got["counters"]["engine"] == "cursor":
        gen = NewScheduler(dict(prob, options=dict(prob["options"], engine="general")), solver_lib=solver_lib).Solve()
        assert gen["counters"]["engine"] == "general"
        parity.assert_same_results(gen, want)
        assert gen["counters"]["referenceBinEvaluations"] == want["counters"]["binEvaluations"]
    return got, want


def test_config1_5000_pods_50_types(oracle):
    got, _ = check(oracle, fx.config1())
    assert got["scheduledPods"] == 5000


@pytest.mark.parametrize("pods,types,seed", [(20000, 500, 42), (6000, 144, 3), (2000, 1000, 9)])
def test_config2_scaled(oracle, pods, types, seed):
    got, _ = check(oracle, fx.config2(pods=pods, n_types=types, seed=seed))
    assert got["scheduledPods"] == pods


def test_reference_known_answers(oracle):
    its = fx.fake_default_instance_types()
    pods = [fx.pod(requests={"memory": "1.8G"}, node_selector=AMD) for _ in range(40)] + [fx.pod(requests={"memory": "400M"}, node_selector=AMD) for _ in range(20)]
    got, _ = check(oracle, fx.problem(its, [fx.node_pool()], pods))
    assert len(got["newNodeClaims"]) == 20                                            # suite_test.go:1613-1644
    got, _ = check(oracle, fx.problem(its, [fx.node_pool()], [fx.pod(requests={"cpu": "1m", "memory": "1m"}, node_selector=AMD) for _ in range(25)]))
    assert len(got["newNodeClaims"]) == 5                                             # suite_test.go:1694-1715


def test_edge_cases(oracle):
    its = fx.fake_default_instance_types()
    check(oracle, fx.problem(its, [fx.node_pool()], []))
    pods = [fx.pod(requests={"memory": "2Ti"}), fx.pod(requests={"cpu": "1"}), fx.pod(node_selector={fx.ZONE: "nowhere"}),
            fx.pod(node_requirements=[fx.req("undefined-key", "In", "x")]), fx.pod(requests={"cpu": "100"}), fx.pod(requests={"cpu": "2"})]
    got, _ = check(oracle, fx.problem(its, [fx.node_pool()], pods))
    assert len(got["podErrors"]) == 4
    pools = [fx.node_pool("tainted", weight=10, taints=[{"key": "dedicated", "value": "x", "effect": "NoSchedule"}]), fx.node_pool("plain", weight=1, limits={"cpu": "12"})]
    pods = [fx.pod(requests={"cpu": "1"}, tolerations=[{"key": "dedicated", "operator": "Exists"}] if i % 3 == 0 else None) for i in range(40)]
    check(oracle, fx.problem(its, pools, pods))
    its8 = fx.fake_instance_types(8)
    for expr in (fx.req(fx.FAKE_INTEGER_LABEL, "Gt", "6"), fx.req(fx.FAKE_INTEGER_LABEL, "Lt", "3"), fx.req(fx.FAKE_EXOTIC_LABEL, "DoesNotExist")):
        check(oracle, fx.problem(its8, [fx.node_pool()], [fx.pod(node_requirements=[expr]), fx.pod(requests={"cpu": "1"})]))
    pods = [fx.pod(node_requirements=[fx.req(fx.ZONE, "In", "test-zone-3")], node_preferences=[fx.req(fx.ZONE, "In", "invalid")]),
            fx.pod(node_requirements=[[fx.req(fx.ZONE, "In", "invalid")], [fx.req(fx.ZONE, "In", "test-zone-2")]])]
    check(oracle, fx.problem(its, [fx.node_pool()], pods))


def test_existing_nodes(oracle):
    import random
    rng = random.Random(5)
    its = fx.kwok_catalog(144)
    nodes = []
    for i in range(300):
        it = rng.choice(its)
        nodes.append(fx.state_node(f"node-{i:04d}", it, rng.choice(fx.KWOK_ZONES), rng.choice(["spot", "on-demand"]), "default",
                                   used={"cpu": f"{rng.choice([0, 500, 1500])}m"}, initialized=rng.random() < 0.9, under_consolidate_after=rng.random() < 0.2))
    pods = [fx.pod(requests={"cpu": f"{rng.choice([100, 500, 2000])}m", "memory": f"{rng.choice([256, 2048])}Mi"},
                   node_selector={fx.ZONE: rng.choice(fx.KWOK_ZONES)} if rng.random() < 0.3 else {}, phase=rng.choice(["Pending", "Running"])) for _ in range(3000)]
    np_ = fx.node_pool("default")
    np_["nodeClassLabelKey"] = "karpenter.kwok.sh/kwoknodeclass"
    got, _ = check(oracle, fx.problem(its, [np_], pods, well_known=fx.KWOK_WELL_KNOWN, state_nodes=nodes, options={"consolidationSimulation": True}))
    assert sum(len(e["pods"]) for e in got["existingNodes"]) > 100


def test_full_size_properties():
    """BASELINE configs[1] scale (the oracle cannot run this in seconds): properties that hold for any correct packing."""
    n = int(os.environ.get("KSOLVE_FULL_PODS", "200000"))
    prob = fx.config2(pods=n)
    s = NewScheduler(prob)
    a = s.Solve()
    b = s.Solve()  # idempotence: a second Solve on the same resident inputs gives the same packing
    assert a["scheduledPods"] == n and not a["podErrors"]
    assert a["counters"]["claims"] == b["counters"]["claims"] and a["packingCost"] == b["packingCost"]
    seen = collections.Counter()
    catalog = {t["name"]: t for t in prob["instanceTypes"]}
    for c, c2 in zip(a["newNodeClaims"], b["newNodeClaims"]):
        assert c["pods"] == c2["pods"] and c["instanceTypes"] == c2["instanceTypes"]
        seen.update(c["pods"])
        assert c["instanceTypes"], "a claim without a surviving instance type"
        cpu = int(c["requests"]["cpu"]); pods = int(c["requests"]["pods"]) // 10**9
        assert pods == len(c["pods"])
        # some surviving type must hold the claim's total (kwok allocatable = capacity - 100m cpu)
        assert any(cpu <= int(catalog[t]["capacity"]["cpu"]) * 10**9 - 10**8 and pods <= int(catalog[t]["capacity"]["pods"]) for t in c["instanceTypes"])
    assert len(seen) == n and set(seen.values()) == {1}  # every pod placed exactly once
    # Results keep the reference's final s.newNodeClaims order: sorted by pod count (scheduler.go:598) except for the one
    # claim the last pod touched (no sort runs after the final commit)
    counts = [len(c["pods"]) for c in a["newNodeClaims"]]
    drops = [i for i in range(1, len(counts)) if counts[i] < counts[i - 1]]
    assert len(drops) <= 1


def test_consolidation_sweep(oracle):
    """SURVEY §8 a20 on the GPU: every probe of a single-node consolidation sweep (independent Solve() calls, run on four
    concurrent device sessions) gives the oracle's decision and the oracle's Results."""
    from karpenter_amd import disruption as dz
    cluster = dz.make_cluster(n_nodes=80, pods_per_node=6, seed=9)
    cands = dz.sort_candidates(cluster, cluster["nodes"])[:24]
    got = dz.sweep(cluster, cands, lambda p: NewScheduler(p).Solve(), workers=4)
    want = dz.sweep(cluster, cands, oracle.solve)
    keys = ("decision", "candidates", "replacement", "replacementCapacityType")
    assert [{k: c.get(k) for k in keys} for c in got] == [{k: c.get(k) for k in keys} for c in want]
    for g, w in zip(got, want):
        parity.assert_same_results(g["results"], w["results"])
    a, pa = dz.first_n_consolidation_option(cluster, cands, lambda p: NewScheduler(p).Solve())
    b, pb = dz.first_n_consolidation_option(cluster, cands, oracle.solve)
    assert pa == pb and a["decision"] == b["decision"] and a["candidates"] == b["candidates"]


# ---- topology ---------------------------------------------------------------------------
def test_topology_reference_cases(oracle):
    lab = {"test": "test"}
    base = lambda pods, **kw: fx.problem(fx.fake_default_instance_types(), kw.pop("pools", [fx.node_pool()]), pods, **kw)
    got, _ = check(oracle, base([fx.pod(labels=lab, topology_spread=[fx.spread(fx.ZONE, lab)]) for _ in range(4)]))   # topology_test.go:110-124
    cnt = collections.Counter()
    for c in got["newNodeClaims"]:
        cnt[[q for q in c["requirements"] if q["key"] == fx.ZONE][0]["values"][0]] += len(c["pods"])
    assert sorted(cnt.values()) == [1, 1, 2]
    got, _ = check(oracle, base([fx.pod(labels=lab, topology_spread=[fx.spread(fx.HOSTNAME, lab)]) for _ in range(4)]))  # :547-560
    assert sorted(len(c["pods"]) for c in got["newNodeClaims"]) == [1, 1, 1, 1]
    got, _ = check(oracle, base([fx.pod(labels=lab, pod_anti_requirements=[fx.affinity_term(fx.ZONE, lab)]) for _ in range(5)]))  # :2502-2531
    assert len(got["newNodeClaims"]) == 1 and len(got["podErrors"]) == 4
    aff = {"security": "s2"}
    zp = [fx.pod(requests={"cpu": "2"}, pod_anti_requirements=[fx.affinity_term(fx.ZONE, aff)], node_selector={fx.ZONE: f"test-zone-{i}"}) for i in (1, 2, 3)]
    victim = fx.pod(labels=aff)
    got, _ = check(oracle, base(zp + [victim]))                                                                        # :2466-2500
    assert list(got["podErrors"]) == [victim["uid"]]
    a = {"app": "a"}
    got, _ = check(oracle, base([fx.pod(labels=a, requests={"cpu": "1"}, pod_requirements=[fx.affinity_term(fx.ZONE, a)]) for _ in range(6)]))
    assert not got["podErrors"]
    check(oracle, base([fx.pod(labels=a, pod_preferences=[fx.weighted(10, fx.affinity_term(fx.ZONE, {"app": "nope"}))],
                               pod_anti_preferences=[fx.weighted(5, fx.affinity_term(fx.HOSTNAME, a))]) for _ in range(4)]))


def test_topology_with_existing_nodes_and_cluster_pods(oracle):
    lab = {"test": "test"}
    its = fx.fake_default_instance_types()
    nodes, cluster = [], []
    for i, zone in enumerate(["test-zone-1", "test-zone-1", "test-zone-2"]):
        nodes.append(fx.state_node(f"node-{i}", its[2], zone, "on-demand", "default", used={"cpu": "1", "pods": "1"}))
        cluster.append(fx.pod(labels=lab, phase="Running", node_name=f"node-{i}", requests={"cpu": "1"}))
    cluster.append(fx.pod(labels={"role": "guard"}, phase="Running", node_name="node-2", pod_anti_requirements=[fx.affinity_term(fx.ZONE, {"role": "intruder"})]))
    pods = [fx.pod(labels=lab, topology_spread=[fx.spread(fx.ZONE, lab)]) for _ in range(5)]
    pods += [fx.pod(labels=lab, topology_spread=[fx.spread(fx.HOSTNAME, lab, max_skew=2)]) for _ in range(4)]
    pods += [fx.pod(labels={"role": "intruder"}) for _ in range(2)]
    check(oracle, fx.problem(its, [fx.node_pool()], pods, state_nodes=nodes, cluster_pods=cluster))


@pytest.mark.parametrize("pods,types,anti,seed", [(3000, 144, None, 3), (12000, 500, 700, 42)])
def test_config3_topology_mix(oracle, pods, types, anti, seed):
    """BASELINE configs[2] shape (anti-affinity + 3-zone spread, the reference benchmark's diverse mix) at sizes the
    oracle finishes in seconds."""
    got, _ = check(oracle, fx.config3(pods=pods, n_types=types, seed=seed, anti_affinity_pods=anti))
    assert got["scheduledPods"] == pods


def test_batched_launch_equals_individual_solves(oracle):
    """ksolve_solve_batch: n problems, one pack launch (block b = problem b) — same Results as n ksolve_solve calls."""
    from karpenter_amd.scheduling import SolveBatch
    from karpenter_amd import disruption as dz
    probs = [fx.config2(pods=1500 + 300 * i, n_types=144, seed=50 + i) for i in range(6)]
    probs.append(fx.config3(pods=1200, n_types=144, seed=7))
    probs.append(fx.problem(fx.fake_default_instance_types(), [fx.node_pool()], []))
    got = SolveBatch([NewScheduler(p) for p in probs])
    for g, p in zip(got, probs):
        parity.assert_same_results(g, oracle.solve(p))
    cluster = dz.make_cluster(n_nodes=60, pods_per_node=6, seed=11)
    cands = dz.sort_candidates(cluster, cluster["nodes"])[:32]
    a = dz.sweep_batched(cluster, cands, lambda ps: SolveBatch([NewScheduler(p) for p in ps]))
    b = dz.sweep(cluster, cands, oracle.solve)
    keys = ("decision", "candidates", "replacement", "replacementCapacityType")
    assert [{k: c.get(k) for k in keys} for c in a] == [{k: c.get(k) for k in keys} for c in b]


def test_daemons_min_values_reservations(oracle):
    """SURVEY §8 rows a17 (daemon overhead), a11/a14 (minValues), a18 (reservations) through the HIP path."""
    import random
    from test_device_algorithm import _mv_types, reserved_types, sorted_its
    its = fx.fake_default_instance_types()
    ds = [fx.pod(requests={"cpu": "1", "memory": "1Gi"}), fx.pod(requests={"cpu": "2"}, node_selector={fx.ARCH: "arm64"})]
    pods = [fx.pod(requests={"cpu": f"{c}m"}) for c in (500, 900, 1500, 2500, 3500) for _ in range(4)] + [fx.pod(node_selector={fx.ARCH: "arm64"}, requests={"cpu": "3"})]
    prob = fx.problem(its, [fx.node_pool()], pods, daemonset_pods=ds)
    parity.assert_same_results(sorted_its(NewScheduler(prob).Solve()), sorted_its(oracle.solve(prob)))
    two = [fx.pod(requests={"cpu": "0.9", "memory": "0.9Gi"}) for _ in range(2)]
    for mv, policy in ((2, "Strict"), (3, "Strict"), (3, "BestEffort")):
        pool = fx.node_pool(requirements=[fx.req(fx.INSTANCE_TYPE, "In", "instance-type-1", "instance-type-2", min_values=mv)])
        check(oracle, fx.problem(_mv_types(), [pool], two, options={"minValuesPolicy": policy}))
    kw = fx.kwok_catalog(144)
    np_ = fx.node_pool(requirements=[fx.req("karpenter.kwok.sh/instance-family", "Exists", min_values=3), fx.req(fx.INSTANCE_TYPE, "Exists", min_values=10)])
    np_["nodeClassLabelKey"] = "karpenter.kwok.sh/kwoknodeclass"
    pods = [fx.pod(requests={"cpu": f"{c}m", "memory": f"{m}Mi"}) for c in (500, 4000, 30000, 120000) for m in (512, 8192, 65536) for _ in range(3)]
    check(oracle, fx.problem(kw, [np_], pods, well_known=fx.KWOK_WELL_KNOWN))
    rng = random.Random(5)
    for mode in ("Strict", "Fallback"):
        pods = [fx.pod(requests={"cpu": f"{rng.choice([300, 900, 1800, 2500])}m"}, node_selector=rng.choice([None, None, {fx.CAPACITY_TYPE: "reserved"}, {fx.ZONE: "test-zone-1"}])) for _ in range(25)]
        got, _ = check(oracle, fx.problem(reserved_types(2) + fx.fake_instance_types(4), [fx.node_pool()], pods, options={"reservedCapacity": True, "reservedOfferingMode": mode}))
        assert any(c["reservedOfferings"] for c in got["newNodeClaims"])


def test_big_engine_claim_order_in_hbm(oracle):
    """A solve that needs more in-flight claims than the LDS-resident order holds is re-run on the BIG engine (order in
    HBM). Parity at a size the oracle finishes (cap lowered with ldsClaimCap), then the real thing without the oracle:
    25k anti-affinity pods = 25k NodeClaims, checked through the properties the constraint implies."""
    its = fx.fake_default_instance_types()
    lab = {"app": "nginx"}
    pods = [fx.pod(labels=lab, requests={"cpu": "100m"}, pod_anti_requirements=[fx.affinity_term(fx.HOSTNAME, lab)]) for _ in range(700)]
    pods += [fx.pod(requests={"cpu": "1"}) for _ in range(200)] + [fx.pod(labels={"h": "s"}, topology_spread=[fx.spread(fx.HOSTNAME, {"h": "s"}, max_skew=2)]) for _ in range(90)]
    got, _ = check(oracle, fx.problem(its, [fx.node_pool()], pods, options={"ldsClaimCap": 128}))
    assert len(got["newNodeClaims"]) > 700
    prob = fx.config2(pods=60000, n_types=144, seed=11)
    prob["options"]["ldsClaimCap"] = 64
    check(oracle, prob)
    big = fx.config3(pods=60000, n_types=144, seed=2, anti_affinity_pods=25000)
    r = NewScheduler(big).Solve()
    assert r["scheduledPods"] == 60000 and not r["podErrors"]
    assert len(r["newNodeClaims"]) >= 25000
    anti = 0
    for c in r["newNodeClaims"]:
        assert len(c["pods"]) >= 1
    # every anti-affinity pod sits alone among its kind: count claims is at least the number of such pods
    assert r["counters"]["claims"] == len(r["newNodeClaims"])


def test_truncate_instance_types_order_by_price(oracle):
    """Results.TruncateInstanceTypes in the finalize kernel: OrderByPrice with Go's unstable sort, cap, minValues after the cap."""
    np_ = fx.node_pool()
    np_["nodeClassLabelKey"] = "karpenter.kwok.sh/kwoknodeclass"
    pods = [fx.pod(requests={"cpu": f"{c}m", "memory": f"{m}Mi"}, node_selector=sel) for c in (100, 1500, 9000) for m in (100, 4096)
            for sel in (None, {fx.ZONE: "test-zone-b"}, {fx.CAPACITY_TYPE: "on-demand"}, {fx.ARCH: "arm64"})]
    for cap in (600, 5):
        got, _ = check(oracle, fx.problem(fx.kwok_catalog(1000), [np_], pods, well_known=fx.KWOK_WELL_KNOWN, options={"truncateInstanceTypes": cap}))
        assert all(len(c["instanceTypes"]) <= cap for c in got["newNodeClaims"])
    from test_device_algorithm import _mv_types
    two = [fx.pod(requests={"cpu": "0.9", "memory": "0.9Gi"}) for _ in range(2)]
    pool = fx.node_pool(requirements=[fx.req(fx.INSTANCE_TYPE, "In", "instance-type-1", "instance-type-2", min_values=2)])
    got, _ = check(oracle, fx.problem(_mv_types(), [pool], two, options={"truncateInstanceTypes": 1}))
    assert not got["newNodeClaims"] and len(got["podErrors"]) == 2


def test_same_hash_topology_groups_and_hugepages(oracle):
    """Groups created by relaxation under one TopologyGroup.Hash() with different node filters (first creator wins,
    topology.go:162-194), and hugepage capacity carved out of allocatable memory (types.go:281-291)."""
    lab = {"test": "test"}

    def workload(cpu, zones, n):
        terms = [[fx.req("example.com/unknown", "In", "x")], [fx.req(fx.ZONE, "In", *zones)]]
        return [fx.pod(labels=lab, requests={"cpu": cpu}, node_requirements=terms, topology_spread=[fx.spread(fx.ZONE, lab)]) for _ in range(n)]
    for a, b in ((["test-zone-1", "test-zone-2"], ["test-zone-2", "test-zone-3"]), (["test-zone-2", "test-zone-3"], ["test-zone-1", "test-zone-2"])):
        got, _ = check(oracle, fx.problem(fx.fake_default_instance_types(), [fx.node_pool()], workload("1", a, 5) + workload("500m", b, 5)))
        assert got["counters"]["topologyAliasClasses"] == 1
    its = fx.fake_instance_types(6)
    for i, it in enumerate(its):
        it["capacity"]["hugepages-2Mi"] = f"{512 * (i + 1)}Mi"
    pods = [fx.pod(requests={"cpu": "500m", "memory": "1Gi"}) for _ in range(9)] + [fx.pod(requests={"cpu": "250m", "hugepages-2Mi": "1Gi"}) for _ in range(5)]
    check(oracle, fx.problem(its, [fx.node_pool()], pods))


def test_cancel_stops_a_running_solve():
    """ksolve_cancel from another thread (the ctx deadline, scheduler.go:477-480): the flag is written over PCIe while the
    pack kernel runs and must be seen by its system-scope poll; the solve returns a prefix of the full result."""
    import threading
    import time
    n = 1000000
    s = NewScheduler(fx.config2(pods=n))
    out = {}
    th = threading.Thread(target=lambda: out.update(r=s.Solve(want_results=False)))
    t0 = time.time()
    th.start()
    time.sleep(0.3)
    while th.is_alive():
        s.Cancel()
        time.sleep(0.01)
    th.join()
    elapsed = time.time() - t0
    r = out["r"]
    assert r["timedOut"] and 0 <= r["scheduledPods"] < n      # 0 when the cancel lands before the pack loop placed its first block
    assert elapsed < 4.0, f"a cancelled 1M-pod solve took {elapsed:.1f}s: the flag was not seen by the running kernel"
    full = s.Solve(want_results=False)
    assert not full["timedOut"] and full["scheduledPods"] == n


def test_config4_components_and_cluster_sweep(oracle):
    """BASELINE configs[3] shape (16 NodePools, pods pinned to one each): the whole batch in one Solve() and every NodePool
    component through one batched launch are exact against the oracle; configs[4] shape: a batched single-node
    consolidation sweep over a 1000-node cluster gives the oracle's decisions and placements."""
    from karpenter_amd import disruption as dz
    from karpenter_amd.components import split_by_nodepool
    from karpenter_amd.scheduling import SolveBatch
    prob = fx.config4(pods=20000, n_types=144, n_pools=16, seed=2)
    check(oracle, prob)
    parts = split_by_nodepool(prob)
    assert len(parts) == 16
    for g, (_, sub) in zip(SolveBatch([NewScheduler(sub) for _, sub in parts]), parts):
        parity.assert_same_results(g, oracle.solve(sub))
    cluster = dz.make_cluster(n_nodes=1000, pods_per_node=6, seed=7)
    cands = dz.sort_candidates(cluster, cluster["nodes"])[:4]
    got = dz.sweep_batched(cluster, cands, lambda ps: SolveBatch([NewScheduler(p) for p in ps]))
    want = dz.sweep(cluster, cands, oracle.solve)
    for g, w in zip(got, want):
        assert g["decision"] == w["decision"]
        parity.assert_same_results(g["results"], w["results"])


@pytest.mark.parametrize("source,expected", [("ksolve_min.c", "EXAMPLE_OUTPUT"), ("ksolve_nodes_topology.c", "EXAMPLE2_OUTPUT")])
def test_plain_c_example_on_the_device(tmp_path, source, expected):
    """examples/*.c linked against karpenter_amd/libksolve.so: the C ABI from plain C, no Python in the path (the second
    one hands over an existing node and a topology group, the shapes go/ksolve_flatten.go emits)."""
    import subprocess
    import test_abi
    try:
        exe = test_abi.build_example(tmp_path, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "karpenter_amd"), "ksolve", source)
    except (subprocess.CalledProcessError, FileNotFoundError) as e:      # no C toolchain on this box: nothing to say about the solver
        pytest.skip(f"cannot build the C example here: {e}")
    p = subprocess.run([exe], capture_output=True)
    assert p.returncode == 0, p.stderr.decode()
    assert p.stdout.decode().strip() == getattr(test_abi, expected)


def test_balanced_consolidation_scoring_and_validator_on_the_device(oracle):
    """SURVEY §8 f-4 on the GPU: Balanced-consolidation scoring (balanced.go:47-183) gates commands whose simulations are
    device solves; the single-node scan, the multi-node binary search (same probe sequence) and the validator replay
    (validation.go:297-357) give what they give with the oracle's solves."""
    from karpenter_amd import disruption as dz
    cluster = dz.make_cluster(n_nodes=40, pods_per_node=5, seed=11)
    cluster["nodePools"][0]["consolidationPolicy"] = dz.BALANCED
    for n in cluster["nodes"][::3]:
        for p in n["pods"]:
            p["priority"] = 1000000000          # eviction cost 10 each: expensive to disrupt
    cands = dz.sort_candidates(cluster, cluster["nodes"])
    ev = dz.BalancedEvaluator(cluster, dz.compute_nodepool_totals(cluster, cands))
    dev = lambda p: NewScheduler(p).Solve()
    keys = ("decision", "candidates", "replacement", "replacementCapacityType")
    a = dz.single_node_consolidation(cluster, cands, dev, ev)
    b = dz.single_node_consolidation(cluster, cands, oracle.solve, ev)
    assert {k: a.get(k) for k in keys} == {k: b.get(k) for k in keys}
    assert a["decision"] != dz.NOOP and a["scores"]["default"].approved()
    assert a["scores"]["default"].score() == b["scores"]["default"].score()      # float arithmetic on identical inputs
    ma, pa = dz.first_n_consolidation_option(cluster, cands, dev, evaluator=ev)
    mb, pb = dz.first_n_consolidation_option(cluster, cands, oracle.solve, evaluator=ev)
    assert pa == pb and {k: ma.get(k) for k in keys} == {k: mb.get(k) for k in keys}
    chosen = [c for c in cands if c["name"] in a["candidates"]]
    assert dz.validate_command(cluster, chosen, a, dev) is None
    shrunk = dict(cluster, nodes=[n for n in cluster["nodes"] if n["name"] in a["candidates"]])
    assert dz.validate_command(shrunk, list(shrunk["nodes"]), a, dev) == dz.validate_command(shrunk, list(shrunk["nodes"]), a, oracle.solve)


# ---- parity at the benchmarked sizes ------------------------------------------------------------
# The oracle needs hours at these sizes, so it ran OFFLINE in the CPU container (tests/golden/make_fullsize_digests.py)
# and committed a digest of its canonical Results per configuration; here the device solves the same seeded problem and
# must produce the same digest (L1-strict: claim order, pod identities and slots, instance types, requirements, requests,
# launch price bits) and the same reference bin-evaluation count V.
def _pins():
    import glob
    return sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fullsize", "*.json")))


@pytest.mark.parametrize("pin", _pins(), ids=lambda p: os.path.basename(p)[:-5])
def test_full_size_digest(pin):
    import json
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    from make_fullsize_digests import build_problem
    g = json.load(open(pin))
    if g["pods"] > 2_000_000 and os.environ.get("KSOLVE_TEST_HUGE_PINS") != "1":
        pytest.skip("a pin of this size takes minutes of GPU time (the general engine) or of host time for the digest of its Results: KSOLVE_TEST_HUGE_PINS=1, or tests/tools/gpu_check_pin.py")
    prob = build_problem(g["config"], g["pods"], g["types"], g["seed"], g["extra"])
    # cursor-wide: the cursor engine with the claims' state in HBM (the plan the library moves to when the LDS plan runs out of claims) —
    # every load of a claim another lane stored has to come from L2, not a stale L1 line: the digest of a million placements says so
    # cursor-hbm: the claim order in HBM too (plan 2, above ~15,000 claims: the 10M-pod configs[3] batch as ONE problem runs there)
    cursor_shape = g["config"] in ("config1", "config2", "config4")
    # config3 (BASELINE configs[2], the topology mix): "auto" must be the spread engine (csrc/topo_engine.h); the general / BIG engine is held
    # to the same pins up to 500k pods (16 s of GPU time there, 33 s at 1M)
    engines = ["auto"] + (["general"] if (g["config"] != "config3" and g["pods"] <= 250000) or (g["config"] == "config3" and g["pods"] <= 500000) else []) + (["cursor-wide", "cursor-hbm"] if cursor_shape and g["pods"] <= 1_000_000 else [])
    if cursor_shape and g["pods"] > 1_000_000:
        engines.append("cursor-hbm")   # beyond the LDS plan "auto" IS the plan with the claims' state in HBM (cursor-wide); the order in HBM too is the other one
    for eng in engines:
        s = NewScheduler(dict(prob, options=dict(prob["options"], engine=eng)))
        r = s.Solve()
        s.close()
        digest, fps = parity.results_digest(r)
        assert len(r["newNodeClaims"]) == g["claims"], (eng, len(r["newNodeClaims"]), g["claims"])
        if digest != g["digest"]:
            first = next(i for i, (a, b) in enumerate(zip(fps, g["claimFingerprints"])) if a[:12] != b)
            raise AssertionError(f"{eng}: digest differs from the oracle's; first differing claim {first}")
        assert r["counters"]["referenceBinEvaluations"] == g["binEvaluations"]
        assert float(r["packingCost"]).hex() == g["packingCost"] or abs(r["packingCost"] - g["packingCostApprox"]) < 1e-9 * g["packingCostApprox"]
        if eng == "auto":
            assert r["counters"]["engine"] == ("spread" if g["config"] == "config3" else "cursor"), r["counters"]
        if eng in ("cursor-wide", "cursor-hbm"):
            assert r["counters"]["engine"] == "cursor" and r["counters"]["cursorMemoryPlan"] == (1 if eng == "cursor-wide" else 2), r["counters"]


def test_the_headline_problem_one_hundred_times():
    """The timed problem of bench.py (BASELINE configs[1], 1M pods x 500 types) solved 100 times on one handle: every run must place
    every pod on the same NodeClaim in the same slot and report the same NodeClaims, and the first run's Results must carry the
    digest of the oracle's 1M pin. (Round-4 review: a kernel whose steps overlap — this round's fast loop issues the next pod's
    reads before the current pod's last write — has to show that nothing depends on timing.)"""
    import hashlib
    import json
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    from make_fullsize_digests import build_problem
    g = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fullsize", "config2_p1000000_t500_s42.json")))
    prob = build_problem(g["config"], g["pods"], g["types"], g["seed"], g["extra"])
    s = NewScheduler(prob)
    full = s.Solve()
    digest, _ = parity.results_digest(full)
    assert digest == g["digest"] and full["counters"]["referenceBinEvaluations"] == g["binEvaluations"]
    seen = set()
    for _ in range(100):
        r = s.Solve(want_results="claims")
        assign, slot = s.Assignment()
        h = hashlib.sha256()
        h.update(assign.tobytes()); h.update(slot.tobytes())
        h.update(json.dumps([[c["nodePool"], c["instanceTypes"], c["requirements"], c["requests"], c["cheapestPrice"]] for c in r["newNodeClaims"]], sort_keys=True).encode())
        h.update(str(r["counters"]["referenceBinEvaluations"]).encode())
        seen.add(h.hexdigest())
    s.close()
    assert len(seen) == 1, f"{len(seen)} different answers in 100 solves of one problem"


@pytest.mark.parametrize("leg", ["single", "single-topology", "multi"])
def test_consolidation_sweeps_against_the_population_pins(leg):
    """The probes the oracle simulated AND judged offline for the committed sweep pins (tests/golden/sweeps/: 1,000 stratified
    single-node probes of the 100k-node bench cluster, plain and with spread constraints on its bound pods, and 320 multi-node
    prefixes — tests/golden/make_sweep_pins.py) swept on the device: decision, replacement instance types, capacity type and
    reference bin evaluations of every one of them in one sha256 (round-4 review: 0.2 % of the swept probes were oracle-judged)."""
    import json
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import make_sweep_pins as msp
    from karpenter_amd import disruption as dz
    path = msp.pin_path(leg, 100000)
    if not os.path.exists(path):
        pytest.skip("no pin for this leg (tests/golden/make_sweep_pins.py)")
    g = json.load(open(path))
    cc = dz.make_resident_cluster(n_nodes=g["nodes"], seed=g["seed"], topology=(leg == "single-topology"))
    rc = dz.ResidentCluster.from_compact(cc)
    full = dz.compact_candidates(cc)
    if leg == "multi":
        K = g["windows"][1]
        sets = [[cc["nodes"][i] for i in full[w * K:w * K + k]] for w, k in g["positions"]]
        cmds = rc.decisions(sets, multi_node=True, library_prices=True)
    else:
        n_cand = g["swept_candidates"]
        swept = full[::max(1, len(full) // max(1, n_cand))][:n_cand]
        sets = [[cc["nodes"][swept[j]]] for j in g["positions"]]
        cmds = rc.decisions(sets, library_prices=True, arrays=True)
    ev = rc.last_sweep["referenceBinEvaluations"]
    keys = [msp.probe_key(c["decision"], c["replacement"], c.get("replacementCapacityType"), ev[j]) for j, c in enumerate(cmds)]
    rc.close()
    bad = [pos for pos, a, b in zip(g["positions"], keys, g["keys"]) if a != b]
    assert not bad, f"{len(bad)} of {len(keys)} probes differ from the oracle's pin, first at {bad[:3]}"
    assert msp.digest_of(keys) == g["digest"]


def test_offering_override_groups_on_the_device(oracle):
    """Offering capacity / overhead override groups (types.go:202-269, nodeclaim.go:624-638) on the GPU: the reference's
    two known answers (suite_test.go:5524-5607), the group semantics and the seeded fuzz of tests/test_device_algorithm.py,
    with libksolve.so instead of the emulation (solver_lib=None = the product library)."""
    import test_device_algorithm as tda
    tda.test_offering_override_groups(oracle, None)
    tda.test_offering_override_groups_fuzz(oracle, None)


def test_host_ports_on_the_device(oracle):
    """HostPortUsage in NodeClaim.CanAdd / ExistingNode.CanAdd (hostportusage.go:39-117) on the GPU: the reference's three
    known answers, the matching rule, existing nodes over two passes and the seeded fuzz of tests/test_device_algorithm.py."""
    import test_device_algorithm as tda
    tda.test_host_ports(oracle, None)
    tda.test_host_ports_fuzz(oracle, None)


def test_cross_feature_fuzz_on_the_device(oracle):
    """tests/test_device_fuzz_all.py's cross-feature problems (topology, daemon groups, minValues, reservations, limits,
    relaxation, existing nodes) through libksolve.so: thirty as they are, thirty with volume requirement alternatives."""
    import test_device_fuzz_all as tdf
    solved = 0
    for seed in range(30):
        solved += tdf.run(oracle, None, seed)[0] != "unsupported"
        solved += tdf.run(oracle, None, seed, volumes=True)[0] != "unsupported"
    assert solved >= 50


@pytest.mark.parametrize("at", [5 * 1024, 150 * 1024])
def test_cancel_at_a_poll_boundary_on_the_device(monkeypatch, at):
    """A cancellation that lands at a poll boundary leaves exactly the full run stopped there (claim order included), on both
    engines: tests/test_device_algorithm.py::test_cancel_at_a_poll_boundary_is_the_full_run_stopped_there with libksolve.so."""
    import test_device_algorithm as tda
    # KSOLVE_TEST_CANCEL_AT exists only in the gfx950 build with -DKSOLVE_TEST_HOOKS (the product binary reads no test switch)
    tda.test_cancel_at_a_poll_boundary_is_the_full_run_stopped_there(parity.build_hooks(), monkeypatch, at)


def test_volume_requirement_alternatives_and_complement_min_values_on_the_device(oracle):
    """PodData.VolumeRequirements in NodeClaim.CanAdd / ExistingNode.CanAdd (nodeclaim.go:138-242, existingnode.go:108-168):
    the known answers of provisioning/suite_test.go "Volume Topology Requirements", the order of the alternatives, late
    failures, topology and the seeded fuzz of tests/test_device_algorithm.py on the GPU; and minValues over instance types
    that constrain the key with NotIn / Exists (types.go:399-433)."""
    import test_device_algorithm as tda
    tda.test_volume_requirement_alternatives(oracle, None)
    tda.test_volume_requirement_alternatives_fuzz(oracle, None)
    tda.test_min_values_on_a_key_instance_types_constrain_with_notin_exists(oracle, None)


@pytest.mark.parametrize("seed,limits", [(1, None), (3, {"cpu": "150", "nodes": "31"}), (4, "volumes")])
def test_resident_cluster_probes_on_the_device(oracle, seed, limits):
    """ksolve_probe_create on the GPU: one ksolve_create for the cluster, a removed-node bitmap + displaced-pod rows per
    simulation, all probes in one ksolve_solve_batch launch — against SimulateScheduling assembled from scratch and solved
    by the oracle (tests/test_disruption.py::test_resident_cluster_probes_match_per_probe_rebuild with libksolve.so)."""
    import test_disruption as td
    td.test_resident_cluster_probes_match_per_probe_rebuild(oracle, None, seed, limits)


def test_row_hash_collisions_are_reported_on_the_device(oracle, monkeypatch):
    """The classing kernel (ksolve_row_hash_coop2) with the row hash narrowed to three bits: distinct rows share a hash, inside
    one wavefront (follower against its leader, LDS to LDS) and across wavefronts (leader against the slot's representative,
    row_diff_far); every such pair has to be reported — the host re-seeds and gives up — never merged into one class."""
    hooks = parity.build_hooks()       # the product binary has no test switches: this is the gfx950 build with -DKSOLVE_TEST_HOOKS
    prob = fx.config2(pods=6000, n_types=144, seed=3)
    monkeypatch.setenv("KSOLVE_TEST_HASH_KEEP", "0x7")
    NewScheduler(prob).Solve()          # ... and the product ignores the switch
    with pytest.raises(RuntimeError, match="row hash collisions persist"):
        NewScheduler(prob, solver_lib=hooks).Solve()
    monkeypatch.setenv("KSOLVE_TEST_HASH_KEEP", "0x1")
    same = fx.problem(fx.fake_default_instance_types(), [fx.node_pool()], [fx.pod(requests={"cpu": "1"}, node_selector=AMD) for _ in range(300)])
    check(oracle, same, solver_lib=hooks)
    monkeypatch.delenv("KSOLVE_TEST_HASH_KEEP")
    for kernel in ("coop1", "plain"):    # the previous kernels (A/B switch of the launcher) class the rows alike
        monkeypatch.setenv("KSOLVE_ROWHASH_KERNEL", kernel)
        check(oracle, prob, solver_lib=hooks)


def test_volume_usage_limits_on_the_device(oracle):
    """VolumeUsage.ExceedsLimits / Add on existing nodes (volumeusage.go:193-209, existingnode.go:88, :179) on the GPU: the
    reference's known answers, the seeded fuzz, and probes of a resident cluster with CSI attach limits."""
    import test_reference_known_answers as tk
    import test_disruption as td
    tk.test_volume_usage_limits_on_existing_nodes(oracle, None)
    tk.test_volume_usage_limits_fuzz(oracle, None)
    td.test_resident_cluster_probes_with_volume_limits(oracle, None)


@pytest.mark.parametrize("seed", [11, 13])
def test_resident_cluster_probes_with_topology_on_the_device(oracle, seed):
    """Probes of a resident cluster whose pods carry spread / affinity / anti-affinity constraints (the device counts the
    cluster once, every probe takes its candidates' share out): tests/test_disruption.py's comparison with libksolve.so."""
    import test_disruption as td
    td.test_resident_cluster_probes_with_topology(oracle, None, seed)


def test_consolidation_sweep_over_a_10k_node_cluster_with_topology_pods(oracle, monkeypatch):
    """BASELINE configs[4] shape at a tenth of its size: a resident cluster of 10k nodes / ~195k bound pods, two fifths of the
    default pool's pod templates with zonal or hostname spread constraints, 1000 single-node candidates through ksolve_sweep —
    in several launches (the arena budget is lowered so that the chunked path runs too). Sampled probes of every verdict are
    re-simulated by the oracle (a fresh Scheduler over the cluster without the candidate, the other bound pods as cluster
    pods): decision, replacement and the reference-equivalent evaluation count must be identical."""
    import random
    from collections import Counter
    from karpenter_amd import disruption as dz
    monkeypatch.setenv("KSOLVE_SWEEP_ARENA_MB", "256")
    cc = dz.make_resident_cluster(n_nodes=10_000, seed=7, topology=True)
    rc = dz.ResidentCluster.from_compact(cc)
    order = dz.compact_candidates(cc)
    order = order[::len(order) // 1000][:1000]
    cmds = rc.decisions([[cc["nodes"][i]] for i in order])
    verdicts = Counter(c["decision"] for c in cmds)
    assert set(verdicts) == {dz.DELETE, dz.REPLACE, dz.NOOP}, verdicts
    rng = random.Random(3)
    sample = []
    for d in sorted(verdicts):
        sample += rng.sample([j for j, c in enumerate(cmds) if c["decision"] == d], 3)
    base = dz.compact_problem(cc, pod_groups=[])
    base["clusterPods"] = dz.compact_cluster_pods(cc)
    probes = [{"removeNodes": [cc["nodes"][order[j]]["name"]], "pods": dz.compact_node_pods(cc, order[j])} for j in sample]
    res = oracle.sweep(base, probes, threads=min(len(probes), os.cpu_count() or 1), verdicts=True)
    for j, r, pr in zip(sample, res, probes):
        # the oracle's own verdict (oracle/consolidation.hpp), not karpenter_amd.disruption's
        assert (cmds[j]["decision"], cmds[j]["replacement"], cmds[j].get("replacementCapacityType")) == oracle.verdict_key(r["verdict"]), (j, cmds[j], r["verdict"])
        assert rc.last_sweep["referenceBinEvaluations"][j] == r["counters"]["binEvaluations"]
    rc.close()


@pytest.mark.parametrize("topology", [False, True], ids=["plain", "topology"])
def test_multi_node_consolidation_windows_on_a_10k_node_cluster(oracle, monkeypatch, topology):
    """The multi-node half of BASELINE configs[4] at a tenth of its size: a resident cluster of 10k nodes, four windows of 41
    candidates in sortCandidates' order, every prefix of 2..41 nodes a probe of ONE ksolve_sweep (chunked: the arena budget is
    lowered), verdicts incl. filterOutSameInstanceType from the host library, firstNConsolidationOption's binary search
    (multinodeconsolidation.go:117-207) as a walk over them. Prefixes of several sizes are re-simulated by the oracle — with
    topology, over a cluster whose bound pods carry spread constraints (the probe takes up to 41 nodes' pods out of the counts)."""
    from karpenter_amd import disruption as dz
    monkeypatch.setenv("KSOLVE_SWEEP_ARENA_MB", "512")
    cc = dz.make_resident_cluster(n_nodes=10_000, seed=11, topology=topology)
    rc = dz.ResidentCluster.from_compact(cc)
    full, K = dz.compact_candidates(cc), 41
    sets, key = [], []
    for w in range(4):
        win = [cc["nodes"][i] for i in full[w * K * 50:w * K * 50 + K]]          # windows from different stretches of the order
        for k in range(2, K + 1):
            sets.append(win[:k]); key.append((w, k))
    cmds = rc.decisions(sets, multi_node=True, library_prices=True)
    refs = rc.last_sweep["referenceBinEvaluations"]
    assert [(c["decision"], c["replacement"]) for c in rc.decisions(sets[:40], multi_node=True)] == [(c["decision"], c["replacement"]) for c in cmds[:40]]   # prices summed by the caller
    by = dict(zip(key, cmds))
    for w in range(4):
        cmd, probes = dz.first_n_from_commands(K, lambda k: by[(w, k)], K - 1)
        assert 4 <= len(probes) <= 6 and (cmd["decision"] == dz.NOOP or len(cmd["candidates"]) >= 2)
    picks = [(0, 2), (1, 9), (2, 20), (3, 41), (0, 41)]
    base = dz.compact_problem(cc, pod_groups=[])
    if topology:
        base["clusterPods"] = dz.compact_cluster_pods(cc)
    probes, cand_sets = [], []
    for w, k in picks:
        idx = full[w * K * 50:w * K * 50 + k]
        pods = [dz.compact_node_pods(cc, i) for i in idx]
        probes.append({"removeNodes": [cc["nodes"][i]["name"] for i in idx], "pods": [p for ps in pods for p in ps]})
        cand_sets.append([dict(cc["nodes"][i], pods=ps) for i, ps in zip(idx, pods)])
    res = oracle.sweep(base, probes, threads=min(len(probes), os.cpu_count() or 1), verdicts=True, multi_node=True)
    for kk, r, cs in zip(picks, res, cand_sets):
        # computeConsolidation + filterOutSameInstanceType as restated in oracle/consolidation.hpp
        assert (by[kk]["decision"], by[kk]["replacement"], by[kk].get("replacementCapacityType")) == oracle.verdict_key(r["verdict"]), (kk, by[kk], r["verdict"])
        assert refs[key.index(kk)] == r["counters"]["binEvaluations"]
    rc.close()


def test_cursor_engine_moves_its_claim_state_to_hbm_on_the_device(oracle):
    """tests/test_cursor_engine.py::test_claim_state_in_hbm_when_the_lds_plan_runs_out_of_claims with libksolve.so: more in-flight
    NodeClaims than the cursor engine's LDS plan holds -> the same engine with the claims' state in HBM, not the general engine."""
    import test_cursor_engine as tce
    tce.test_claim_state_in_hbm_when_the_lds_plan_runs_out_of_claims(oracle, None)


def test_cursor_engine_moves_its_claim_order_to_hbm_on_the_device(oracle, monkeypatch):
    """tests/test_cursor_engine.py::test_claim_order_in_hbm_above_the_wide_plan on the GPU (the -DKSOLVE_TEST_HOOKS build of the device
    library: it reads KSOLVE_TEST_WIDE_CAP, the product does not): plan 2 reached straight from the LDS plan and step by step."""
    import test_cursor_engine as tce
    tce.test_claim_order_in_hbm_above_the_wide_plan(oracle, parity.build_hooks(), monkeypatch)


def test_two_wavefront_cursor_kernel_on_the_device(oracle):
    """ksolve_pack_fast2 (engine "cursor-pair": a placer wavefront and a refresher wavefront behind an LDS mailbox, fast_engine.h
    FastMail) against the oracle and against the one-wavefront kernel `auto` runs: on the device the refresher serves a request at
    whatever moment the hardware gives it — anywhere between the two extremes the emulation runs (tests/test_cursor_engine.py) — and
    the Results must not depend on it: several problem sizes, each solved five times on one handle."""
    import random
    import test_cursor_engine as tce
    for seed, pods in ((1, 900), (2, 6000), (3, 40000)):
        prob = tce.lite_problem(random.Random(8100 + seed), pods)
        try:
            s = NewScheduler(tce.with_engine(prob, "cursor-pair"))
            pair = s.Solve()
        except tce.Unsupported:
            continue
        want = oracle.solve(prob)
        parity.assert_same_results(pair, want)
        assert pair["counters"]["engine"] == "cursor" and pair["counters"]["cursorMemoryPlan"] == 0
        assert pair["counters"]["referenceBinEvaluations"] == want["counters"]["binEvaluations"]
        first = parity.results_digest(pair)[0]
        for _ in range(4):
            assert parity.results_digest(s.Solve())[0] == first
        s.close()
        solo = NewScheduler(tce.with_engine(prob, "cursor")).Solve()
        assert parity.results_digest(solo)[0] == first
