"""Cross-feature fuzz of the DEVICE ALGORITHM (host emulation of the product's engine, tests/emu) against the oracle: random
problems that mix NodePool weights / taints / limits / minValues, reservations, daemonsets, existing nodes with running
pods, every topology constraint kind, node selectors / affinities / preferences and the small LDS claim cap that forces
the BIG engine. Features interact (this is how the minValues-lost-through-a-topology-step bug was found)."""
import copy
import random

import pytest

import invariants
import parity
from karpenter_amd import fixtures as fx
from karpenter_amd.scheduling import NewScheduler, Unsupported
from test_device_algorithm import emu  # noqa: F401  (fixture)

ZONES = ["test-zone-1", "test-zone-2", "test-zone-3"]
def run(oracle, emu, seed, volumes=False):
    rng = random.Random(seed)
    its = copy.deepcopy(fx.fake_default_instance_types() if rng.random() < 0.5 else fx.fake_instance_types(rng.choice([4, 8, 16])))
    opts = {}
    if rng.random() < 0.3:
        opts["minValuesPolicy"] = rng.choice(["Strict", "BestEffort"])
    if rng.random() < 0.3:
        opts.update({"reservedCapacity": True, "reservedOfferingMode": rng.choice(["Strict", "Fallback"])})
        for i in range(rng.randrange(1, 3)):
            it = rng.choice(its)
            for r in it["requirements"]:
                if r["key"] == fx.CAPACITY_TYPE and "reserved" not in r["values"]: r["values"].append("reserved")
            it["offerings"].append(fx.offering("reserved", rng.choice(ZONES[:2]), 0.001 * (i + 1), reservation_id=f"cr-{i}", reservation_capacity=rng.randrange(1, 4)))
    if rng.random() < 0.2:
        opts["ldsClaimCap"] = 64
    pools = []
    npools = rng.choice([1, 1, 2, 3])
    for i in range(npools):
        kw = {}
        if rng.random() < 0.3: kw["taints"] = [{"key": "team", "value": f"t{i}", "effect": rng.choice(["NoSchedule", "PreferNoSchedule"])}]
        if rng.random() < 0.3: kw["limits"] = {"cpu": str(rng.choice([8, 30, 100]))}
        reqs = []
        if rng.random() < 0.3: reqs.append(fx.req(fx.ZONE, "In", *rng.sample(ZONES, 2)))
        if rng.random() < 0.15: reqs.append(fx.req(fx.INSTANCE_TYPE, "Exists", min_values=rng.choice([1, 2, 3])))
        if rng.random() < 0.2: kw["labels"] = {"team": f"t{i}"}
        pools.append(fx.node_pool(f"pool-{i}", weight=rng.randrange(0, 50), requirements=reqs, **kw))
    labels = [{"app": c} for c in "abc"]
    def rand_pod(**extra):
        kw = dict(labels=rng.choice(labels), requests={"cpu": f"{rng.choice([100, 250, 500, 1000, 2000, 3500])}m", "memory": f"{rng.choice([64, 256, 1024, 3000])}Mi"})
        r = rng.random()
        sel = rng.choice(labels)
        if r < 0.12: kw["topology_spread"] = [fx.spread(fx.ZONE, sel, max_skew=rng.choice([1, 2]), when=rng.choice(["DoNotSchedule", "ScheduleAnyway"]))]
        elif r < 0.22: kw["topology_spread"] = [fx.spread(fx.HOSTNAME, sel, max_skew=rng.choice([1, 3]))]
        elif r < 0.30: kw["pod_requirements"] = [fx.affinity_term(rng.choice([fx.ZONE, fx.HOSTNAME]), sel)]
        elif r < 0.38: kw["pod_anti_requirements"] = [fx.affinity_term(rng.choice([fx.ZONE, fx.HOSTNAME]), sel)]
        elif r < 0.44: kw["pod_anti_preferences"] = [fx.weighted(rng.randrange(1, 9), fx.affinity_term(fx.HOSTNAME, sel))]
        r = rng.random()
        if r < 0.15: kw["node_selector"] = {fx.ZONE: rng.choice(ZONES)}
        elif r < 0.25: kw["node_requirements"] = [fx.req(fx.ARCH, rng.choice(["In", "NotIn"]), rng.choice(["amd64", "arm64"]))]
        elif r < 0.32: kw["node_preferences"] = [fx.req(fx.ZONE, "In", rng.choice(ZONES))]
        elif r < 0.36: kw["node_requirements"] = [[fx.req(fx.ZONE, "In", "nowhere")], [fx.req(fx.ZONE, "In", rng.choice(ZONES))]]
        if rng.random() < 0.3: kw["tolerations"] = [{"key": "team", "operator": "Exists"}]
        kw.update(extra)
        return fx.pod(**kw)
    pods = [rand_pod() for _ in range(rng.randrange(10, 120))]
    daemons = []
    if rng.random() < 0.4:
        for _ in range(rng.randrange(1, 3)):
            kw = dict(requests={"cpu": f"{rng.choice([50, 200])}m", "memory": "64Mi"})
            if rng.random() < 0.4: kw["node_selector"] = {fx.ARCH: rng.choice(["amd64", "arm64"])}
            if rng.random() < 0.5: kw["tolerations"] = [{"operator": "Exists"}]
            daemons.append(fx.pod(**kw))
    nodes, cluster = [], []
    if rng.random() < 0.4:
        for i in range(rng.randrange(1, 5)):
            it = rng.choice(its)
            n = fx.state_node(f"node-{i}", it, rng.choice(ZONES), "on-demand", pools[0]["name"], used={"cpu": "200m", "pods": "2"}, initialized=rng.random() < 0.8)
            if rng.random() < 0.3: n["taints"] = [{"key": "team", "value": "t0", "effect": "NoSchedule"}]
            nodes.append(n)
            for _ in range(rng.randrange(0, 3)):
                cluster.append(rand_pod(phase="Running", node_name=f"node-{i}"))
    if volumes:
        # volume requirement alternatives (volumeReqsByPod) on a quarter of the pods, from a generator of their own so that the
        # problems of the seeds above stay what they were
        vr = random.Random(seed * 7919 + 13)
        def alt():
            reqs = [fx.req(fx.ZONE, vr.choice(["In", "In", "NotIn"]), *vr.sample(ZONES, vr.choice([1, 1, 2])))] if vr.random() < 0.8 else []
            if vr.random() < 0.3: reqs.append(fx.req(fx.CAPACITY_TYPE, "In", vr.choice(["spot", "on-demand", "reserved"])))
            if vr.random() < 0.15: reqs.append(fx.req(fx.HOSTNAME, vr.choice(["In", "NotIn"]), vr.choice(["node-0", "node-1", "elsewhere"])))
            return reqs or [fx.req(fx.ARCH, "In", vr.choice(["amd64", "arm64"]))]
        for p in pods:
            if vr.random() < 0.25:
                p["volumeRequirements"] = [alt() for _ in range(vr.choice([1, 1, 2, 3]))]
    prob = fx.problem(its, pools, pods, state_nodes=nodes, cluster_pods=cluster, daemonset_pods=daemons, options=opts)
    want = oracle.solve(prob)
    invariants.check(prob, want)
    try:
        got = NewScheduler(prob, solver_lib=emu).Solve()
    except Unsupported as e:
        return ("unsupported", str(e)[:60])
    for r in (want, got):
        for c in r["newNodeClaims"]: c["instanceTypes"] = sorted(c["instanceTypes"])
    parity.assert_same_results(got, want)
    assert got["counters"]["referenceBinEvaluations"] == want["counters"]["binEvaluations"]
    return (len(pods), len(got["newNodeClaims"]), len(got["podErrors"]), got["counters"]["topologyAliasClasses"])


@pytest.mark.parametrize("block", range(6))
def test_cross_feature_fuzz(oracle, emu, block):
    for seed in range(block * 40, block * 40 + 40):
        run(oracle, emu, seed)


@pytest.mark.parametrize("block", range(3))
def test_cross_feature_fuzz_with_volume_alternatives(oracle, emu, block):
    """The same problems with volume requirement alternatives on a quarter of the pods: every later stage of CanAdd —
    topology, daemon overhead groups, minValues (Strict and BestEffort), reservations (Strict and Fallback), NodePool limits,
    relaxation — now runs once per alternative (nodeclaim.go:149-157, existingnode.go:116-139)."""
    solved = 0
    for seed in range(block * 40, block * 40 + 40):
        solved += run(oracle, emu, seed, volumes=True)[0] != "unsupported"
    assert solved >= 35


def run_wide(oracle, emu, seed):
    rng = random.Random(seed)
    kwok = rng.random() < 0.5
    if kwok:
        its = fx.kwok_catalog(rng.choice([24, 72, 144])); wk = fx.KWOK_WELL_KNOWN; ZONES = list(fx.KWOK_ZONES)
    else:
        its = copy.deepcopy(fx.fake_instance_types(rng.choice([6, 20, 60]))); wk = fx.FAKE_WELL_KNOWN; ZONES = ["test-zone-1", "test-zone-2", "test-zone-3"]
    opts = {}
    if rng.random() < 0.15: opts["ldsClaimCap"] = 64
    if rng.random() < 0.2: opts["preferencePolicy"] = "Ignore"
    pools = []
    for i in range(rng.choice([1, 1, 2])):
        kw = {}
        reqs = []
        if rng.random() < 0.3: reqs.append(fx.req(fx.ZONE, rng.choice(["In", "NotIn"]), *rng.sample(ZONES, rng.choice([1, 2]))))
        if kwok and rng.random() < 0.3: reqs.append(fx.req(fx.KWOK_CPU, rng.choice(["Gt", "Lt"]), str(rng.choice([2, 8, 32]))))
        if rng.random() < 0.25: kw["taints"] = [{"key": "dedicated", "value": "x", "effect": "NoSchedule"}]
        if rng.random() < 0.2: kw["limits"] = {"cpu": str(rng.choice([50, 400]))}
        np_ = fx.node_pool(f"pool-{i}", weight=rng.randrange(0, 20), requirements=reqs, **kw)
        if kwok: np_["nodeClassLabelKey"] = "karpenter.kwok.sh/kwoknodeclass"
        pools.append(np_)
    labels = [{"app": c, "tier": rng.choice(["fe", "be"])} for c in "abcd"]
    nss = ["default", "default", "other"]
    pods = []
    n = rng.choice([30, 80, 200, 500])
    for _ in range(n):
        lab = rng.choice(labels); sel = {"app": rng.choice("abcd")}
        kw = dict(labels=lab, namespace=rng.choice(nss), requests={"cpu": f"{rng.choice([100, 250, 500, 1000, 1500, 4000])}m", "memory": f"{rng.choice([128, 512, 2048])}Mi"})
        r = rng.random()
        tsc = []
        if r < 0.2:
            c = fx.spread(rng.choice([fx.ZONE, fx.ZONE, fx.HOSTNAME, fx.CAPACITY_TYPE]), sel, max_skew=rng.choice([1, 1, 2, 5]), when=rng.choice(["DoNotSchedule", "DoNotSchedule", "ScheduleAnyway"]),
                          min_domains=rng.choice([None, None, None, 2, 5]), taints_policy=rng.choice([None, None, "Honor"]), affinity_policy=rng.choice([None, None, "Ignore"]))
            if rng.random() < 0.2: c["matchLabelKeys"] = ["tier"]
            tsc.append(c)
            if rng.random() < 0.3: tsc.append(fx.spread(fx.HOSTNAME, sel, max_skew=rng.choice([2, 4])))
            kw["topology_spread"] = tsc
        elif r < 0.3:
            t = fx.affinity_term(rng.choice([fx.ZONE, fx.HOSTNAME]), sel, namespaces=rng.choice([None, None, ["default", "other"]]))
            kw["pod_requirements"] = [t]
        elif r < 0.4:
            kw["pod_anti_requirements"] = [fx.affinity_term(rng.choice([fx.ZONE, fx.HOSTNAME, fx.HOSTNAME]), sel)]
        elif r < 0.46:
            kw["pod_preferences"] = [fx.weighted(rng.randrange(1, 50), fx.affinity_term(rng.choice([fx.ZONE, fx.HOSTNAME]), sel)) for _ in range(rng.randrange(1, 3))]
        r = rng.random()
        if r < 0.12: kw["node_selector"] = {fx.ZONE: rng.choice(ZONES)}
        elif r < 0.2: kw["node_requirements"] = [fx.req(fx.ZONE, "NotIn", rng.choice(ZONES))]
        elif r < 0.3: kw["node_preferences"] = [{"weight": rng.randrange(1, 9), "matchExpressions": [fx.req(fx.ZONE, "In", rng.choice(ZONES))]} for _ in range(rng.randrange(1, 3))]
        elif r < 0.36 and kwok: kw["node_requirements"] = [fx.req(fx.KWOK_CPU, rng.choice(["Gt", "Lt"]), str(rng.choice([1, 4, 16, 64])))]
        elif r < 0.42: kw["node_requirements"] = [[fx.req(fx.ZONE, "In", rng.choice(ZONES))], [fx.req(fx.CAPACITY_TYPE, "In", "spot")]]
        if rng.random() < 0.3: kw["tolerations"] = [{"key": "dedicated", "operator": "Exists"}]
        pods.append(fx.pod(**kw))
    prob = fx.problem(its, pools, pods, well_known=wk, options=opts)
    want = oracle.solve(prob)
    invariants.check(prob, want)
    try:
        got = NewScheduler(prob, solver_lib=emu).Solve()
    except Unsupported as e:
        return ("unsupported", str(e)[:80])
    parity.assert_same_results(got, want)
    assert got["counters"]["referenceBinEvaluations"] == want["counters"]["binEvaluations"]
    return (len(pods), len(got["newNodeClaims"]), len(got["podErrors"]), got["counters"]["relaxations"], got["counters"]["topologyAliasClasses"])


@pytest.mark.parametrize("block", range(4))
def test_wide_fuzz(oracle, emu, block):
    """KWOK and fake catalogues, Gt/Lt bounds on NodePools and pods, namespaces, matchLabelKeys, minDomains, node taint /
    affinity policies, preferred node affinities, PreferencePolicy=Ignore, up to 500 pods (claim order beyond the
    insertion-sort regime of pdqsort)."""
    for seed in range(block * 10, block * 10 + 10):
        run_wide(oracle, emu, seed)


def run_passes(oracle, emu, seed):
    """Several provisioning passes on a growing cluster (fixtures.launch between them): every pass sees the nodes and the
    bound pods of the passes before it, so countDomains, inverse anti-affinity groups of bound pods, node filters and
    existing-node packing are exercised on states the solver itself produced. Also draws namespaces with labels,
    namespaceSelector terms and hugepage capacity."""
    rng = random.Random(10_000 + seed)
    its = copy.deepcopy(fx.fake_default_instance_types() if rng.random() < 0.6 else fx.fake_instance_types(rng.choice([6, 12])))
    if rng.random() < 0.3:
        for it in rng.sample(its, max(1, len(its) // 3)):
            it["capacity"]["hugepages-2Mi"] = rng.choice(["256Mi", "1Gi"])
    namespaces = [{"name": "default", "labels": {"tier": "a"}}, {"name": "other", "labels": {"tier": "b"}}, {"name": "third", "labels": {}}]
    pools = [fx.node_pool("pool-0")]
    if rng.random() < 0.4:
        pools.append(fx.node_pool("pool-1", weight=10, requirements=[fx.req(fx.ZONE, "In", *rng.sample(ZONES, 2))], taints=[{"key": "team", "value": "x", "effect": "NoSchedule"}]))
    labels = [{"app": c} for c in "ab"]

    def rand_pod():
        kw = dict(labels=rng.choice(labels), namespace=rng.choice(["default", "default", "other"]),
                  requests={"cpu": f"{rng.choice([100, 500, 1000, 2500])}m", "memory": f"{rng.choice([128, 1024, 2048])}Mi"})
        sel = rng.choice(labels)
        r = rng.random()
        if r < 0.15: kw["topology_spread"] = [fx.spread(rng.choice([fx.ZONE, fx.HOSTNAME, fx.CAPACITY_TYPE]), sel, max_skew=rng.choice([1, 2]),
                                                        taints_policy=rng.choice([None, "Honor"]), affinity_policy=rng.choice([None, "Ignore"]))]
        elif r < 0.27:
            nss = rng.choice([None, None, {"matchLabels": {}}, {"matchLabels": {"tier": "b"}}, {"matchLabels": {"tier": "zzz"}}])
            kw["pod_requirements"] = [fx.affinity_term(rng.choice([fx.ZONE, fx.HOSTNAME]), sel, namespaces=rng.choice([None, ["other"]]), namespace_selector=nss)]
        elif r < 0.37:
            nss = rng.choice([None, None, {"matchLabels": {}}, {"matchLabels": {"tier": "a"}}])
            kw["pod_anti_requirements"] = [fx.affinity_term(rng.choice([fx.ZONE, fx.HOSTNAME]), sel, namespace_selector=nss)]
        elif r < 0.42: kw["pod_preferences"] = [fx.weighted(5, fx.affinity_term(fx.ZONE, sel))]
        r = rng.random()
        if r < 0.15: kw["node_selector"] = {fx.ZONE: rng.choice(ZONES)}
        elif r < 0.22: kw["node_requirements"] = [fx.req(fx.CAPACITY_TYPE, "In", rng.choice(["spot", "on-demand"]))]
        if rng.random() < 0.25: kw["tolerations"] = [{"key": "team", "operator": "Exists"}]
        if rng.random() < 0.1: kw["requests"]["hugepages-2Mi"] = "128Mi"
        return fx.pod(**kw)

    nodes, bound, alias = [], [], 0
    for pass_no in range(rng.choice([2, 3, 4])):
        pods = [rand_pod() for _ in range(rng.randrange(3, 30))]
        prob = fx.problem(its, pools, pods, state_nodes=nodes, cluster_pods=bound, namespaces=namespaces)
        want = oracle.solve(prob)
        invariants.check(prob, want)
        try:
            got = NewScheduler(prob, solver_lib=emu).Solve()
        except Unsupported as e:
            return ("unsupported", str(e)[:80])
        parity.assert_same_results(got, want)
        assert got["counters"]["referenceBinEvaluations"] == want["counters"]["binEvaluations"]
        alias += got["counters"]["topologyAliasClasses"]
        nn, bb = fx.launch(want, its, pods, name_prefix=f"s{seed}p{pass_no}")
        nodes += nn
        bound += bb
    return (len(nodes), len(bound), alias)


@pytest.mark.parametrize("block", range(4))
def test_multi_pass_fuzz(oracle, emu, block):
    for seed in range(block * 30, block * 30 + 30):
        run_passes(oracle, emu, seed)


@pytest.fixture(scope="module")
def emu_reversed():
    import __graft_entry__  # noqa: F401
    return parity.build_emu(reverse_lanes=True)


def test_lane_order_does_not_matter(oracle, emu_reversed):
    """On the device the 64 lanes of a wave-wide step run in lockstep, in the emulation one after the other. If a lane's
    work depended on what another lane of the same step wrote, the two would disagree — and so would the emulation with
    itself when the lanes run in the opposite order. Same fuzzers, same oracle, reversed lanes."""
    for seed in range(40):
        run(oracle, emu_reversed, seed)
        run_passes(oracle, emu_reversed, seed)
    for seed in range(8):
        run_wide(oracle, emu_reversed, seed)
    # the BIG engine (claim order in HBM) and the lite engine (no topology)
    prob = fx.config3(pods=1500, n_types=72, seed=5, anti_affinity_pods=200)
    prob["options"]["ldsClaimCap"] = 64
    parity.assert_same_results(NewScheduler(prob, solver_lib=emu_reversed).Solve(), oracle.solve(prob))
    prob = fx.config2(pods=6000, n_types=144, seed=11)
    parity.assert_same_results(NewScheduler(prob, solver_lib=emu_reversed).Solve(), oracle.solve(prob))
