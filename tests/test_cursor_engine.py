"""CPU tests of the CURSOR ENGINE (karpenter_amd/csrc/fast_engine.h) through the host emulation of the device code
(tests/emu, test infrastructure only), the real C ABI and the real host flattener: every problem is solved with
options.engine = "cursor" (the cursor engine or an error — never the fallback) and compared with the oracle claim by claim
(L1-strict), and with the general engine. Shapes the cursor engine declines must be declined LOUDLY under engine="cursor"
and solved — identically to the oracle — by the automatic fallback. The GPU run of the same comparisons is
tests/test_gpu_parity.py."""
import copy
import random

import pytest

import parity
from karpenter_amd import fixtures as fx
from karpenter_amd.scheduling import NewScheduler, Unsupported
from test_device_algorithm import emu  # noqa: F401  (fixture)


def with_engine(prob, engine, **opts):
    return dict(prob, options=dict(prob["options"], engine=engine, **opts))


def check_cursor(oracle, emu, prob, want=None):
    """engine=cursor result == oracle; returns the result."""
    want = want or oracle.solve(prob)
    got = NewScheduler(with_engine(prob, "cursor"), solver_lib=emu).Solve()
    assert got["counters"]["engine"] == "cursor"
    parity.assert_same_results(got, want)
    assert got["counters"]["referenceBinEvaluations"] == want["counters"]["binEvaluations"]
    assert abs(got["packingCost"] - want["packingCost"]) < 1e-9 * max(1.0, want["packingCost"])
    # the same engine as the placer + refresher pair of ksolve_pack_fast2 (engine="cursor-pair": the LDS plan with one row of class
    # slots on two wavefronts; the emulation runs the refresher as late as the protocol allows), and with its claim state / claim
    # order in HBM (fast_engine.h FastMem)
    for plan, engine in ((0, "cursor-pair"), (1, "cursor-wide"), (2, "cursor-hbm")):
        other = NewScheduler(with_engine(prob, engine), solver_lib=emu).Solve()
        assert other["counters"]["cursorMemoryPlan"] == plan
        parity.assert_same_results(other, want)
        assert other["counters"]["referenceBinEvaluations"] == want["counters"]["binEvaluations"] and other["counters"]["slowSorts"] == got["counters"]["slowSorts"]
    return got


def check_declined(oracle, emu, prob, reason=None):
    """engine=cursor refuses; automatic selection falls back to the general engine and matches the oracle."""
    with pytest.raises(Unsupported, match="cursor engine"):
        NewScheduler(with_engine(prob, "cursor"), solver_lib=emu).Solve()
    got = NewScheduler(prob, solver_lib=emu).Solve()
    assert got["counters"]["engine"] == "general"
    if reason is not None:
        assert got["counters"]["engineFallbackReason"] == reason, got["counters"]["engineFallbackReason"]
    parity.assert_same_results(got, oracle.solve(prob))
    return got


def test_baseline_shapes_run_on_the_cursor_engine(oracle, emu):
    for prob in (fx.config1(), fx.config2(pods=6000, n_types=144, seed=3), fx.config2(pods=20000, n_types=500, seed=9),
                 fx.config4(pods=8000, n_types=1000, n_pools=16, seed=5)):
        got = check_cursor(oracle, emu, prob)
        auto = NewScheduler(prob, solver_lib=emu).Solve()
        assert auto["counters"]["engine"] == "cursor"          # what bench.py measures
        general = NewScheduler(with_engine(prob, "general"), solver_lib=emu).Solve()
        assert general["counters"]["engine"] == "general"
        parity.assert_same_results(got, general)
        assert got["counters"]["slowSorts"] == general["counters"]["slowSorts"]   # pdqsort left its single-move path equally often


def test_the_x16_pin_at_one_million_pods_on_the_emulation(emu):
    """The oracle's offline pin of the exact configs[3] shape (1M pods x 1000 types x 16 NodePools as ONE Solve(): 2,767 NodeClaims, 160
    pod classes live at once — four rows of class slots, clustered by compatibility, rows a claim cannot meet skipped) reproduced by the
    device algorithm without a GPU: digest, NodeClaim count and reference evaluation count, on the plan `auto` picks (claim records
    outside LDS). The GPU tests and bench.py hold the device against the same file."""
    import json
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    from make_fullsize_digests import build_problem
    g = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fullsize", "config4_p1000000_t1000_s42_x16.json")))
    r = NewScheduler(build_problem(g["config"], g["pods"], g["types"], g["seed"], g["extra"]), solver_lib=emu).Solve()
    assert r["counters"]["engine"] == "cursor" and r["counters"]["cursorMemoryPlan"] == 1
    assert len(r["newNodeClaims"]) == g["claims"] and r["counters"]["referenceBinEvaluations"] == g["binEvaluations"]
    assert parity.results_digest(r)[0] == g["digest"]


def lite_problem(rng, n_pods):
    """Random provisioning batches of the shape the cursor engine solves: In selectors on arch / os / zone / capacity type and
    a custom NodePool label, taints + tolerations, weighted pools, instance types whose allocatable vectors do not dominate
    each other (several Pareto-maximal types per requirement set), requests that straddle them."""
    kwok = rng.random() < 0.4
    if kwok:
        its = fx.kwok_catalog(rng.choice([24, 72, 144, 300])); wk = fx.KWOK_WELL_KNOWN; zones = list(fx.KWOK_ZONES)
    else:
        its = copy.deepcopy(fx.fake_instance_types(rng.choice([6, 20, 60]))); wk = fx.FAKE_WELL_KNOWN; zones = ["test-zone-1", "test-zone-2", "test-zone-3"]
        for it in its:   # cpu-heavy and memory-heavy shapes: no single type dominates
            if rng.random() < 0.5:
                it["capacity"]["memory"] = f"{rng.choice([1, 2, 4, 8, 64, 256])}Gi"
            if rng.random() < 0.3:
                it["capacity"]["pods"] = str(rng.choice([3, 8, 30, 110]))
    archs = sorted({v for it in its for r in it["requirements"] if r["key"] == fx.ARCH for v in r["values"]})
    pools, teams = [], []
    for i in range(rng.choice([0, 0, 1, 2])):
        kw = {}
        reqs = []
        if rng.random() < 0.4: reqs.append(fx.req(fx.ZONE, "In", *rng.sample(zones, rng.choice([1, 2, 3]))))
        if rng.random() < 0.3: reqs.append(fx.req(fx.CAPACITY_TYPE, "In", rng.choice(["spot", "on-demand"])))
        if rng.random() < 0.4: kw["taints"] = [{"key": "dedicated", "value": f"t{i}", "effect": "NoSchedule"}]
        if rng.random() < 0.5: kw["labels"] = {"team": f"t{i}"}; teams.append(f"t{i}")
        if rng.random() < 0.3: kw["limits"] = {"cpu": "100000"}     # present, never binding
        pools.append(fx.node_pool(f"pool-{i}", weight=rng.randrange(1, 20), requirements=reqs, **kw))
    pools.append(fx.node_pool("catch-all", weight=0))   # every pod without a team selector can land here
    if kwok:
        for np_ in pools: np_["nodeClassLabelKey"] = "karpenter.kwok.sh/kwoknodeclass"
    sizes = [(c, m) for c in (100, 250, 700, 1500, 3000) for m in (64, 300, 1024, 5000)]
    classes = []
    for _ in range(rng.randrange(1, 40)):
        sel = {}
        if rng.random() < 0.3: sel[fx.ARCH] = rng.choice(archs)
        if rng.random() < 0.3: sel[fx.ZONE] = rng.choice(zones[:2])
        if rng.random() < 0.2: sel[fx.CAPACITY_TYPE] = rng.choice(["spot", "on-demand"])
        if rng.random() < 0.15: sel[fx.OS] = "linux"
        if teams and rng.random() < 0.15: sel["team"] = rng.choice(teams)
        reqs = None
        if fx.ZONE not in sel and rng.random() < 0.2: reqs = [fx.req(fx.ZONE, "In", *rng.sample(zones, 2))]
        tol = [{"key": "dedicated", "operator": "Exists"}] if (rng.random() < 0.4 or "team" in sel) else None
        c, m = rng.choice(sizes)
        classes.append(dict(requests={"cpu": f"{c}m", "memory": f"{m}Mi"}, node_selector=sel or None, node_requirements=reqs, tolerations=tol))
    pods = [fx.pod(**rng.choice(classes)) for _ in range(n_pods)]
    return fx.problem(its, pools, pods, well_known=wk)


@pytest.mark.parametrize("block", range(5))
def test_fuzz_against_the_oracle(oracle, emu, block):
    ran = declined = 0
    for seed in range(block * 30, block * 30 + 30):
        rng = random.Random(1000 + seed)
        prob = lite_problem(rng, rng.choice([30, 200, 900]))
        want = oracle.solve(prob)
        try:
            check_cursor(oracle, emu, prob, want)
            ran += 1
        except Unsupported:
            # an unschedulable pod (a selector no pool / type satisfies): the general engine owns error codes
            got = NewScheduler(prob, solver_lib=emu).Solve()
            assert got["counters"]["engine"] == "general" and got["counters"]["engineFallbackReason"] in (8, 23, 24, 27), got["counters"]["engineFallbackReason"]
            parity.assert_same_results(got, want)
            declined += 1
    assert ran >= 15, (ran, declined)


def test_refresher_wavefront_early_or_late_same_results(oracle, emu, monkeypatch):
    """The two-wavefront kernel's answer must not depend on WHEN the refresher serves a request: the emulation runs it as late as
    the protocol allows by default (a request is served when the placer waits for it: every stale acceptance word the placer can
    ever see, it sees) and right behind the request with KSOLVE_EMU_REFRESHER_EAGER=1; the device is anywhere in between."""
    for seed in range(12):
        rng = random.Random(7000 + seed)
        prob = lite_problem(rng, rng.choice([200, 900, 2500]))
        try:
            monkeypatch.delenv("KSOLVE_EMU_REFRESHER_EAGER", raising=False)
            late = NewScheduler(with_engine(prob, "cursor-pair"), solver_lib=emu).Solve()
        except Unsupported:
            continue
        monkeypatch.setenv("KSOLVE_EMU_REFRESHER_EAGER", "1")
        early = NewScheduler(with_engine(prob, "cursor-pair"), solver_lib=emu).Solve()
        monkeypatch.delenv("KSOLVE_EMU_REFRESHER_EAGER")
        solo = NewScheduler(with_engine(prob, "cursor"), solver_lib=emu).Solve()
        want = oracle.solve(prob)
        for got in (late, early, solo):
            parity.assert_same_results(got, want)
            assert got["counters"]["referenceBinEvaluations"] == want["counters"]["binEvaluations"]


@pytest.mark.parametrize("block", range(4))
def test_group_speculation_against_the_general_engine(emu, block):
    """Group speculation (one CanAdd test for eight queue entries, fast_engine.h) only runs with at most 12 or at least 50
    in-flight claims: mid-size batches with hundreds of claims, many classes that share them, and runs of equal pod counts —
    touched claims, claims that moved past others, rejected re-tests, the block's last entry. Compared with the general
    engine (itself fuzzed against the oracle) claim by claim, including the reference-equivalent evaluation count."""
    ran = 0
    for seed in range(block * 6, block * 6 + 6):
        rng = random.Random(7000 + seed)
        prob = lite_problem(rng, rng.choice([2500, 4000, 6000]))
        try:
            c = NewScheduler(with_engine(prob, "cursor"), solver_lib=emu).Solve()
        except Unsupported:
            continue
        g = NewScheduler(with_engine(prob, "general"), solver_lib=emu).Solve()
        assert c["counters"]["engine"] == "cursor" and g["counters"]["engine"] == "general"
        parity.assert_same_results(c, g)
        assert c["counters"]["referenceBinEvaluations"] == g["counters"]["referenceBinEvaluations"]
        ran += 1
    assert ran >= 2, ran


def test_several_pareto_vectors_per_requirement_set(oracle, emu):
    """A cpu-heavy and a memory-heavy instance type: neither dominates, so "some type still fits" needs both vectors."""
    def it(name, cpu, mem):
        return fx.fake_instance_type(name, resources={"cpu": str(cpu), "memory": f"{mem}Gi", "pods": "200"})
    its = [it("cpu-heavy", 64, 8), it("mem-heavy", 8, 256), it("small", 2, 2), it("mid", 16, 32)]
    pods = [fx.pod(requests={"cpu": "3", "memory": "100Mi"}) for _ in range(40)] + [fx.pod(requests={"cpu": "100m", "memory": "20Gi"}) for _ in range(40)] + \
           [fx.pod(requests={"cpu": "500m", "memory": "1Gi"}) for _ in range(100)]
    got = check_cursor(oracle, emu, fx.problem(its, [fx.node_pool()], pods))
    assert len(got["newNodeClaims"]) >= 2


def test_more_classes_than_slots_evicts_and_stays_exact(oracle, emu):
    """More than 256 pod classes LIVE AT ONCE in one size run (a class's slot is free again after its last pod, so the classes must
    overlap: 700 of them, two pods each): class slots are recycled wholesale (cursors restart, acceptance bits recomputed)."""
    its = fx.kwok_catalog(72)
    pods = []
    for rep in range(2):
        for i in range(700):   # 700 classes of one (cpu, memory) size: they differ in ephemeral-storage and zone; a class's two pods lie 700 queue entries apart
            pods.append(fx.pod(requests={"cpu": "500m", "memory": "128Mi", "ephemeral-storage": f"{i + 1}Mi"}, node_selector={fx.ZONE: fx.KWOK_ZONES[i % 4]}))
    np_ = fx.node_pool()
    np_["nodeClassLabelKey"] = "karpenter.kwok.sh/kwoknodeclass"
    got = check_cursor(oracle, emu, fx.problem(its, [np_], pods, well_known=fx.KWOK_WELL_KNOWN))
    assert got["counters"]["phaseCycles"][22] > 0   # evictions happened
    assert got["scheduledPods"] == 1400


@pytest.mark.parametrize("seed", range(4))
def test_several_rows_of_class_slots_with_classes_that_overlap(oracle, emu, seed):
    """More than 64 pod classes live at once (several rows of class slots) whose selectors and tolerations OVERLAP: classes pinned to a
    team's pool, classes any pool takes, zone / capacity-type / arch selectors in every combination, three pools with and without taints
    — so the rows new_slot builds (a class goes where the classes it could share a NodeClaim with sit) are not separable, and the
    refresh's row skip (no class of the row tolerates the claim's template and keeps a value on every selected key) must fire exactly
    when the row has no bit to compute. One size per run of the queue keeps every class live for the whole run. Against the oracle on
    the three memory plans (check_cursor)."""
    rng = random.Random(9100 + seed)
    its = fx.kwok_catalog(rng.choice([72, 144])); zones = list(fx.KWOK_ZONES)
    archs = sorted({v for it in its for r in it["requirements"] if r["key"] == fx.ARCH for v in r["values"]})
    pools = []
    for i in range(3):
        kw = {"labels": {"team": f"t{i}"}}
        if i < 2: kw["taints"] = [{"key": "dedicated", "value": f"t{i}", "effect": "NoSchedule"}]
        pools.append(fx.node_pool(f"pool-{i}", weight=rng.randrange(1, 20), **kw))
    pools.append(fx.node_pool("catch-all", weight=0))
    for np_ in pools: np_["nodeClassLabelKey"] = "karpenter.kwok.sh/kwoknodeclass"
    classes, seen = [], set()
    while len(classes) < rng.choice([90, 150, 230]):
        sel = {}
        if rng.random() < 0.5: sel["team"] = f"t{rng.randrange(3)}"
        if rng.random() < 0.5: sel[fx.ZONE] = rng.choice(zones)
        if rng.random() < 0.4: sel[fx.CAPACITY_TYPE] = rng.choice(["spot", "on-demand"])
        if rng.random() < 0.4: sel[fx.ARCH] = rng.choice(archs)
        tol = rng.choice([None, [{"key": "dedicated", "operator": "Exists"}], [{"key": "dedicated", "operator": "Equal", "value": "t0", "effect": "NoSchedule"}]])
        if sel.get("team") == "t1" or (sel.get("team") == "t0" and tol is None): tol = [{"key": "dedicated", "operator": "Exists"}]   # (every pod can be scheduled)
        mem = 64 if len(classes) % 10 else 96    # one size for nine classes in ten: they are all live at once for the whole run of that size
        key = (tuple(sorted(sel.items())), str(tol), mem)
        if key in seen: continue
        seen.add(key)
        classes.append(dict(requests={"cpu": "250m", "memory": f"{mem}Mi"}, node_selector=sel or None, tolerations=tol))
    pods = [fx.pod(**rng.choice(classes)) for _ in range(rng.choice([3000, 6000]))]
    prob = fx.problem(its, pools, pods, well_known=fx.KWOK_WELL_KNOWN)
    want = oracle.solve(prob)
    try:
        check_cursor(oracle, emu, prob, want)
    except Unsupported:
        pytest.skip("declined by the cursor engine (an unschedulable pod in this draw)")


def test_long_runs_of_equally_full_claims(oracle, emu):
    """Every pod needs its own claim-sized share: hundreds of claims with the same pod count, so a commit moves a claim past
    more than 64 others (the in-window move gives way to the general one)."""
    its = fx.fake_instance_types(8)   # the largest type holds one 7-cpu pod
    pods = [fx.pod(requests={"cpu": "7"}) for _ in range(400)] + [fx.pod(requests={"cpu": "300m"}) for _ in range(1500)]
    got = check_cursor(oracle, emu, fx.problem(its, [fx.node_pool()], pods))
    assert len(got["newNodeClaims"]) >= 400


def test_declined_shapes_fall_back_loudly(oracle, emu):
    its = fx.fake_instance_types(8)   # four resource dimensions (the default fake catalogue adds two GPU resources: general engine)
    pool = fx.node_pool()
    # an unschedulable pod: error codes and InstanceTypeFilterError diagnostics are the general engine's
    check_declined(oracle, emu, fx.problem(its, [pool], [fx.pod(requests={"cpu": "1"}) for _ in range(20)] + [fx.pod(requests={"memory": "2Ti"})]), reason=27)
    # NotIn on a pod, Exists on a pod, NotIn on the NodePool: not purely positive
    check_declined(oracle, emu, fx.problem(its, [pool], [fx.pod(node_requirements=[fx.req(fx.ZONE, "NotIn", "test-zone-1")]) for _ in range(5)]), reason=4)
    check_declined(oracle, emu, fx.problem(its, [pool], [fx.pod(node_requirements=[fx.req(fx.ZONE, "Exists")]) for _ in range(5)]), reason=4)
    check_declined(oracle, emu, fx.problem(its, [fx.node_pool(requirements=[fx.req(fx.ZONE, "NotIn", "test-zone-2")])], [fx.pod() for _ in range(5)]), reason=3)
    # a pod that selects on the instance type (a 500-value key does not pack into the claim word)
    check_declined(oracle, emu, fx.problem(its, [pool], [fx.pod(node_selector={fx.INSTANCE_TYPE: "fake-it-3"}) for _ in range(5)]), reason=5)
    # NodePool limits that exclude instance types
    check_declined(oracle, emu, fx.problem(its, [fx.node_pool(limits={"cpu": "20"})], [fx.pod(requests={"cpu": "1"}) for _ in range(40)]))
    # more than four resource dimensions (the fake provider's default catalogue carries two GPU resources)
    got = NewScheduler(fx.problem(fx.fake_default_instance_types(), [pool], [fx.pod(requests={"cpu": "1"}) for _ in range(9)]), solver_lib=emu).Solve()
    assert got["counters"]["engine"] == "general"
    # more in-flight claims than the LDS plan holds
    prob = fx.problem(its, [pool], [fx.pod(requests={"cpu": "7"}) for _ in range(200)], options={"ldsClaimCap": 64})
    check_declined(oracle, emu, prob, reason=26)
    # shapes outside `plain`: topology, relaxation rows
    with pytest.raises(Unsupported, match="cursor engine"):
        NewScheduler(with_engine(fx.problem(its, [pool], [fx.pod(labels={"a": "b"}, topology_spread=[fx.spread(fx.ZONE, {"a": "b"})]) for _ in range(4)]), "cursor"), solver_lib=emu).Solve()
    with pytest.raises(Unsupported, match="cursor engine"):
        NewScheduler(with_engine(fx.problem(its, [pool], [fx.pod(node_preferences=[fx.req(fx.ZONE, "In", "test-zone-1")]) for _ in range(4)]), "cursor"), solver_lib=emu).Solve()


def test_a_handle_that_fell_back_stays_on_the_general_engine(oracle, emu):
    its = fx.fake_instance_types(8)
    prob = fx.problem(its, [fx.node_pool()], [fx.pod(requests={"cpu": "1"}) for _ in range(20)] + [fx.pod(requests={"memory": "2Ti"})])
    s = NewScheduler(prob, solver_lib=emu)
    a = s.Solve()
    b = s.Solve()
    assert a["counters"]["engine"] == b["counters"]["engine"] == "general"
    parity.assert_same_results(a, b)
    parity.assert_same_results(a, oracle.solve(prob))


def test_deadline_mid_block_keeps_the_reference_state(oracle, emu):
    """ctx deadline (maxSteps) in the middle of a 64-pod block: partial Results equal the general engine's at the same
    step — including the claim order, whose last move must NOT have been made (the reference sorts at the next add)."""
    prob = fx.config2(pods=3000, n_types=144, seed=21)
    for steps in (1, 63, 64, 65, 777, 2999):
        c = NewScheduler(with_engine(prob, "cursor", maxSteps=steps), solver_lib=emu).Solve()
        g = NewScheduler(with_engine(prob, "general", maxSteps=steps), solver_lib=emu).Solve()
        assert c["counters"]["engine"] == "cursor" and c["timedOut"] and g["timedOut"]
        parity.assert_same_results(c, g)
        assert c["counters"]["referenceBinEvaluations"] == g["counters"]["referenceBinEvaluations"]


def test_repeated_solves_on_one_handle_are_identical(emu):
    s = NewScheduler(with_engine(fx.config2(pods=5000, n_types=144, seed=8), "cursor"), solver_lib=emu)
    a, b = s.Solve(), s.Solve()
    parity.assert_same_results(a, b)
    assert parity.results_digest(a)[0] == parity.results_digest(b)[0]


def test_batched_launch_mixes_engines(oracle, emu):
    """ksolve_solve_batch: problems of the cursor engine's shape run on it (one block each), a problem it hands back (an
    unschedulable pod) and a problem outside its shape (topology) run on the general engine's batched launch — one call,
    every Results document equal to the oracle's."""
    from karpenter_amd.scheduling import SolveBatch
    its = fx.fake_instance_types(8)
    probs = [fx.config2(pods=1500, n_types=60, seed=70 + i) for i in range(3)]
    probs.append(fx.problem(its, [fx.node_pool()], [fx.pod(requests={"cpu": "1"}) for _ in range(20)] + [fx.pod(requests={"memory": "2Ti"})]))
    probs.append(fx.problem(its, [fx.node_pool()], [fx.pod(labels={"a": "b"}, topology_spread=[fx.spread(fx.ZONE, {"a": "b"})]) for _ in range(6)]))
    got = SolveBatch([NewScheduler(p, solver_lib=emu) for p in probs])
    assert [g["counters"]["engine"] for g in got] == ["cursor", "cursor", "cursor", "general", "general"]
    assert got[3]["counters"]["engineFallbackReason"] == 27
    for g, p in zip(got, probs):
        parity.assert_same_results(g, oracle.solve(p))
    with pytest.raises(Unsupported, match="cursor engine"):
        SolveBatch([NewScheduler(with_engine(probs[3], "cursor"), solver_lib=emu), NewScheduler(probs[0], solver_lib=emu)])


def test_batched_launch_with_handles_on_the_hbm_plans(oracle, emu):
    """ksolve_solve_batch with handles whose claims do not live in LDS (the batched kernel is compiled for the LDS plan
    only): engine = cursor-wide / cursor-hbm handles, and a handle an earlier Solve() moved to plan 1 — each runs alone on the
    kernel of its plan inside the same call; a problem that runs out of LDS claim slots in the batched launch moves to plan 1
    instead of the general engine."""
    from karpenter_amd.scheduling import SolveBatch
    probs = [fx.config2(pods=1200, n_types=60, seed=80 + i) for i in range(4)]
    want = [oracle.solve(p) for p in probs]
    scheds = [NewScheduler(probs[0], solver_lib=emu), NewScheduler(with_engine(probs[1], "cursor-wide"), solver_lib=emu),
              NewScheduler(with_engine(probs[2], "cursor-hbm"), solver_lib=emu), NewScheduler(with_engine(probs[3], "cursor"), solver_lib=emu)]
    got = SolveBatch(scheds)
    assert [g["counters"]["cursorMemoryPlan"] for g in got] == [0, 1, 2, 0]
    for g, w in zip(got, want):
        assert g["counters"]["engine"] == "cursor"
        parity.assert_same_results(g, w)
    # more claims than the LDS plan holds: 3,400 pods that each need a node of their own
    its = fx.fake_instance_types(8)
    big = fx.problem(its, [fx.node_pool()], [fx.pod(requests={"cpu": "7"}) for _ in range(3400)] + [fx.pod(requests={"cpu": "300m"}) for _ in range(600)])
    pair = SolveBatch([NewScheduler(big, solver_lib=emu), NewScheduler(probs[0], solver_lib=emu)])
    assert pair[0]["counters"]["engine"] == "cursor" and pair[0]["counters"]["cursorMemoryPlan"] == 1
    parity.assert_same_results(pair[0], oracle.solve(big))
    parity.assert_same_results(pair[1], want[0])
    # ... and the handle stays there: batched again, it runs alone on plan 1
    s = NewScheduler(big, solver_lib=emu)
    first = s.Solve()
    assert first["counters"]["cursorMemoryPlan"] == 1
    again = SolveBatch([s, NewScheduler(probs[1], solver_lib=emu)])
    assert again[0]["counters"]["cursorMemoryPlan"] == 1
    parity.assert_same_results(again[0], first)
    parity.assert_same_results(again[1], want[1])


def test_claim_state_in_hbm_when_the_lds_plan_runs_out_of_claims(oracle, emu):
    """The cursor engine's LDS plan holds ~3,000 in-flight NodeClaims. A batch that needs more stops it with reason 26, and the
    library runs it once more with the claims' state in HBM (ClaimStates<true>: ~15,000 claims, only the order arrays in LDS)
    instead of handing the problem to the general engine. 3,600 pods that each need a node of their own, plus small pods that fill
    the gaps: both plans, the automatic switch and the general engine agree with the oracle claim by claim."""
    big = fx.pod(uid="t", requests={"cpu": "130", "memory": "1Gi"})
    small = fx.pod(uid="t", requests={"cpu": "500m", "memory": "256Mi"}, node_selector={fx.ARCH: "amd64"})
    np_ = fx.node_pool("default")
    np_["nodeClassLabelKey"] = "karpenter.kwok.sh/kwoknodeclass"
    prob = fx.problem(fx.kwok_catalog(144), [np_], pod_groups=[{"count": 3600, "uidSeed": 7, "template": big}, {"count": 5000, "uidSeed": 8, "template": small}], well_known=fx.KWOK_WELL_KNOWN)
    want = oracle.solve(prob)
    assert len(want["newNodeClaims"]) > 3100
    seen = {}
    for engine in ("auto", "cursor-wide", "cursor-hbm", "general"):
        s = NewScheduler(dict(prob, options=dict(prob["options"], engine=engine)), solver_lib=emu)
        got = s.Solve()
        parity.assert_same_results(got, want)
        assert got["counters"]["referenceBinEvaluations"] == want["counters"]["binEvaluations"]
        seen[engine] = (got["counters"]["engine"], got["counters"]["cursorMemoryPlan"], got["counters"]["engineFallbackReason"])
        if engine == "auto":      # the handle remembers: the second Solve() starts with the wide plan
            again = s.Solve()
            parity.assert_same_results(again, want)
        s.close()
    assert seen == {"auto": ("cursor", 1, 0), "cursor-wide": ("cursor", 1, 0), "cursor-hbm": ("cursor", 2, 0), "general": ("general", -1, 0)}, seen


def test_claim_order_in_hbm_above_the_wide_plan(oracle, emu, monkeypatch):
    """Above ~15,000 in-flight NodeClaims the order arrays leave LDS too (plan 2, FastMem<2>: 65,472 claims — the exact configs[3]
    batch of 10M pods ends with 27,345). With plan 1 shrunk by the test switch KSOLVE_TEST_WIDE_CAP the automatic path reaches plan 2
    both ways: straight from the LDS plan when the first attempt's claims-per-pod rate says plan 1 cannot hold the queue either
    (the big pods come first: one claim per pod placed), and step by step (0 -> 1 -> 2) when the rate said it would
    (claims that open late: pods with little cpu — the end of the queue — and 1.5 TiB of memory each; 3,000 of them join the
    big pods' claims, the other 1,500 need their own)."""
    big = fx.pod(uid="t", requests={"cpu": "130", "memory": "1Gi"})
    small = fx.pod(uid="t", requests={"cpu": "500m", "memory": "256Mi"}, node_selector={fx.ARCH: "amd64"})
    late = fx.pod(uid="t", requests={"cpu": "100m", "memory": "1500Gi"})
    np_ = fx.node_pool("default")
    np_["nodeClassLabelKey"] = "karpenter.kwok.sh/kwoknodeclass"
    cases = (("straight", 3200, [{"count": 3600, "uidSeed": 7, "template": big}, {"count": 5000, "uidSeed": 8, "template": small}]),
             ("stepwise", 4160, [{"count": 3000, "uidSeed": 7, "template": big}, {"count": 20000, "uidSeed": 8, "template": small}, {"count": 4500, "uidSeed": 9, "template": late}]))
    for how, wide_cap, groups in cases:
        monkeypatch.setenv("KSOLVE_TEST_WIDE_CAP", str(wide_cap))
        prob = fx.problem(fx.kwok_catalog(144), [np_], pod_groups=groups, well_known=fx.KWOK_WELL_KNOWN)
        want = oracle.solve(prob)
        assert len(want["newNodeClaims"]) > wide_cap, (how, len(want["newNodeClaims"]))
        s = NewScheduler(prob, solver_lib=emu)
        got = s.Solve()
        parity.assert_same_results(got, want)
        assert got["counters"]["referenceBinEvaluations"] == want["counters"]["binEvaluations"]
        assert (got["counters"]["engine"], got["counters"]["cursorMemoryPlan"], got["counters"]["engineFallbackReason"]) == ("cursor", 2, 0), (how, got["counters"])
        # (round 6: NewScheduler bounds the NodeClaims from below — total requests over the largest allocatable — and starts on the first
        # plan that holds the bound: both cases skip the LDS plan; "stepwise" used to take three attempts)
        assert got["counters"]["cursorAttempts"] == 2, (how, got["counters"]["cursorAttempts"])
        s.close()
