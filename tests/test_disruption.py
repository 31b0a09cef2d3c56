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
