"""The invariant checker used by the fuzzers must itself notice broken packings."""
import copy

import pytest

import invariants
from karpenter_amd import fixtures as fx


def _solved(oracle):
    lab = {"app": "x"}
    pods = [fx.pod(labels=lab, requests={"cpu": "100m"}, pod_anti_requirements=[fx.affinity_term(fx.HOSTNAME, lab)]) for _ in range(2)]
    pods += [fx.pod(requests={"cpu": "500m"}, node_selector={fx.ZONE: "test-zone-2"}), fx.pod(requests={"cpu": "64"})]
    pool = fx.node_pool(taints=[{"key": "team", "value": "a", "effect": "NoSchedule"}])
    for p in pods:
        p["tolerations"] = [{"key": "team", "operator": "Exists", "value": "", "effect": ""}]
    prob = fx.problem(fx.fake_default_instance_types(), [pool], pods)
    return prob, oracle.solve(prob)


def test_valid_result_passes_and_corruptions_are_caught(oracle):
    prob, res = _solved(oracle)
    invariants.check(prob, res)
    assert len(res["podErrors"]) == 1 and len(res["newNodeClaims"]) == 2

    bad = copy.deepcopy(res)                       # both repelling pods on one claim
    a, b = [c for c in bad["newNodeClaims"] if prob["pods"][0]["uid"] in c["pods"] or prob["pods"][1]["uid"] in c["pods"]]
    a["pods"] += b["pods"]; b["pods"] = []
    a["instanceTypes"] = ["default-instance-type"]; a["requirements"] = a["requirements"] + [q for q in b["requirements"] if q["key"] == fx.ZONE and not any(x["key"] == fx.ZONE for x in a["requirements"])]
    with pytest.raises(AssertionError, match="repel"):
        invariants.check(prob, bad)

    bad = copy.deepcopy(res)                       # a pod reported twice
    bad["newNodeClaims"][0]["pods"].append(bad["newNodeClaims"][1]["pods"][0])
    with pytest.raises(AssertionError, match="placed twice"):
        invariants.check(prob, bad)

    bad = copy.deepcopy(res)                       # the pod that fits nowhere "placed" on a small claim
    uid = next(iter(bad["podErrors"]))
    del bad["podErrors"][uid]
    bad["newNodeClaims"][0]["pods"].append(uid)
    with pytest.raises(AssertionError, match="does not fit"):
        invariants.check(prob, bad)

    bad = copy.deepcopy(res)                       # zone selector dropped from the claim
    c = next(c for c in bad["newNodeClaims"] if prob["pods"][2]["uid"] in c["pods"])
    c["requirements"] = [q for q in c["requirements"] if q["key"] != fx.ZONE]
    with pytest.raises(AssertionError, match="nodeSelector"):
        invariants.check(prob, bad)

    noprob = copy.deepcopy(prob)                   # a pod that does not tolerate the pool's taint
    noprob["pods"][2]["tolerations"] = []
    with pytest.raises(AssertionError, match="does not tolerate"):
        invariants.check(noprob, res)

    bad = copy.deepcopy(res)                       # a pod that vanished
    bad["newNodeClaims"][0]["pods"].pop()
    with pytest.raises(AssertionError):
        invariants.check(prob, bad)


def test_volume_requirement_alternatives_invariant(oracle):
    zone = lambda *z: fx.req(fx.ZONE, "In", *z)
    pods = [fx.pod(requests={"cpu": "500m"}, volume_requirements=[[zone("test-zone-1")], [zone("test-zone-3")]]),
            fx.pod(requests={"cpu": "500m"}, volume_requirements=[[fx.req(fx.ZONE, "NotIn", "test-zone-1", "test-zone-2")]])]
    prob = fx.problem(fx.fake_default_instance_types(), [fx.node_pool(requirements=[zone("test-zone-2", "test-zone-3")])], pods)
    res = oracle.solve(prob)
    invariants.check(prob, res)
    assert not res["podErrors"]
    bad = copy.deepcopy(res)                       # the claim drifts to a zone no alternative admits
    for c in bad["newNodeClaims"]:
        for q in c["requirements"]:
            if q["key"] == fx.ZONE:
                q["values"] = ["test-zone-2"]
    with pytest.raises(AssertionError, match="no volume requirement alternative"):
        invariants.check(prob, bad)


def test_fullsize_checkers_accept_the_oracle_and_reject_tampering(oracle):
    """check_claims / check_topology_mix (the checkers bench.py runs on the results no oracle pin exists for): the oracle's own
    Results of the configs[2] shape pass — including the replay of every zonal-spread choice with its tie rule — and every
    tampered copy is caught: a pod moved onto a claim that already hosts an anti-affinity peer, a claim moved to another zone, a
    claim whose requests are raised above an option's allocatable, a pod listed twice."""
    import copy
    p = fx.config3(pods=4000, n_types=144, seed=9)
    r = oracle.solve(p)
    got = invariants.check_topology_mix(p, r)
    assert got["pods"] == 4000 and all(got[k] > 0 for k in ("zonal_spread", "hostname_spread", "zonal_affinity", "hostname_anti_affinity"))
    assert invariants.check_claims(p, r, expect_pods=4000)["node_claims"] == len(r["newNodeClaims"])
    nginx = {fx.group_pod_uid(g["uidSeed"], i) for g in p["podGroups"] if g["template"]["labels"].get("app") == "nginx" for i in range(g["count"])}
    hosts = [i for i, c in enumerate(r["newNodeClaims"]) if nginx & set(c["pods"])]
    bad = copy.deepcopy(r)                        # two app=nginx pods on one NodeClaim
    u = next(x for x in bad["newNodeClaims"][hosts[0]]["pods"] if x in nginx)
    bad["newNodeClaims"][hosts[0]]["pods"].remove(u)
    bad["newNodeClaims"][hosts[1]]["pods"].append(u)
    with pytest.raises(AssertionError):
        invariants.check_topology_mix(p, bad)
    bad = copy.deepcopy(r)                        # a zonal claim in another zone
    ci = next(i for i, c in enumerate(bad["newNodeClaims"]) if any(q["key"] == fx.ZONE and len(q["values"]) == 1 for q in c["requirements"]))
    for q in bad["newNodeClaims"][ci]["requirements"]:
        if q["key"] == fx.ZONE:
            q["values"] = [z for z in fx.KWOK_ZONES[:3] if z != q["values"][0]][:1]
    with pytest.raises(AssertionError):
        invariants.check_topology_mix(p, bad)
    bad = copy.deepcopy(r)                        # requests above an option's allocatable
    bad["newNodeClaims"][0]["requests"]["cpu"] = str(10 ** 15)
    with pytest.raises(AssertionError):
        invariants.check_claims(p, bad)
    bad = copy.deepcopy(r)                        # a pod listed twice
    bad["newNodeClaims"][1]["pods"].append(bad["newNodeClaims"][0]["pods"][0])
    with pytest.raises(AssertionError):
        invariants.check_topology_mix(p, bad)
