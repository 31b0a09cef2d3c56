"""Splitting one provisioning batch into the NodePool components SURVEY.md §8(e) describes, for sharding across GPUs.

A component is the set of pods that can only ever land on one NodePool, together with that pool. `split_by_nodepool`
returns one sub-problem per pool when it can PROVE the pools do not interact through anything Solve() models:

  * every pod pins `karpenter.sh/nodepool` to exactly one pool with a node selector (so no relaxation can move it),
  * no existing nodes, no cluster pods (they would be shared bins / shared topology counts),
  * no topology spread / pod affinity / anti-affinity (a group may select pods of several pools),
  * no reserved capacity (reservations are shared between pools, reservationmanager.go:28-110).

Even then the split is NOT bit-identical to solving the whole batch in one Solve(): the reference re-sorts ALL in-flight
NodeClaims with an unstable sort before every scan (scheduler.go:598), so which of two equally full claims of pool A a pod
joins depends on where pool B's claims sit in the array. The components are independent as packing problems — same
number of NodeClaims, same or nearly the same cost (tests/test_components.py) — but pod identities per claim differ.
Exact multi-GPU scaling therefore shards across Solve() CALLS (consolidation probes, separate provisioning passes:
ksolve_solve_batch, bench.py --gpus N); use this split when the caller accepts a packing of equal quality instead of the
reference's exact one.
"""
from . import fixtures as fx


def split_by_nodepool(problem):
    """[(pool name, sub-problem)] in NodePool order, or None when independence cannot be shown."""
    if problem.get("stateNodes") or problem.get("clusterPods") or problem.get("options", {}).get("reservedCapacity"):
        return None
    pools = {np_["name"]: np_ for np_ in problem["nodePools"]}
    by_pool = {name: {"pods": [], "podGroups": []} for name in pools}

    def owner(p):
        if p.get("topologySpreadConstraints") or p.get("podAffinity") or p.get("podAntiAffinity"):
            return None
        name = (p.get("nodeSelector") or {}).get(fx.NODEPOOL)
        return name if name in pools else None

    for p in problem.get("pods", []):
        o = owner(p)
        if o is None:
            return None
        by_pool[o]["pods"].append(p)
    for g in problem.get("podGroups", []):
        o = owner(g["template"])
        if o is None:
            return None
        by_pool[o]["podGroups"].append(g)
    out = []
    for np_ in problem["nodePools"]:
        part = by_pool[np_["name"]]
        if part["pods"] or part["podGroups"]:
            out.append((np_["name"], dict(problem, nodePools=[np_], pods=part["pods"], podGroups=part["podGroups"])))
    return out
