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


def _topology_selectors(p):
    """[(namespaces, matchLabels)] of every topology group the pod can own — spread constraints, required and preferred pod
    (anti-)affinity terms (topology.go:461-533) — or None when a term cannot be evaluated here (matchExpressions,
    namespaceSelector). A group counts exactly the pods its namespaces + selector match (topologygroup.go:442), whatever
    NodePool they land on: those pods and the owner must be solved together."""
    out = []
    ns = p.get("namespace", "default")
    # (a group on any key but the hostname draws its domain universe from EVERY NodePool — buildDomainGroups, topology.go:104-142 —
    # and domainMinCount takes the minimum over all of it, topologygroup.go:300-322: its owner cannot be cut off from the other pools)
    for c in p.get("topologySpreadConstraints") or []:
        sel = c.get("labelSelector") or {}
        if c.get("topologyKey") != fx.HOSTNAME or sel.get("matchExpressions"):
            return None
        out.append(({ns}, dict(sel.get("matchLabels") or {})))
    for field in ("podAffinity", "podAntiAffinity"):
        aff = p.get(field) or {}
        terms = list(aff.get("required") or []) + [w["term"] for w in (aff.get("preferred") or [])]
        for t in terms:
            sel = t.get("labelSelector") or {}
            if t.get("topologyKey") != fx.HOSTNAME or sel.get("matchExpressions") or t.get("namespaceSelector") is not None:
                return None
            out.append((set(t.get("namespaces") or [ns]), dict(sel.get("matchLabels") or {})))
    return out


def _pinned_pools(p, pools):
    """The NodePools a pod can ever land on, as far as its REQUIRED constraints on `karpenter.sh/nodepool` say (a node
    selector, and/or every required node-affinity term carrying `In [...]` on that key — terms are OR-ed and relaxation only
    drops terms, preferences.go:38-57, so the union over the terms bounds every relaxed variant). None = not provably pinned."""
    allowed = None
    sel = (p.get("nodeSelector") or {}).get(fx.NODEPOOL)
    if sel is not None:
        allowed = {sel}
    terms = (p.get("nodeAffinity") or {}).get("required") or []
    if terms:
        union = set()
        for term in terms:
            ins = [set(q["values"]) for q in term if q["key"] == fx.NODEPOOL and q["operator"] == "In"]
            if not ins:
                union = None        # a term without the pin can reach any pool
                break
            union |= set.intersection(*ins)
        if union is not None:
            allowed = union if allowed is None else allowed & union
    if allowed is None:
        return None
    allowed &= set(pools)
    return allowed or None


def split_components(problem, bins=None):
    """The host library's split (ksched_split_components in karpenter_amd/host/ksched.cpp — what a Go controller calls through
    go/ksolve_components.go): [(tuple of pool names, sub-problem)] in NodePool order, or None when independence cannot be shown.
    With `bins` = N also the deal of the components over N devices: ([(pools, sub-problem)], [[component index] per device]),
    components by pod count, largest first, each to the device with the fewest pods so far (LPT). `split_components_reference`
    below is the same rule in Python, kept as the cross-check of tests/test_components.py."""
    from .scheduling import SplitComponents
    out = SplitComponents(problem, bins or 1)
    if out["components"] is None:
        return None
    parts = [(tuple(c["pools"]), c["problem"]) for c in out["components"]]
    return parts if bins is None else (parts, out["bins"])


def split_components_reference(problem):
    """Connected components of the pods x NodePools graph: two NodePools are in one component when some
    pod may land on either, or when a topology group owned by a pod of one (spread constraint, pod affinity / anti-affinity
    term) selects a pod of the other — the group's domain counts move with every selected pod that is placed
    (topology.go:197-224), so owner and selected pods stay in one Solve(). Returns [(tuple of pool names, sub-problem)] in NodePool order, or None when some pod is not
    provably pinned or the batch has anything `split_by_nodepool` refuses. Same contract as `split_by_nodepool`: each
    component is a packing problem of its own, solved bit-exactly as such; the union is a packing of equal quality, not the
    reference's pod-for-pod answer for the whole batch."""
    if problem.get("stateNodes") or problem.get("clusterPods") or problem.get("options", {}).get("reservedCapacity"):
        return None
    names = [np_["name"] for np_ in problem["nodePools"]]
    parent = {n: n for n in names}

    def find(x):
        while parent[x] != x:
            parent[x] = parent[parent[x]]
            x = parent[x]
        return x

    owners = []
    for kind, items in (("pods", problem.get("pods", [])), ("podGroups", problem.get("podGroups", []))):
        for item in items:
            allowed = _pinned_pools(item["template"] if kind == "podGroups" else item, names)
            if allowed is None:
                return None
            first = min(allowed, key=names.index)
            for other in allowed:
                parent[find(other)] = find(first)
            owners.append((kind, item, first))
    # topology groups tie their owner to every pod they select
    by_sig = {}
    for kind, item, first in owners:
        t = item["template"] if kind == "podGroups" else item
        by_sig.setdefault((t.get("namespace", "default"), tuple(sorted((t.get("labels") or {}).items()))), []).append(first)
    for kind, item, first in owners:
        sels = _topology_selectors(item["template"] if kind == "podGroups" else item)
        if sels is None:
            return None
        for namespaces, match in sels:
            for (ns, labels), firsts in by_sig.items():
                if ns in namespaces and all(dict(labels).get(k) == v for k, v in match.items()):
                    for f in firsts:
                        parent[find(f)] = find(first)
    comps = {}
    for n in names:
        comps.setdefault(find(n), {"pools": [], "pods": [], "podGroups": []})["pools"].append(n)
    for kind, item, first in owners:
        comps[find(first)][kind].append(item)
    by_name = {np_["name"]: np_ for np_ in problem["nodePools"]}
    out = []
    for c in sorted(comps.values(), key=lambda c: names.index(c["pools"][0])):
        if c["pods"] or c["podGroups"]:
            out.append((tuple(c["pools"]), dict(problem, nodePools=[by_name[n] for n in c["pools"]], pods=c["pods"], podGroups=c["podGroups"])))
    return out
