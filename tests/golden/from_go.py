"""Converts between the reference's wire shapes and this repository's problem format.

`from_go(dump)` takes a document written by go/golden_dump_test.go (run inside the reference's tree: corev1.Pod and
v1.NodePool as json.Marshal emits them, instance types flattened, plus what Solve() returned) and produces
(problem, expected): the problem in karpenter_amd/fixtures.py's format and the recorded outcome in the shape of this
repository's Results. tests/test_go_dump.py solves `problem` with the oracle and compares with `expected` — that is how
the oracle gets pinned to the real Go implementation on a machine that has Go.

`to_go(problem)` is the inverse for the subset of the format the dumps use; the round trip from_go(to_go(p)) == p is
tested on fuzzed problems, so the converter itself is covered here, without Go.
"""
import calendar
import re
import time
from decimal import Decimal

_MULT = {"": 1, "n": Decimal("1e-9"), "u": Decimal("1e-6"), "m": Decimal("0.001"), "k": 10**3, "M": 10**6, "G": 10**9, "T": 10**12, "P": 10**15,
         "Ki": 2**10, "Mi": 2**20, "Gi": 2**30, "Ti": 2**40, "Pi": 2**50}


def _q(v):
    m = re.fullmatch(r"(-?[0-9.]+)(e[0-9]+)?([a-zA-Z]*)", str(v))
    return Decimal(m.group(1) + (m.group(2) or "")) * _MULT[m.group(3)]


def _fmt(d):
    """A resource.Quantity string for an exact decimal (nano granularity)."""
    n = int(d * 10**9)
    assert Decimal(n) == d * 10**9, f"quantity finer than nano: {d}"
    return f"{n}n" if n % 10**9 else str(n // 10**9)


def _timestamp(ts):
    if not ts:
        return 0
    return calendar.timegm(time.strptime(ts, "%Y-%m-%dT%H:%M:%SZ"))


def _container_requests(c):
    res = c.get("resources") or {}
    out = dict(res.get("limits") or {})          # API defaulting: a missing request takes the limit
    out.update(res.get("requests") or {})
    return {k: _q(v) for k, v in out.items()}


def pod_requests(spec):
    """resourcehelper.PodRequests (what resources.Ceiling uses, utils/resources/resources.go:115-120) for pods without
    restartable init containers: max(sum of containers, largest init container) + overhead."""
    total = {}
    for c in spec.get("containers") or []:
        for k, v in _container_requests(c).items():
            total[k] = total.get(k, Decimal(0)) + v
    for c in spec.get("initContainers") or []:
        assert c.get("restartPolicy") != "Always", "sidecar containers are not handled by this converter"
        for k, v in _container_requests(c).items():
            total[k] = max(total.get(k, Decimal(0)), v)
    for k, v in (spec.get("overhead") or {}).items():
        total[k] = total.get(k, Decimal(0)) + _q(v)
    return {k: _fmt(v) for k, v in total.items()}


def _exprs(term):
    return [{"key": e["key"], "operator": e["operator"], "values": list(e.get("values") or [])} for e in term.get("matchExpressions") or []]


def _selector(sel):
    if sel is None:
        return None
    out = {}
    if sel.get("matchLabels") is not None:
        out["matchLabels"] = dict(sel["matchLabels"])
    if sel.get("matchExpressions"):
        out["matchExpressions"] = [{"key": e["key"], "operator": e["operator"], "values": list(e.get("values") or [])} for e in sel["matchExpressions"]]
    if "matchLabels" not in out:
        out["matchLabels"] = {}
    return out


def _aff_term(t):
    out = {"labelSelector": _selector(t.get("labelSelector")), "topologyKey": t["topologyKey"]}
    if t.get("namespaces"):
        out["namespaces"] = list(t["namespaces"])
    if t.get("namespaceSelector") is not None:
        out["namespaceSelector"] = _selector(t["namespaceSelector"])
    return out


def pod_from_go(p):
    md, spec = p.get("metadata") or {}, p.get("spec") or {}
    out = {"uid": md.get("uid", ""), "name": md.get("name") or md.get("uid", ""), "namespace": md.get("namespace") or "default",
           "labels": dict(md.get("labels") or {}), "requests": pod_requests(spec), "creationTimestamp": _timestamp(md.get("creationTimestamp")),
           "phase": (p.get("status") or {}).get("phase") or "Pending", "nodeName": spec.get("nodeName") or ""}
    if spec.get("nodeSelector"):
        out["nodeSelector"] = dict(spec["nodeSelector"])
    aff = spec.get("affinity") or {}
    na = aff.get("nodeAffinity")
    if na:
        d = {}
        req = na.get("requiredDuringSchedulingIgnoredDuringExecution")
        if req and req.get("nodeSelectorTerms"):
            d["required"] = [_exprs(t) for t in req["nodeSelectorTerms"]]
        pref = na.get("preferredDuringSchedulingIgnoredDuringExecution")
        if pref:
            d["preferred"] = [{"weight": t["weight"], "matchExpressions": _exprs(t["preference"])} for t in pref]
        if d:
            out["nodeAffinity"] = d
    for src, dst in (("podAffinity", "podAffinity"), ("podAntiAffinity", "podAntiAffinity")):
        a = aff.get(src)
        if a is not None:
            out[dst] = {"required": [_aff_term(t) for t in a.get("requiredDuringSchedulingIgnoredDuringExecution") or []],
                        "preferred": [{"weight": t["weight"], "term": _aff_term(t["podAffinityTerm"])} for t in a.get("preferredDuringSchedulingIgnoredDuringExecution") or []]}
    if spec.get("tolerations"):
        out["tolerations"] = [{"key": t.get("key", ""), "operator": t.get("operator", ""), "value": t.get("value", ""), "effect": t.get("effect", "")} for t in spec["tolerations"]]
    if spec.get("topologySpreadConstraints"):
        tscs = []
        for c in spec["topologySpreadConstraints"]:
            t = {"maxSkew": c["maxSkew"], "topologyKey": c["topologyKey"], "whenUnsatisfiable": c.get("whenUnsatisfiable", "DoNotSchedule"),
                 "labelSelector": _selector(c.get("labelSelector"))}
            for k in ("minDomains", "nodeAffinityPolicy", "nodeTaintsPolicy"):
                if c.get(k) is not None:
                    t[k] = c[k]
            if c.get("matchLabelKeys"):
                t["matchLabelKeys"] = list(c["matchLabelKeys"])
            tscs.append(t)
        out["topologySpreadConstraints"] = tscs
    if md.get("annotations"):
        out["annotations"] = dict(md["annotations"])
    if spec.get("priority") is not None:
        out["priority"] = spec["priority"]
    return out


def _reqs(rs):
    out = []
    for r in rs or []:
        x = {"key": r["key"], "operator": r["operator"], "values": list(r.get("values") or [])}
        if r.get("minValues") is not None:
            x["minValues"] = r["minValues"]
        out.append(x)
    return out


def nodepool_from_go(np_):
    spec = np_.get("spec") or {}
    tmpl = spec.get("template") or {}
    tspec = tmpl.get("spec") or {}
    ref = tspec.get("nodeClassRef") or {}
    out = {"name": np_["metadata"]["name"], "weight": spec.get("weight") or 0, "requirements": _reqs(tspec.get("requirements")),
           "labels": dict((tmpl.get("metadata") or {}).get("labels") or {}),
           "taints": [{"key": t.get("key", ""), "value": t.get("value", ""), "effect": t.get("effect", "")} for t in tspec.get("taints") or []],
           "nodeClassLabelKey": f"{ref.get('group', '')}/{ref.get('kind', '').lower()}", "nodeClassName": ref.get("name", "")}
    if spec.get("limits"):
        out["limits"] = dict(spec["limits"])
    if spec.get("replicas") is not None:
        out["static"] = True
    return out


def instance_type_from_go(it):
    offs = []
    for o in it.get("offerings") or []:
        d = {"requirements": _reqs(o["requirements"]), "price": o["price"], "available": o["available"]}
        if o.get("reservationCapacity"):
            d["reservationCapacity"] = o["reservationCapacity"]
        offs.append(d)
    return {"name": it["name"], "requirements": _reqs(it["requirements"]), "capacity": dict(it["capacity"]), "overhead": dict(it.get("overhead") or {}), "offerings": offs}


def from_go(dump):
    problem = {"namespaces": [], "wellKnownLabels": list(dump["wellKnownLabels"]), "options": {"preferencePolicy": dump.get("preferencePolicy", "Respect")},
               "instanceTypes": [instance_type_from_go(t) for t in dump["instanceTypes"]], "nodePools": [nodepool_from_go(n) for n in dump["nodePools"]],
               "stateNodes": [], "pods": [pod_from_go(p) for p in dump["pods"]], "podGroups": [], "daemonSetPods": [], "clusterPods": [], "deletingNodeNames": []}
    expected = {"newNodeClaims": [{"nodePool": c["nodePool"], "pods": list(c["pods"] or []), "instanceTypes": list(c["instanceTypes"] or []),
                                   "requirements": _reqs(c["requirements"]), "requests": dict(c.get("requests") or {})} for c in dump["results"]["newNodeClaims"] or []],
                "podErrors": dict(dump["results"].get("podErrors") or {})}
    return problem, expected


def selector_form(q):
    """Requirement.NodeSelectorRequirements() (requirement.go:121-176) of a requirement as this repository reports it
    ({key, complement, values, gte, lte}): a list of (key, operator, sorted values)."""
    if q.get("gte") is not None:
        out = [(q["key"], "Gte", (str(q["gte"]),))]
        if q.get("lte") is not None:
            out.append((q["key"], "Lte", (str(q["lte"]),)))
        return out
    if q.get("lte") is not None:
        return [(q["key"], "Lte", (str(q["lte"]),))]
    if q["complement"]:
        return [(q["key"], "NotIn" if q["values"] else "Exists", tuple(sorted(q["values"])))]
    return [(q["key"], "In" if q["values"] else "DoesNotExist", tuple(sorted(q["values"])))]


# ---- inverse (for the round-trip test) ------------------------------------------------------------------------------

def _go_selector(sel):
    if sel is None:
        return None
    out = {}
    if sel.get("matchLabels"):
        out["matchLabels"] = dict(sel["matchLabels"])
    if sel.get("matchExpressions"):
        out["matchExpressions"] = [dict(e) for e in sel["matchExpressions"]]
    return out


def _go_aff_term(t):
    out = {"labelSelector": _go_selector(t.get("labelSelector")), "topologyKey": t["topologyKey"]}
    if t.get("namespaces"):
        out["namespaces"] = list(t["namespaces"])
    if t.get("namespaceSelector") is not None:
        out["namespaceSelector"] = _go_selector(t["namespaceSelector"])
    return out


def pod_to_go(p):
    md = {"name": p.get("name", p["uid"]), "namespace": p.get("namespace", "default"), "uid": p["uid"], "labels": dict(p.get("labels") or {})}
    if p.get("creationTimestamp"):
        md["creationTimestamp"] = time.strftime("%Y-%m-%dT%H:%M:%SZ", time.gmtime(p["creationTimestamp"]))
    if p.get("annotations"):
        md["annotations"] = dict(p["annotations"])
    spec = {"containers": [{"name": "c", "resources": {"requests": dict(p.get("requests") or {})}}]}
    if p.get("nodeName"):
        spec["nodeName"] = p["nodeName"]
    if p.get("nodeSelector"):
        spec["nodeSelector"] = dict(p["nodeSelector"])
    aff = {}
    na = p.get("nodeAffinity")
    if na:
        d = {}
        if na.get("required"):
            d["requiredDuringSchedulingIgnoredDuringExecution"] = {"nodeSelectorTerms": [{"matchExpressions": [dict(e) for e in term]} for term in na["required"]]}
        if na.get("preferred"):
            d["preferredDuringSchedulingIgnoredDuringExecution"] = [{"weight": t["weight"], "preference": {"matchExpressions": [dict(e) for e in t["matchExpressions"]]}} for t in na["preferred"]]
        aff["nodeAffinity"] = d
    for key in ("podAffinity", "podAntiAffinity"):
        a = p.get(key)
        if a is not None:
            aff[key] = {"requiredDuringSchedulingIgnoredDuringExecution": [_go_aff_term(t) for t in a.get("required") or []],
                        "preferredDuringSchedulingIgnoredDuringExecution": [{"weight": t["weight"], "podAffinityTerm": _go_aff_term(t["term"])} for t in a.get("preferred") or []]}
    if aff:
        spec["affinity"] = aff
    if p.get("tolerations"):
        spec["tolerations"] = [{k: v for k, v in t.items() if v != ""} for t in p["tolerations"]]
    if p.get("topologySpreadConstraints"):
        spec["topologySpreadConstraints"] = [dict(c, labelSelector=_go_selector(c.get("labelSelector"))) for c in p["topologySpreadConstraints"]]
    if p.get("priority") is not None:
        spec["priority"] = p["priority"]
    return {"metadata": md, "spec": spec, "status": {"phase": p.get("phase", "Pending")}}


def nodepool_to_go(np_):
    group, _, kind = np_.get("nodeClassLabelKey", "karpenter.test.sh/testnodeclass").partition("/")
    tspec = {"requirements": [dict(r) for r in np_.get("requirements") or []], "nodeClassRef": {"group": group, "kind": kind, "name": np_.get("nodeClassName", "default")}}
    if np_.get("taints"):
        tspec["taints"] = [dict(t) for t in np_["taints"]]
    spec = {"template": {"metadata": {"labels": dict(np_.get("labels") or {})}, "spec": tspec}}
    if np_.get("weight"):
        spec["weight"] = np_["weight"]
    if np_.get("limits") is not None:
        spec["limits"] = dict(np_["limits"])
    return {"metadata": {"name": np_["name"]}, "spec": spec}


def to_go(problem, results=None, name="round-trip"):
    return {"name": name, "preferencePolicy": problem.get("options", {}).get("preferencePolicy", "Respect"), "wellKnownLabels": list(problem["wellKnownLabels"]),
            "nodePools": [nodepool_to_go(n) for n in problem["nodePools"]],
            "instanceTypes": [{"name": t["name"], "requirements": [dict(r) for r in t["requirements"]], "capacity": dict(t["capacity"]), "overhead": dict(t["overhead"]),
                               "offerings": [{"requirements": [dict(r) for r in o["requirements"]], "price": o["price"], "available": o.get("available", True),
                                              "reservationCapacity": o.get("reservationCapacity", 0)} for o in t["offerings"]]} for t in problem["instanceTypes"]],
            "pods": [pod_to_go(p) for p in problem["pods"]],
            "results": {"newNodeClaims": [{"nodePool": c["nodePool"], "pods": list(c["pods"]), "instanceTypes": list(c["instanceTypes"]),
                                           "requirements": [{"key": k, "operator": op, "values": list(vals)} for q in c["requirements"] for k, op, vals in selector_form(q)],
                                           "requests": {k: f"{int(v)}n" for k, v in c["requests"].items()}} for c in (results or {}).get("newNodeClaims", [])],   # this repository reports nano units
                        "podErrors": {u: str(e) for u, e in (results or {}).get("podErrors", {}).items()}}}
