#!/usr/bin/env python3
"""Transcribes the reference's own truth tables for the Solve() path into JSON golden vectors.

Run in the build container (needs /root/reference, which does not exist on the GPU box):
    python tests/golden/make_golden.py
Sources (all under /root/reference):
  pkg/scheduling/requirement_test.go:33-64   requirement definitions
  pkg/scheduling/requirement_test.go:103-747 three 14x14 Intersection tables
  pkg/scheduling/requirement_test.go:749-873 Has / Operator / Len tables
  pkg/scheduling/requirements_test.go:41-56  requirement-set definitions (zone key)
  pkg/scheduling/requirements_test.go:57-543 15x15 Compatible tables, loose (AllowUndefinedWellKnownLabels) and strict
The Go source is only read as data (regex over `Entry(nil, a, b, c)` / `Expect(a.Compatible(b...)).To(Succeed())` lines);
no reference code is copied into the repository.
"""
import json
import os
import re
import sys

REF = "/root/reference/pkg/scheduling"
OUT = os.path.dirname(os.path.abspath(__file__))
OPS = {"NodeSelectorOpIn": "In", "NodeSelectorOpNotIn": "NotIn", "NodeSelectorOpExists": "Exists", "NodeSelectorOpDoesNotExist": "DoesNotExist",
       "NodeSelectorOpGt": "Gt", "NodeSelectorOpLt": "Lt", "NodeSelectorOpGte": "Gte", "NodeSelectorOpLte": "Lte"}
KEYS = {"corev1.LabelTopologyZone": "topology.kubernetes.io/zone", '"key"': "key"}


def parse_defs(src, wrapped):
    """name := NewRequirement("key", corev1.NodeSelectorOpIn, "A") / NewRequirementWithFlexibility(key, op, new(1), vals...)"""
    defs = {}
    pat = re.compile(r"^\s*(\w+)\s*:=\s*(NewRequirements\()?NewRequirement(WithFlexibility)?\((.*)\)\s*$", re.M)
    for m in pat.finditer(src):
        name, flex, args = m.group(1), m.group(3), m.group(4)
        if m.group(2):
            args = args[:-1]  # closing paren of NewRequirements(
        parts = [a.strip() for a in re.split(r",\s*(?![^()]*\))", args)]
        key = KEYS.get(parts[0], parts[0].strip('"'))
        op = OPS[parts[1].split(".")[-1]]
        rest = parts[2:]
        mv = None
        if flex:
            mm = re.fullmatch(r"new\((\d+)\)?", rest[0])
            mv = int(mm.group(1))
            rest = rest[1:]
        vals = [v.strip('"') for v in rest]
        d = {"key": key, "operator": op, "values": vals}
        if mv is not None:
            d["minValues"] = mv
        if name not in defs:
            defs[name] = d
    return defs


def bound_of(defs, ref):
    """greaterThan1.gte -> canonical inclusive bound (Gt n => gte n+1, Lt n => lte n-1; requirement.go:82-98)."""
    name, field = ref.split(".")
    d = defs[name]
    v = int(d["values"][0])
    return {"Gt": v + 1, "Gte": v, "Lt": v - 1, "Lte": v}[d["operator"]]


def parse_literal(text, defs):
    """&Requirement{Key: "key", complement: true, gte: greaterThan1.gte, values: sets.New("2"), MinValues: new(1)}"""
    body = text[text.index("{") + 1: text.rindex("}")]
    out = {"struct": True, "key": "key", "complement": False, "values": [], "gte": None, "lte": None, "minValues": None}
    out["complement"] = re.search(r"complement:\s*(true|false)", body).group(1) == "true"
    mv = re.search(r"MinValues:\s*new\((\d+)\)", body)
    if mv:
        out["minValues"] = int(mv.group(1))
    for b in ("gte", "lte"):
        mb = re.search(b + r":\s*(\w+\.\w+)", body)
        if mb:
            out[b] = bound_of(defs, mb.group(1))
    vs = re.search(r"values:\s*sets\.(?:New(?:\[string\])?\(([^)]*)\)|Set\[string\]\{\})", body)
    if vs and vs.group(1):
        out["values"] = sorted(v.strip().strip('"') for v in vs.group(1).split(","))
    return out


def section(src, start_pat, end_pat):
    s = re.search(start_pat, src).end()
    e = re.search(end_pat, src[s:]).start() + s
    return src[s:e]


def main():
    if not os.path.isdir(REF):
        sys.exit("reference not present; golden vectors are generated in the build container only")
    rt = open(os.path.join(REF, "requirement_test.go")).read()
    head = rt[: rt.index('Context("NewRequirements"')]
    defs = parse_defs(head, False)
    out = {"source": "pkg/scheduling/requirement_test.go", "definitions": defs, "intersection": [], "has": [], "operator": [], "len": []}
    inter = section(rt, r'Context\("Intersect requirements"', r'Context\("Has"')
    for m in re.finditer(r"Entry\(nil,\s*(\w+),\s*(\w+),\s*(\w+|&Requirement\{.*\})\),?\s*$", inter, re.M):
        exp = m.group(3)
        out["intersection"].append([m.group(1), m.group(2), exp if not exp.startswith("&") else parse_literal(exp, defs)])
    has = section(rt, r'Context\("Has"', r'Context\("Operator"')
    for m in re.finditer(r'Entry\(nil,\s*(\w+),\s*"([^"]*)",\s*Be(True|False)\(\)\)', has):
        out["has"].append([m.group(1), m.group(2), m.group(3) == "True"])
    op = section(rt, r'Context\("Operator"', r'Context\("Len"')
    for m in re.finditer(r"Entry\(nil,\s*(\w+),\s*corev1\.(\w+)\)", op):
        out["operator"].append([m.group(1), OPS[m.group(2)]])
    ln = section(rt, r'Context\("Len"', r'Context\("Any"')
    for m in re.finditer(r"Entry\(nil,\s*(\w+),\s*([\w.\-]+)\)", ln):
        v = m.group(2)
        val = (2**63 - 1) - int(v.split("-")[1]) if v.startswith("math.MaxInt64-") else (2**63 - 1 if v == "math.MaxInt64" else int(v))
        out["len"].append([m.group(1), val])
    assert len(out["intersection"]) == 590, len(out["intersection"])  # 196 + 197 + 197 entries in the three tables
    json.dump(out, open(os.path.join(OUT, "requirement_tables.json"), "w"), indent=0)

    rs = open(os.path.join(REF, "requirements_test.go")).read()
    comp = section(rs, r'Context\("Compatibility"', r'Context\("Error Messages"')
    sets_ = parse_defs(comp, True)
    sets_["unconstrained"] = None
    loose, strict = [], []
    for m in re.finditer(r"Expect\((\w+)\.Compatible\((\w+)(, AllowUndefinedWellKnownLabels)?\)\)\.(To|ToNot)\(Succeed\(\)\)", comp):
        row = [m.group(1), m.group(2), m.group(4) == "To"]
        (loose if m.group(3) else strict).append(row)
    assert len(loose) == 15 * 15 and len(strict) == 15 * 15, (len(loose), len(strict))
    json.dump({"source": "pkg/scheduling/requirements_test.go", "definitions": sets_, "loose": loose, "strict": strict},
              open(os.path.join(OUT, "requirements_compatible.json"), "w"), indent=0)
    print("wrote requirement_tables.json (%d intersections, %d has, %d operator, %d len) and requirements_compatible.json (%d loose, %d strict)" % (
        len(out["intersection"]), len(out["has"]), len(out["operator"]), len(out["len"]), len(loose), len(strict)))


def kwok_catalogue():
    """The KWOK provider's stock catalogue (kwok/cloudprovider/instance_types.json, embedded by helpers.go:67-68 and turned
    into cloudprovider.InstanceTypes by ConstructInstanceTypes, :70-95): a digest fixtures.kwok_instance_types() must
    reproduce exactly — names, resources, architecture, operating systems and every offering's capacity type, zone and
    price (bit-exact floats)."""
    root = os.path.dirname(os.path.dirname(REF))
    path = os.path.join(root, "kwok", "cloudprovider", "instance_types.json")
    raw = json.load(open(path))
    out = []
    for t in raw:
        offs = []
        for o in t["offerings"]:
            r = {q["key"]: q["values"][0] for q in o["Requirements"]}
            offs.append([r["karpenter.sh/capacity-type"], r["topology.kubernetes.io/zone"], o["Price"]])
        out.append({"name": t["name"], "architecture": t["architecture"], "operatingSystems": t["operatingSystems"], "resources": t["resources"], "offerings": offs})
    json.dump({"source": "kwok/cloudprovider/instance_types.json", "instanceTypes": out}, open(os.path.join(OUT, "kwok_instance_types.json"), "w"), separators=(",", ":"))
    print("kwok catalogue:", len(out), "instance types")




if __name__ == "__main__":
    main()
    kwok_catalogue()
