"""Problem builders: synthetic instance types, NodePools and pods, as the JSON problem document both the product host
library (karpenter_amd/host) and the test oracle read.

Each builder mirrors a fixture of the reference so parity tests read like the reference's own tests:
  fake_instance_type / fake_instance_types / fake_default_instance_types / fake_instance_types_assorted
      -> pkg/cloudprovider/fake/instancetype.go:99-205, :411-439, :369-409 ; fake/cloudprovider.go:220-267
  kwok_instance_types   -> kwok/tools/gen_instance_types.go:36-113 + kwok/cloudprovider/helpers.go:125-214
  pod(...)              -> pkg/test/pods.go:88 (test.Pod / PodOptions)
  node_pool(...)        -> pkg/test/nodepool.go:35 (test.NodePool)
Quantities are Kubernetes quantity strings ("100m", "4Gi", "1.8G").
"""
from __future__ import annotations

import copy
import math
import random
import re

ZONE = "topology.kubernetes.io/zone"
HOSTNAME = "kubernetes.io/hostname"
ARCH = "kubernetes.io/arch"
OS = "kubernetes.io/os"
INSTANCE_TYPE = "node.kubernetes.io/instance-type"
CAPACITY_TYPE = "karpenter.sh/capacity-type"
NODEPOOL = "karpenter.sh/nodepool"

# pkg/cloudprovider/fake/instancetype.go:36-52
FAKE_LABEL_INSTANCE_SIZE = "size"
FAKE_EXOTIC_LABEL = "special"
FAKE_INTEGER_LABEL = "integer"
FAKE_WELL_KNOWN = [FAKE_LABEL_INSTANCE_SIZE, FAKE_EXOTIC_LABEL, FAKE_INTEGER_LABEL, "karpenter.sh/reservation-id"]
GPU_VENDOR_A = "fake.com/vendor-a"
GPU_VENDOR_B = "fake.com/vendor-b"

# kwok/apis/v1alpha1/labels.go:25-46
KWOK_SIZE = "karpenter.kwok.sh/instance-size"
KWOK_FAMILY = "karpenter.kwok.sh/instance-family"
KWOK_MEMORY = "karpenter.kwok.sh/instance-memory"
KWOK_CPU = "karpenter.kwok.sh/instance-cpu"
KWOK_WELL_KNOWN = [KWOK_SIZE, KWOK_FAMILY, KWOK_CPU, KWOK_MEMORY]
KWOK_ZONES = ["test-zone-a", "test-zone-b", "test-zone-c", "test-zone-d"]

_SUFFIX = {"": 1.0, "m": 1e-3, "k": 1e3, "M": 1e6, "G": 1e9, "T": 1e12, "Ki": 2.0**10, "Mi": 2.0**20, "Gi": 2.0**30, "Ti": 2.0**40}


def quantity_float(q) -> float:
    """resource.Quantity.AsApproximateFloat64 for the quantities the fixtures use."""
    m = re.fullmatch(r"([0-9.]+)([a-zA-Z]*)", str(q))
    return float(m.group(1)) * _SUFFIX[m.group(2)]


def req(key, operator, *values, min_values=None):
    if len(values) == 1 and isinstance(values[0], (list, tuple)):
        values = tuple(values[0])
    r = {"key": key, "operator": operator, "values": [str(v) for v in values]}
    if min_values is not None:
        r["minValues"] = min_values
    return r


def offering(capacity_type, zone, price, available=True, reservation_id=None, reservation_capacity=0):
    reqs = [req(CAPACITY_TYPE, "In", capacity_type), req(ZONE, "In", zone)]
    if reservation_id is not None:
        reqs.append(req("karpenter.sh/reservation-id", "In", reservation_id))
    o = {"requirements": reqs, "price": price, "available": available}
    if reservation_capacity:
        o["reservationCapacity"] = reservation_capacity
    return o


def fake_price(resources) -> float:
    """fake.PriceFromResources — fake/instancetype.go:426-439."""
    price = 0.0
    for k, v in resources.items():
        if k == "cpu":
            price += 0.1 * quantity_float(v)
        elif k == "memory":
            price += 0.1 * quantity_float(v) / 1e9
        elif k in (GPU_VENDOR_A, GPU_VENDOR_B):
            price += 1.0
    return price


def fake_instance_type(name, resources=None, offerings=None, architecture=None, operating_systems=None, requirements=None):
    """fake.NewInstanceType — fake/instancetype.go:99-205."""
    res = dict(resources or {})
    res.setdefault("cpu", "4")
    res.setdefault("memory", "4Gi")
    res.setdefault("pods", "5")
    if not offerings:
        p = fake_price(res)
        offerings = [offering("spot", "test-zone-1", p), offering("spot", "test-zone-2", p), offering("on-demand", "test-zone-1", p),
                     offering("on-demand", "test-zone-2", p), offering("on-demand", "test-zone-3", p)]
    arch = architecture or "amd64"
    oses = sorted(operating_systems) if operating_systems else sorted(["linux", "windows", "darwin"])
    avail = [o for o in offerings if o.get("available", True)]

    def off_val(o, key):
        return next(r["values"][0] for r in o["requirements"] if r["key"] == key)

    cpu_val = int(quantity_float(res["cpu"]))  # Quantity.Value() rounds up; fixtures use whole CPUs
    reqs = [
        req(INSTANCE_TYPE, "In", name),
        req(ARCH, "In", arch),
        req(OS, "In", *oses),
        req(ZONE, "In", *[off_val(o, ZONE) for o in avail]),
        req(CAPACITY_TYPE, "In", *[off_val(o, CAPACITY_TYPE) for o in avail]),
        req(FAKE_INTEGER_LABEL, "In", cpu_val),
    ]
    large = quantity_float(res["cpu"]) > 4 and quantity_float(res["memory"]) > 8 * 2.0**30
    if large:
        reqs += [req(FAKE_LABEL_INSTANCE_SIZE, "In", "large"), req(FAKE_EXOTIC_LABEL, "In", "optional")]
    else:
        reqs += [req(FAKE_LABEL_INSTANCE_SIZE, "In", "small"), req(FAKE_EXOTIC_LABEL, "DoesNotExist")]
    # NOTE the reference builds size/special as DoesNotExist and then Insert()s a value, which turns them into In
    # (complement stays false, instancetype.go:170-185); extra requirements are intersected in before that.
    reqs += list(requirements or [])
    return {"name": name, "requirements": reqs, "capacity": res, "overhead": {"cpu": "100m", "memory": "10Mi"}, "offerings": offerings}


def fake_instance_types(total):
    """fake.InstanceTypes — fake/instancetype.go:411-424."""
    return [fake_instance_type(f"fake-it-{i}", {"cpu": str(i + 1), "memory": f"{(i + 1) * 2}Gi", "pods": str((i + 1) * 10)}) for i in range(total)]


def fake_default_instance_types():
    """fake CloudProvider default catalogue — fake/cloudprovider.go:232-262."""
    return [
        fake_instance_type("default-instance-type"),
        fake_instance_type("small-instance-type", {"cpu": "2", "memory": "2Gi"}),
        fake_instance_type("gpu-vendor-instance-type", {GPU_VENDOR_A: "2"}),
        fake_instance_type("gpu-vendor-b-instance-type", {GPU_VENDOR_B: "2"}),
        fake_instance_type("arm-instance-type", {"cpu": "16", "memory": "128Gi"}, architecture="arm64", operating_systems=["ios", "linux", "windows", "darwin"]),
        fake_instance_type("single-pod-instance-type", {"pods": "1"}),
    ]


def fake_instance_types_assorted():
    """fake.InstanceTypesAssorted — fake/instancetype.go:369-409."""
    out = []
    for cpu in [1, 2, 4, 8, 16, 32, 64]:
        for mem in [1, 2, 4, 8, 16, 32, 64, 128]:
            for zone in ["test-zone-1", "test-zone-2", "test-zone-3"]:
                for ct in ["spot", "on-demand"]:
                    for os_ in ["linux", "windows"]:
                        for arch in ["amd64", "arm64"]:
                            res = {"cpu": str(cpu), "memory": f"{mem}Gi"}
                            out.append(fake_instance_type(f"{cpu}-cpu-{mem}-mem-{arch}-{os_}-{zone}-{ct}", res, [offering(ct, zone, fake_price(res))], arch, [os_]))
    return out


def kwok_instance_types(cpus=(1, 2, 4, 8, 16, 32, 48, 64, 96, 128, 192, 256), mem_factors=(2, 4, 8), oses=("linux", "windows"),
                        archs=("amd64", "arm64"), zones=KWOK_ZONES, limit=None):
    """kwok generic catalogue — gen_instance_types.go:68-113 through helpers.go:125-214 (newInstanceType).

    The stock grid gives 144 types. Wider grids (more cpu sizes / memory factors) keep the same naming and price rule
    and are used for the 500- and 1000-type configurations.
    """
    out = []
    for cpu in cpus:
        for mf in mem_factors:
            for os_ in oses:
                for arch in archs:
                    family = {2: "c", 4: "s", 8: "m"}.get(mf, "e")
                    name = f"{family}-{cpu}x-{arch}-{os_}" if mf in (2, 4, 8) else f"e{mf}-{cpu}x-{arch}-{os_}"
                    mem = cpu * mf
                    pods = max(0, min(cpu * 16, 1024))
                    mem_q = f"{mem // 1024}Ti" if mem % 1024 == 0 else f"{mem}Gi"      # resource.Quantity's canonical form (1024Gi prints as 1Ti)
                    res = {"cpu": str(cpu), "memory": mem_q, "pods": str(pods), "ephemeral-storage": "20Gi"}
                    price = 0.025 * cpu + 0.001 * (mem * 2.0**30) / 1e9
                    offs = []
                    for z in zones:
                        for ct in ["spot", "on-demand"]:
                            offs.append(offering(ct, z, price * 0.7 if ct == "spot" else price))
                    fam = re.split(r"[.-]", name, maxsplit=1)[0]
                    reqs = [req(INSTANCE_TYPE, "In", name), req(ARCH, "In", arch), req(OS, "In", os_), req(ZONE, "In", *zones),
                            req(CAPACITY_TYPE, "In", "spot", "on-demand"), req(KWOK_SIZE, "In", str(cpu)), req(KWOK_FAMILY, "In", fam),
                            req(KWOK_CPU, "In", str(cpu)), req(KWOK_MEMORY, "In", mem_q)]
                    out.append({"name": name, "requirements": reqs, "capacity": res, "overhead": {"cpu": "100m", "memory": "10Mi"}, "offerings": offs})
                    if limit and len(out) >= limit:
                        return out
    return out


def kwok_catalog(n):
    """The N-type kwok-style catalogues of BASELINE.json's configs: 50 -> first 50 linux types of the stock grid,
    144 -> stock grid, 500 / 1000 -> widened cpu grid x memory factors (names stay unique)."""
    if n <= 72:
        return kwok_instance_types(oses=("linux",))[:n]
    if n <= 144:
        return kwok_instance_types()[:n]
    cpus = (1, 2, 4, 8, 12, 16, 24, 32, 48, 64, 96, 128, 192, 256)
    mfs = (2, 3, 4, 5, 6, 8, 10, 12, 16, 20, 24, 32)
    out = []
    for mf in mfs:
        out += kwok_instance_types(cpus=cpus, mem_factors=(mf,))
    if n > len(out):
        out += kwok_instance_types(cpus=(3, 6, 10, 20, 40, 80, 160, 224), mem_factors=mfs)
    assert len(out) >= n, (len(out), n)
    assert len({t["name"] for t in out}) == len(out)
    return out[:n]


_uid_counter = [0]


def pod(uid=None, name=None, namespace="default", labels=None, requests=None, node_selector=None, node_requirements=None,
        node_preferences=None, tolerations=None, topology_spread=None, pod_requirements=None, pod_preferences=None,
        pod_anti_requirements=None, pod_anti_preferences=None, creation=0, phase="Pending", node_name="", host_ports=None,
        volume_requirements=None, volumes=None):
    """test.Pod — pkg/test/pods.go:88. node_requirements: list of NodeSelectorRequirement (one term) or list of terms.
    host_ports: port numbers (PodOptions.HostPorts: TCP, no hostIP) or {"port", "ip", "protocol"} dicts.
    volume_requirements: volumeReqsByPod[uid] (scheduler.go:138) — a list of alternatives, each a list of requirements."""
    if uid is None:
        _uid_counter[0] += 1
        uid = f"00000000-0000-0000-0000-{_uid_counter[0]:012d}"
    p = {"uid": uid, "name": name or uid, "namespace": namespace, "labels": dict(labels or {}), "requests": dict(requests or {}),
         "creationTimestamp": creation, "phase": phase, "nodeName": node_name}
    if node_selector:
        p["nodeSelector"] = dict(node_selector)
    if volume_requirements:
        p["volumeRequirements"] = [list(alt) for alt in volume_requirements]
    if volumes:   # scheduling.GetVolumes(pod) (volumeusage.go:83-114): (CSI driver, namespace/name of the PVC) per tracked volume
        p["volumes"] = [{"driver": d, "pvc": c} for d, c in volumes]
    if host_ports:
        p["hostPorts"] = [host_port(h) if not isinstance(h, dict) else host_port(**h) for h in host_ports]
    if node_requirements or node_preferences:
        na = {}
        if node_requirements:
            terms = node_requirements if isinstance(node_requirements[0], list) else [node_requirements]
            na["required"] = terms
        if node_preferences:
            prefs = node_preferences
            if prefs and "matchExpressions" not in prefs[0]:
                prefs = [{"weight": 1, "matchExpressions": prefs}]
            na["preferred"] = prefs
        p["nodeAffinity"] = na
    if tolerations:
        p["tolerations"] = [dict({"key": "", "operator": "", "value": "", "effect": ""}, **t) for t in tolerations]
    if topology_spread:
        p["topologySpreadConstraints"] = topology_spread
    if pod_requirements or pod_preferences:
        p["podAffinity"] = {"required": pod_requirements or [], "preferred": pod_preferences or []}
    if pod_anti_requirements or pod_anti_preferences:
        p["podAntiAffinity"] = {"required": pod_anti_requirements or [], "preferred": pod_anti_preferences or []}
    return p


def host_port(port, ip="", protocol="TCP"):
    """corev1.ContainerPort with a HostPort (scheduling.GetHostPorts, hostportusage.go:93-117: hostIP "" reads 0.0.0.0)."""
    return {"port": int(port), "ip": ip, "protocol": protocol}


def spread(key, labels, max_skew=1, when="DoNotSchedule", min_domains=None, taints_policy=None, affinity_policy=None):
    c = {"maxSkew": max_skew, "topologyKey": key, "whenUnsatisfiable": when, "labelSelector": {"matchLabels": dict(labels)}}
    if min_domains is not None:
        c["minDomains"] = min_domains
    if taints_policy:
        c["nodeTaintsPolicy"] = taints_policy
    if affinity_policy:
        c["nodeAffinityPolicy"] = affinity_policy
    return c


def affinity_term(key, labels, namespaces=None, namespace_selector=None):
    """corev1.PodAffinityTerm; namespace_selector is a metav1.LabelSelector dict ({} selects every namespace)."""
    t = {"labelSelector": {"matchLabels": dict(labels)}, "topologyKey": key}
    if namespaces:
        t["namespaces"] = list(namespaces)
    if namespace_selector is not None:
        t["namespaceSelector"] = dict(namespace_selector)
    return t


def weighted(weight, term):
    """WeightedPodAffinityTerm."""
    return {"weight": weight, "term": term}


def node_pool(name="default", weight=0, requirements=None, labels=None, taints=None, limits=None, instance_types=None):
    """test.NodePool — pkg/test/nodepool.go:35 (node class label as the test fixtures produce it)."""
    np = {"name": name, "weight": weight, "requirements": list(requirements or []), "labels": dict(labels or {}),
          "taints": [dict({"key": "", "value": "", "effect": ""}, **t) for t in (taints or [])],
          "nodeClassLabelKey": "karpenter.test.sh/testnodeclass", "nodeClassName": "default"}
    if limits is not None:
        np["limits"] = dict(limits)
    if instance_types is not None:
        np["instanceTypes"] = list(instance_types)
    return np


KNOWN_EPHEMERAL_TAINTS = [("node.kubernetes.io/not-ready", "NoSchedule"), ("node.kubernetes.io/not-ready", "NoExecute"),
                          ("node.kubernetes.io/unreachable", "NoSchedule"), ("node.cloudprovider.kubernetes.io/uninitialized", "NoSchedule"),
                          ("karpenter.sh/unregistered", "NoExecute")]          # scheduling/taints.go:38-44 (MatchTaint: key + effect)
KNOWN_EPHEMERAL_TAINT_PREFIXES = ["readiness.k8s.io/"]                        # taints.go:49-52


def state_node_taints(taints, startup_taints=None, initialized=True, managed=True):
    """StateNode.Taints() — state/statenode.go:311-339: until a managed node is initialized its well-known ephemeral
    taints and the NodeClaim's startup taints are not held against pods (they are expected to go away)."""
    taints = [dict({"key": "", "value": "", "effect": ""}, **t) for t in (taints or [])]
    if initialized or not managed:
        return taints
    startup = {(t["key"], t["effect"]) for t in (startup_taints or [])}

    def ephemeral(t):
        return (t["key"], t["effect"]) in KNOWN_EPHEMERAL_TAINTS or any(t["key"].startswith(p) for p in KNOWN_EPHEMERAL_TAINT_PREFIXES)
    return [t for t in taints if not ephemeral(t) and (t["key"], t["effect"]) not in startup]


def state_node(name, instance_type, zone, capacity_type="on-demand", nodepool="default", used=None, taints=None, initialized=True,
               extra_labels=None, under_consolidate_after=False, startup_taints=None, host_ports=None, volumes=None, volume_limits=None):
    """A state.StateNode as the scheduler reads it (existingnode.go:47-75): labels of a node launched from `instance_type`
    in `zone` (single-valued instance-type requirements become labels, like the fake/KWOK providers do on Create),
    Available() = allocatable - used, Capacity() incl. nodes: 1 (statenode.go:370-374)."""
    labels = {}
    for r in instance_type["requirements"]:
        if r["operator"] == "In" and len(r["values"]) == 1:
            labels[r["key"]] = r["values"][0]
    labels.update({ZONE: zone, CAPACITY_TYPE: capacity_type, NODEPOOL: nodepool, HOSTNAME: name, "karpenter.sh/registered": "true"})
    if initialized:
        labels["karpenter.sh/initialized"] = "true"
    labels.update(extra_labels or {})
    from decimal import Decimal
    def nano(q):
        m = re.fullmatch(r"([0-9.]+)([a-zA-Z]*)", str(q))
        mult = {"": 1, "m": Decimal("0.001"), "k": 10**3, "M": 10**6, "G": 10**9, "T": 10**12, "Ki": 2**10, "Mi": 2**20, "Gi": 2**30, "Ti": 2**40}[m.group(2)]
        return int(Decimal(m.group(1)) * mult * 10**9)
    avail = {}
    for k, v in instance_type["capacity"].items():
        a = nano(v) - nano(instance_type["overhead"].get(k, "0")) - nano((used or {}).get(k, "0"))
        avail[k] = f"{a}n"
    cap = dict(instance_type["capacity"]); cap["nodes"] = "1"
    node = {"name": name, "labels": labels, "taints": state_node_taints(taints, startup_taints, initialized, managed=True),
            "available": avail, "capacity": cap, "initialized": initialized, "managed": True, "underConsolidateAfter": under_consolidate_after}
    if volumes or volume_limits:   # StateNode.VolumeUsage() (statenode.go:411,466-490): volumes of the bound pods, the CSINode's attach limits
        node["volumeUsage"] = {"volumes": [{"driver": d, "pvc": c} for d, c in (volumes or [])], "limits": dict(volume_limits or {})}
    if host_ports:   # StateNode.HostPortUsage(): host ports of the pods bound to the node (statenode.go:489)
        node["hostPorts"] = [host_port(h) if not isinstance(h, dict) else host_port(**h) for h in host_ports]
    return node


_M64 = (1 << 64) - 1


def _splitmix64(x):
    x = (x + 0x9E3779B97F4A7C15) & _M64
    z = x
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
    return x, z ^ (z >> 31)


def group_pod_uid(seed, i):
    """uid of pod i of a podGroup — the same 128 bits the host library and the oracle derive (ksched.cpp group_uid)."""
    st = (seed * 0x9E3779B97F4A7C15 + i * 0xD1B54A32D192ED03 + 0x2545F4914F6CDD1D) & _M64
    st, a = _splitmix64(st)
    st, b = _splitmix64(st)
    return "%08x-%04x-%04x-%04x-%012x" % (a >> 32, (a >> 16) & 0xFFFF, a & 0xFFFF, b >> 48, b & 0xFFFFFFFFFFFF)


def expand_pod_groups(problem):
    """The same problem with every podGroup written out as explicit pods (for exporting a BASELINE configuration to the
    reference's wire shapes, tests/golden/from_go.py:to_go)."""
    out = dict(problem)
    pods = list(problem.get("pods", []))
    for g in problem.get("podGroups", []):
        for i in range(g["count"]):
            uid = group_pod_uid(g.get("uidSeed", 0), i)
            pods.append(dict(g["template"], uid=uid, name=uid))
    out["pods"], out["podGroups"] = pods, []
    return out


def launch(results, instance_types, pods, name_prefix="node"):
    """What happens between two provisioning passes in the reference's tests (ExpectProvisioned + node state reconcile):
    every NodeClaim of `results` is created by the cloud provider — cheapest instance type option, cheapest available
    offering its requirements admit (fake/cloudprovider.go:108-170) — and becomes an initialized state node with the
    claim's pods bound to it. Returns (state_nodes, bound_pods): feed them to the next problem as state_nodes /
    cluster_pods. Pods placed on existing nodes are returned as bound to those nodes."""
    by_name = {t["name"]: t for t in instance_types}
    by_uid = {p["uid"]: p for p in pods}

    def has(r, value):
        if r.get("gte") is not None or r.get("lte") is not None:
            try:
                v = int(value)
            except ValueError:
                return False
            if (r.get("gte") is not None and v < r["gte"]) or (r.get("lte") is not None and v > r["lte"]):
                return False
        return (value not in r["values"]) if r["complement"] else (value in r["values"])

    nodes, bound = [], []
    for i, c in enumerate(results["newNodeClaims"]):
        reqs = {r["key"]: r for r in c["requirements"]}
        best = None
        for n in c["instanceTypes"]:
            for o in by_name[n]["offerings"]:
                if not o.get("available", True):
                    continue
                if all(r["key"] not in reqs or has(reqs[r["key"]], r["values"][0]) for r in o["requirements"]):
                    if best is None or (o["price"], n) < (best[0]["price"], best[1]):
                        best = (o, n)
        assert best is not None, "a NodeClaim without a launchable offering"
        off, it_name = best
        labels = {r["key"]: r["values"][0] for r in off["requirements"]}
        name = f"{name_prefix}-{i:04d}"
        members = [by_uid[u] for u in c["pods"]]
        used = {}
        for p in members:
            for k, v in p["requests"].items():
                used[k] = used.get(k, 0.0) + quantity_float(v)
        used_q = {k: f"{int(round(v * 1000))}m" for k, v in used.items()}
        used_q["pods"] = str(len(members))
        extra = {r["key"]: r["values"][0] for r in c["requirements"] if not r["complement"] and len(r["values"]) == 1 and r["key"] != HOSTNAME}
        ports = [h for p in members for h in p.get("hostPorts", [])]
        node = state_node(name, by_name[it_name], labels[ZONE], labels[CAPACITY_TYPE], c["nodePool"], used=used_q, extra_labels=extra, host_ports=ports)
        nodes.append(node)
        for p in members:
            bound.append(dict(p, phase="Running", nodeName=name))
    for e in results.get("existingNodes", []):
        for u in e["pods"]:
            bound.append(dict(by_uid[u], phase="Running", nodeName=e["name"]))
    return nodes, bound


def problem(instance_types, node_pools, pods=None, pod_groups=None, well_known=FAKE_WELL_KNOWN, state_nodes=None, cluster_pods=None,
            daemonset_pods=None, options=None, deleting_node_names=None, namespaces=None):
    """`namespaces`: [{"name", "labels"}] — what the namespace lister returns, for affinity terms with a namespaceSelector
    (topology.go:536-557)."""
    return {"namespaces": list(namespaces or []), "wellKnownLabels": list(well_known), "options": dict(options or {}), "instanceTypes": instance_types, "nodePools": node_pools,
            "stateNodes": list(state_nodes or []), "pods": list(pods or []), "podGroups": list(pod_groups or []),
            "daemonSetPods": list(daemonset_pods or []), "clusterPods": list(cluster_pods or []),
            "deletingNodeNames": list(deleting_node_names or [])}


# ---------------------------------------------------------------------------------------------------------------
# BASELINE.json configurations. Pods are emitted as podGroups: {count, uidSeed, template}; the host
# library (and the oracle) expand a group into `count` pods whose uids come from splitmix64(uidSeed, i), so a
# million-pod problem is a few hundred JSON objects. Because queue order breaks (cpu, memory) ties by uid
# (queue.go:98-107), pods of different groups interleave pseudo-randomly exactly like uuid.NewUUID() pods do in
# scheduling_benchmark_test.go.
# ---------------------------------------------------------------------------------------------------------------
BENCH_CPU_M = [100, 250, 500, 1000, 1500]        # scheduling_benchmark_test.go:451-455
BENCH_MEM_MI = [100, 256, 512, 1024, 2048, 4096]  # :447-450


def _split_counts(total, weights, rng):
    """Deterministic multinomial-ish split of `total` over len(weights) groups (largest remainder)."""
    s = float(sum(weights))
    raw = [total * w / s for w in weights]
    base = [int(math.floor(x)) for x in raw]
    rem = total - sum(base)
    order = sorted(range(len(weights)), key=lambda i: (raw[i] - base[i], i), reverse=True)
    for i in order[:rem]:
        base[i] += 1
    return base


def config1(pods=5000, n_types=50, seed=42):
    """C1: KWOK provider, cpu+mem requests only, 50 instance types, 1 NodePool."""
    rng = random.Random(seed)
    combos = [(c, m) for c in BENCH_CPU_M for m in BENCH_MEM_MI]
    counts = _split_counts(pods, [1 + rng.random() for _ in combos], rng)
    groups = []
    for gi, ((c, m), n) in enumerate(zip(combos, counts)):
        if n:
            groups.append({"count": n, "uidSeed": seed * 100003 + gi, "template": pod(uid="t", requests={"cpu": f"{c}m", "memory": f"{m}Mi"}, labels={"my-label": "abcdefg"[gi % 7]})})
    np_ = node_pool("default", limits={"cpu": "10000000", "memory": "10000000Gi"})
    np_["nodeClassLabelKey"] = "karpenter.kwok.sh/kwoknodeclass"
    return problem(kwok_catalog(n_types), [np_], pod_groups=groups, well_known=KWOK_WELL_KNOWN)


def config2(pods=1_000_000, n_types=500, seed=42, tolerating_fraction=0.25):
    """C2: nodeSelector + taint/toleration constraints, 500 instance types.

    Two NodePools: `dedicated` (weight 10, taint dedicated=batch:NoSchedule) and `default` (weight 0, untainted).
    A quarter of the pods tolerate the taint (Exists) and therefore try `dedicated` first. nodeSelectors are drawn
    from the instance-type universe: arch, os, zone, capacity-type, each either unset or one value.
    """
    rng = random.Random(seed)
    combos = [(c, m) for c in BENCH_CPU_M for m in BENCH_MEM_MI]
    selectors = [{}]
    for arch in ("amd64", "arm64"):
        selectors.append({ARCH: arch})
    for z in KWOK_ZONES:
        selectors.append({ZONE: z})
    selectors.append({CAPACITY_TYPE: "spot"})
    selectors.append({CAPACITY_TYPE: "on-demand"})
    for arch in ("amd64", "arm64"):
        for z in KWOK_ZONES[:2]:
            selectors.append({ARCH: arch, ZONE: z})
    selectors.append({OS: "linux", ARCH: "arm64", CAPACITY_TYPE: "spot"})
    selectors.append({OS: "linux", ZONE: KWOK_ZONES[3], CAPACITY_TYPE: "on-demand"})
    classes = []
    for (c, m) in combos:
        for si, sel in enumerate(selectors):
            for tol in (False, True):
                w = (3.0 if not sel else 1.0) * (tolerating_fraction if tol else 1 - tolerating_fraction) * (1 + rng.random())
                classes.append((c, m, si, sel, tol, w))
    counts = _split_counts(pods, [k[5] for k in classes], rng)
    groups = []
    for gi, ((c, m, si, sel, tol, _), n) in enumerate(zip(classes, counts)):
        if not n:
            continue
        sel2 = dict(sel)
        if not tol and OS not in sel2 and si % 3 == 0:
            sel2[OS] = "linux"
        t = pod(uid="t", requests={"cpu": f"{c}m", "memory": f"{m}Mi"}, node_selector=sel2,
                tolerations=[{"key": "dedicated", "operator": "Exists", "effect": "NoSchedule"}] if tol else None)
        groups.append({"count": n, "uidSeed": seed * 100003 + gi, "template": t})
    dedicated = node_pool("dedicated", weight=10, taints=[{"key": "dedicated", "value": "batch", "effect": "NoSchedule"}],
                          requirements=[req(OS, "In", "linux")])
    default = node_pool("default", weight=0)
    for np_ in (dedicated, default):
        np_["nodeClassLabelKey"] = "karpenter.kwok.sh/kwoknodeclass"
    return problem(kwok_catalog(n_types), [dedicated, default], pod_groups=groups, well_known=KWOK_WELL_KNOWN)


def config4(pods=10_000_000, n_types=1000, n_pools=16, seed=42):
    """C4: 16 NodePools with distinct weights; every pod pins its NodePool with a `karpenter.sh/nodepool` node selector, so
    the pools are independent components of one provisioning pass. Pod shapes as in C2 (cpu x memory grid,
    arch / zone / capacity-type selectors). karpenter_amd/components.py splits it, one sub-problem per pool."""
    rng = random.Random(seed)
    combos = [(c, m) for c in BENCH_CPU_M for m in BENCH_MEM_MI]
    selectors = [{}, {ARCH: "amd64"}, {ARCH: "arm64"}, {CAPACITY_TYPE: "spot"}, {CAPACITY_TYPE: "on-demand"}] + [{ZONE: z} for z in KWOK_ZONES]
    pools = []
    for i in range(n_pools):
        np_ = node_pool(f"pool-{i:02d}", weight=n_pools - i)
        np_["nodeClassLabelKey"] = "karpenter.kwok.sh/kwoknodeclass"
        pools.append(np_)
    classes = [(c, m, sel, i, 1 + rng.random()) for (c, m) in combos for sel in selectors for i in range(n_pools)]
    counts = _split_counts(pods, [k[4] for k in classes], rng)
    groups = []
    for gi, ((c, m, sel, i, _), n) in enumerate(zip(classes, counts)):
        if n:
            t = pod(uid="t", requests={"cpu": f"{c}m", "memory": f"{m}Mi"}, node_selector=dict(sel, **{NODEPOOL: pools[i]["name"]}))
            groups.append({"count": n, "uidSeed": seed * 100003 + gi, "template": t})
    return problem(kwok_catalog(n_types), pools, pod_groups=groups, well_known=KWOK_WELL_KNOWN)


def config3(pods=1_000_000, n_types=500, seed=42, anti_affinity_pods=None):
    """C3: podAntiAffinity + 3-zone topologySpreadConstraints, the reference benchmark's diverse mix
    (scheduling_benchmark_test.go:259-272): one fifth each of generic pods, zonal spread (maxSkew 1), hostname spread,
    zonal self-affinity and hostname anti-affinity (app=nginx pods repel each other: one node each), labels and spread
    selectors drawn from my-label in a..g (randomLabels, :430-436), NodePool limited to three zones.
    anti_affinity_pods overrides the size of the anti-affinity fifth (every such pod needs its own NodeClaim)."""
    rng = random.Random(seed)
    combos = [(c, m) for c in BENCH_CPU_M for m in BENCH_MEM_MI]
    letters = "abcdefg"
    fifth = pods // 5
    n_anti = fifth if anti_affinity_pods is None else min(anti_affinity_pods, pods)
    rest = pods - n_anti
    kinds = []   # (weight, template kwargs)
    for lab in letters:
        kinds.append(("generic", 1.0 + pods % 5, dict(labels={"my-label": lab})))
        for sel in letters:
            kinds.append(("zonal", 1.0 / 7, dict(labels={"my-label": lab}, topology_spread=[spread(ZONE, {"my-label": sel})])))
            kinds.append(("host", 1.0 / 7, dict(labels={"my-label": lab}, topology_spread=[spread(HOSTNAME, {"my-label": sel})])))
        kinds.append(("affinity", 1.0, dict(labels={"my-affininity": lab}, pod_requirements=[affinity_term(ZONE, {"my-affininity": lab})])))
    classes = [(k, w * (1 + 0.2 * rng.random()), kw, c, m) for (k, w, kw) in kinds for (c, m) in combos]
    counts = _split_counts(rest, [k[1] for k in classes], rng)
    groups = []
    gi = 0
    for (kind, _, kw, c, m), n in zip(classes, counts):
        gi += 1
        if n:
            groups.append({"count": n, "uidSeed": seed * 100003 + gi, "template": pod(uid="t", requests={"cpu": f"{c}m", "memory": f"{m}Mi"}, **kw)})
    nginx = {"app": "nginx"}
    for (c, m), n in zip(combos, _split_counts(n_anti, [1 + rng.random() for _ in combos], rng)):
        gi += 1
        if n:
            groups.append({"count": n, "uidSeed": seed * 100003 + gi,
                           "template": pod(uid="t", requests={"cpu": f"{c}m", "memory": f"{m}Mi"}, labels=nginx, pod_anti_requirements=[affinity_term(HOSTNAME, nginx)])})
    np_ = node_pool("default", requirements=[req(ZONE, "In", *KWOK_ZONES[:3])])
    np_["nodeClassLabelKey"] = "karpenter.kwok.sh/kwoknodeclass"
    return problem(kwok_catalog(n_types), [np_], pod_groups=groups, well_known=KWOK_WELL_KNOWN)


def scale_problem(prob, pods):
    """Same problem with the pod groups rescaled to `pods` total (keeps the class mix)."""
    p = copy.deepcopy(prob)
    total = sum(g["count"] for g in p["podGroups"])
    counts = _split_counts(pods, [g["count"] for g in p["podGroups"]], None)
    for g, n in zip(p["podGroups"], counts):
        g["count"] = n
    p["podGroups"] = [g for g in p["podGroups"] if g["count"]]
    assert sum(g["count"] for g in p["podGroups"]) == pods, total
    return p
