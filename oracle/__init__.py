"""ORACLE — TEST INFRASTRUCTURE ONLY.

ctypes client for oracle/liboracle.so, the CPU restatement of the reference Solve() path. Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package; the product (karpenter_amd/) never does.
"""
import ctypes
import json
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".hpp", ".cpp"))]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])
    return so


def _lib():
    global _LIB
    if _LIB is None:
        lib = ctypes.CDLL(build())
        for fn in (lib.oracle_solve_json, lib.oracle_eval_json, lib.oracle_sweep_json):
            fn.restype = ctypes.c_void_p
            fn.argtypes = [ctypes.c_char_p]
        lib.oracle_free.argtypes = [ctypes.c_void_p]
        _LIB = lib
    return _LIB


def _call(fn, doc):
    lib = _lib()
    ptr = fn(json.dumps(doc).encode())
    try:
        out = json.loads(ctypes.string_at(ptr).decode())
    finally:
        lib.oracle_free(ptr)
    if isinstance(out, dict) and "error" in out and len(out) == 1:
        raise RuntimeError("oracle: " + out["error"])
    return out


def solve(problem: dict) -> dict:
    """Reference-semantics Solve() on the CPU. Returns the results document."""
    return _call(_lib().oracle_solve_json, problem)


def sweep(problem: dict, probes: list, threads: int = 1, verdicts: bool = False, multi_node: bool = False) -> list:
    """SimulateScheduling for many candidate sets of one cluster (oracle_api.cpp: oracle_sweep_json): `problem` = the cluster as
    a problem document, `probes` = [{"removeNodes": [names], "pods": [pod documents]}]; one Results document per probe.
    The simulations are independent: `threads` of them run at a time. verdicts: every document also carries "verdict", the
    command computeConsolidation (consolidation.go:159-256) derives from that simulation, restated in oracle/consolidation.hpp
    — with multi_node the command of one step of firstNConsolidationOption (filterOutSameInstanceType applied to a replacement)."""
    return _call(_lib().oracle_sweep_json, {"problem": problem, "probes": probes, "threads": int(threads), "verdicts": bool(verdicts), "multiNode": bool(multi_node)})["results"]


def verdict_key(v: dict):
    """A verdict of sweep(..., verdicts=True) in the form the callers compare: (decision, replacement instance types by name, "spot"
    when consolidation.go:238-243 pinned the replacement to spot)."""
    return (v["decision"], sorted(v["replacement"]) if v["replacement"] is not None else None, "spot" if v["pinnedToSpot"] and v["decision"] == "replace" else None)


def cluster_verdicts(cluster: dict, candidate_sets: list, multi_node: bool = False, threads: int = 1, well_known=()) -> list:
    """computeConsolidation for candidate sets of a cluster document of the test fixtures ({"nodes": [node with "pods"],
    "pendingPods", "instanceTypes", "nodePools", "options", ...}): assembles what SimulateScheduling assembles (helpers.go:53-155:
    the active nodes without the candidates; pending pods + the candidates' pods + the pods of deleting nodes) and returns one
    verdict per set. Everything downstream of the document is the oracle's own code."""
    nodes = cluster["nodes"]
    deleting = [n for n in nodes if n.get("markedForDeletion")]
    state_nodes = [{k: v for k, v in n.items() if k != "pods"} for n in nodes if not n.get("markedForDeletion")]
    bound = [p for n in nodes if not n.get("markedForDeletion") for p in n.get("pods", [])]
    options = dict(cluster.get("options", {}), consolidationSimulation=True, truncateInstanceTypes=600)   # scheduling.MaxInstanceTypes, helpers.go:131
    base = {"options": options, "wellKnownLabels": list(cluster.get("wellKnownLabels", well_known)), "instanceTypes": cluster["instanceTypes"], "nodePools": cluster["nodePools"],
            "stateNodes": state_nodes, "pods": [], "podGroups": [], "daemonSetPods": cluster.get("daemonSetPods", []), "clusterPods": bound,
            "deletingNodeNames": [n["name"] for n in deleting], "namespaces": cluster.get("namespaces") or []}
    probes = []
    for cands in candidate_sets:
        names = [c["name"] for c in cands]
        pods = list(cluster.get("pendingPods", []))
        for c in cands:
            pods += c.get("pods", [])
        pods += [p for n in deleting if n["name"] not in names for p in n.get("pods", [])]
        probes.append({"removeNodes": names, "pods": pods})
    return [r["verdict"] for r in sweep(base, probes, threads=threads, verdicts=True, multi_node=multi_node)]


def evaluate(query: dict):
    """Unit-level algebra probe (see oracle_api.cpp: oracle_eval_json)."""
    return _call(_lib().oracle_eval_json, query)["result"]
