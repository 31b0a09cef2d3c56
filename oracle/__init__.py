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


def sweep(problem: dict, probes: list, threads: int = 1) -> list:
    """SimulateScheduling for many candidate sets of one cluster (oracle_api.cpp: oracle_sweep_json): `problem` = the cluster as
    a problem document, `probes` = [{"removeNodes": [names], "pods": [pod documents]}]; one Results document per probe.
    The simulations are independent: `threads` of them run at a time."""
    return _call(_lib().oracle_sweep_json, {"problem": problem, "probes": probes, "threads": int(threads)})["results"]


def evaluate(query: dict):
    """Unit-level algebra probe (see oracle_api.cpp: oracle_eval_json)."""
    return _call(_lib().oracle_eval_json, query)["result"]
