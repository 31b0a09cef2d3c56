"""Python host mirror of the reference's scheduler interface for the one path this repository accelerates:

    scheduling.NewScheduler(...)      pkg/controllers/provisioning/scheduling/scheduler.go:127
    (*Scheduler).Solve(ctx, pods)     pkg/controllers/provisioning/scheduling/scheduler.go:440
    Results{NewNodeClaims, ExistingNodes, PodErrors}                       scheduler.go:281

`NewScheduler(problem)` takes the problem document built by karpenter_amd.fixtures (the same inputs
Provisioner.NewScheduler assembles, provisioner.go:265-360) and `Scheduler.Solve()` runs it on the MI355X through the
C ABI (include/ksolve.h): karpenter_amd/libksched.so (host flattener, C++) -> karpenter_amd/libksolve.so (HIP).

There is no CPU path: if the HIP library is missing or no gfx950 device is usable, Solve raises.
"""
from __future__ import annotations

import ctypes
import json
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
KSCHED_LIB = os.path.join(_HERE, "libksched.so")
KSOLVE_LIB = os.path.join(_HERE, "libksolve.so")

POD_ERRORS = {
    0: "ok", 1: "did not tolerate taint", 2: "incompatible requirements", 3: "unsatisfiable topology constraint",
    4: "no instance type met the requirements/resources/offering tuple", 5: "exceeds node resources",
    6: "nodepool requirements filtered out all available instance types", 7: "nodepool limits",
    8: "reserved offering error", 9: "failed scheduling pod to existing nodes", 10: "minValues requirement is not met",
}


class SolverUnavailable(RuntimeError):
    """The HIP solver library is not built, or there is no usable gfx950 device."""


class Unsupported(RuntimeError):
    """Valid Karpenter input that this build does not solve on the device (it is never solved on the CPU instead)."""


_ksched = None


def _load_ksched():
    global _ksched
    if _ksched is None:
        if not os.path.exists(KSCHED_LIB):
            raise SolverUnavailable(f"{KSCHED_LIB} is not built; run `python -c 'import __graft_entry__ as g; g.build()'`")
        lib = ctypes.CDLL(KSCHED_LIB)
        lib.ksched_solve_json.restype = ctypes.c_void_p
        lib.ksched_solve_json.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_int]
        lib.ksched_free.argtypes = [ctypes.c_void_p]
        _ksched = lib
    return _ksched


class Results(dict):
    """scheduling.Results (scheduler.go:281-286) as a dict: newNodeClaims, existingNodes, podErrors, counters, timings."""

    @property
    def new_node_claims(self):
        return self["newNodeClaims"]

    @property
    def pod_errors(self):
        return self["podErrors"]

    def all_pods_scheduled(self):
        return not self["podErrors"]


class Scheduler:
    def __init__(self, problem: dict, solver_lib: str | None = None):
        self.problem = problem
        self._solver_lib = solver_lib or KSOLVE_LIB

    def Solve(self, repeat: int = 1, want_results: bool = True) -> Results:
        """Runs Solve() on the device. `repeat` re-solves on the same handle (inputs stay resident in HBM)."""
        if not os.path.exists(self._solver_lib):
            raise SolverUnavailable(f"{self._solver_lib} is not built (hipcc --offload-arch=gfx950; see __graft_entry__.build)")
        lib = _load_ksched()
        ptr = lib.ksched_solve_json(json.dumps(self.problem).encode(), self._solver_lib.encode(), int(repeat), 1 if want_results else 0)
        try:
            out = json.loads(ctypes.string_at(ptr).decode())
        finally:
            lib.ksched_free(ptr)
        if "error" in out:
            kind = out.get("kind")
            if kind == "unsupported":
                raise Unsupported(out["error"])
            if kind in ("no_device", "load"):
                raise SolverUnavailable(out["error"])
            raise RuntimeError(f"ksolve {kind}: {out['error']}")
        return Results(out)


def NewScheduler(problem: dict, solver_lib: str | None = None) -> Scheduler:
    return Scheduler(problem, solver_lib)


def device_available() -> bool:
    if not os.path.exists(KSOLVE_LIB):
        return False
    try:
        lib = ctypes.CDLL(KSOLVE_LIB)
        lib.ksolve_device_available.restype = ctypes.c_int
        return bool(lib.ksolve_device_available())
    except OSError:
        return False
