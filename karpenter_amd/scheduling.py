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
        lib.ksched_open.restype = ctypes.c_void_p
        lib.ksched_open.argtypes = [ctypes.c_char_p, ctypes.c_char_p]
        lib.ksched_solve.restype = ctypes.c_void_p
        lib.ksched_solve.argtypes = [ctypes.c_void_p, ctypes.c_int]
        lib.ksched_error.restype = ctypes.c_char_p
        lib.ksched_error.argtypes = [ctypes.c_void_p]
        lib.ksched_error_kind.restype = ctypes.c_char_p
        lib.ksched_error_kind.argtypes = [ctypes.c_void_p]
        lib.ksched_close.argtypes = [ctypes.c_void_p]
        lib.ksched_cancel.restype = ctypes.c_int
        lib.ksched_cancel.argtypes = [ctypes.c_void_p]
        lib.ksched_free.argtypes = [ctypes.c_void_p]
        lib.ksched_solve_batch.restype = ctypes.c_int
        lib.ksched_solve_batch.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]
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


def _raise(kind, msg):
    if kind == "unsupported":
        raise Unsupported(msg)
    if kind in ("no_device", "load"):
        raise SolverUnavailable(msg)
    raise RuntimeError(f"ksolve {kind}: {msg}")


class Scheduler:
    """scheduling.Scheduler: the problem flattened and resident in HBM (ksolve_create); Solve() runs the hot path."""

    def __init__(self, problem: dict, solver_lib: str | None = None):
        self.problem = problem
        self._solver_lib = solver_lib or KSOLVE_LIB
        if not os.path.exists(self._solver_lib):
            raise SolverUnavailable(f"{self._solver_lib} is not built (hipcc --offload-arch=gfx950; see __graft_entry__.build)")
        self._lib = _load_ksched()
        self._session = self._lib.ksched_open(json.dumps(problem).encode(), self._solver_lib.encode())
        err = self._lib.ksched_error(self._session)
        if err is not None:
            kind = self._lib.ksched_error_kind(self._session).decode()
            msg = err.decode()
            self.close()
            _raise(kind, msg)

    def close(self):
        if getattr(self, "_session", None):
            self._lib.ksched_close(self._session)
            self._session = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def Solve(self, repeat: int = 1, want_results: bool = True) -> Results:
        """Runs Solve() on the device; `repeat` re-solves on the same resident inputs and collects every run's timings."""
        if not self._session:
            raise RuntimeError("scheduler is closed")
        timings = []
        out = None
        for _ in range(max(1, int(repeat))):
            ptr = self._lib.ksched_solve(self._session, 1 if want_results else 0)
            try:
                out = json.loads(ctypes.string_at(ptr).decode())
            finally:
                self._lib.ksched_free(ptr)
            if "error" in out:
                _raise(out.get("kind"), out["error"])
            timings += out["timings"]
        out["timings"] = timings
        return Results(out)


    def Cancel(self) -> None:
        """The ctx deadline of Solve (scheduler.go:477-480, provisioner.go:427): call from another thread while Solve()
        runs; Solve returns the pods placed so far with `timedOut` set. A cancel that arrives between solves is dropped
        (every Solve starts with a fresh context)."""
        if self._session:
            self._lib.ksched_cancel(self._session)


def NewScheduler(problem: dict, solver_lib: str | None = None) -> Scheduler:
    return Scheduler(problem, solver_lib)


def SolveBatch(schedulers, want_results: bool = True):
    """Solve() for many independent Schedulers with ONE launch of the pack kernel (ksolve_solve_batch): each problem is
    its own wavefront on its own CU. Same Results as calling Solve() on each; this is how a consolidation pass
    (disruption/helpers.go:53-155: one simulation per candidate set) fills the GPU. NodePool components of ONE pass are
    not the same thing: solved separately they give an equally good, not the identical, packing (components.py)."""
    schedulers = list(schedulers)
    if not schedulers:
        return []
    lib = schedulers[0]._lib
    n = len(schedulers)
    sessions = (ctypes.c_void_p * n)(*[s._session for s in schedulers])
    outs = (ctypes.c_void_p * n)()
    lib.ksched_solve_batch(sessions, n, 1 if want_results else 0, outs)
    results = []
    for i in range(n):
        try:
            doc = json.loads(ctypes.string_at(outs[i]).decode())
        finally:
            lib.ksched_free(outs[i])
        if "error" in doc:
            _raise(doc.get("kind"), doc["error"])
        results.append(Results(doc))
    return results


def device_available() -> bool:
    if not os.path.exists(KSOLVE_LIB):
        return False
    try:
        lib = ctypes.CDLL(KSOLVE_LIB)
        lib.ksolve_device_available.restype = ctypes.c_int
        return bool(lib.ksolve_device_available())
    except OSError:
        return False
