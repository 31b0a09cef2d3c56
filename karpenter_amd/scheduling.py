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
        lib.ksched_probe.restype = ctypes.c_void_p
        lib.ksched_probe.argtypes = [ctypes.c_void_p, ctypes.c_char_p]
        lib.ksched_sweep.restype = ctypes.c_void_p
        lib.ksched_sweep.argtypes = [ctypes.c_void_p, ctypes.c_char_p]
        lib.ksched_sweep_replicas.restype = ctypes.c_void_p
        lib.ksched_sweep_replicas.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, ctypes.c_char_p]
        lib.ksched_sweep_arrays.restype = ctypes.c_void_p
        lib.ksched_sweep_arrays.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 7 + [ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32]
        lib.ksched_pods_by_claim.restype = ctypes.c_uint32
        lib.ksched_pods_by_claim.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32]
        lib.ksched_split_components.restype = ctypes.c_void_p
        lib.ksched_split_components.argtypes = [ctypes.c_char_p, ctypes.c_int]
        lib.ksched_assignment.restype = ctypes.c_uint32
        lib.ksched_assignment.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32]
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


def _want(want_results):
    # "claims": the NodeClaims without their pod lists; "claims-compact": ... and with instance type options as positions in the
    # problem's instanceTypes list ("instanceTypeIndices") — both leave the pods to Scheduler.Assignment / PodsByClaim
    return 3 if want_results == "claims-compact" else 2 if want_results == "claims" else (1 if want_results else 0)


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
        # solver_lib is a TEST HOOK (the host emulation of the engine, the gfx950 build with test switches): the product loads its own
        # libksolve.so and nothing else; the tests' conftest.py opens the gate
        if solver_lib and os.environ.get("KSOLVE_TEST_SOLVER_LIB") != "1":
            raise SolverUnavailable("solver_lib is a test hook (set KSOLVE_TEST_SOLVER_LIB=1); the product has no CPU path and one device library")
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
        for p in getattr(self, "_probes", []):   # probes share this session's tables: they go first
            p.close()
        self._probes = []
        if getattr(self, "_session", None):
            self._lib.ksched_close(self._session)
            self._session = None

    def Probe(self, remove_nodes=(), pods=(), pods_of_removed_nodes=False) -> "Scheduler":
        """One probe of a RESIDENT cluster (ksolve_probe_create; disruption/helpers.go:53-155): this scheduler holds the
        whole cluster — every node as a state node, every pod some probe may place as a pod — and the probe is that
        cluster without `remove_nodes` (names), scheduling `pods` (uids). The returned Scheduler shares the tables already
        on the device (nothing is flattened or uploaded again), solves alone or through SolveBatch, and is closed with
        this one. Raises Unsupported for clusters whose pods carry topology constraints (one NewScheduler per probe then).
        pods_of_removed_nodes=True: the probe schedules what a sweep schedules for these candidates — the pods of every simulation and
        the pods bound to the removed nodes (a resident cluster whose pods travel as groups has no uids to list them by)."""
        if not self._session:
            raise RuntimeError("scheduler is closed")
        doc = json.dumps({"removeNodes": list(remove_nodes), "pods": list(pods), "podsOfRemovedNodes": bool(pods_of_removed_nodes)}).encode()
        probe = object.__new__(Scheduler)
        probe.problem, probe._solver_lib, probe._lib, probe._probes = None, self._solver_lib, self._lib, []
        probe._session = self._lib.ksched_probe(self._session, doc)
        err = self._lib.ksched_error(probe._session)
        if err is not None:
            kind = self._lib.ksched_error_kind(probe._session).decode()
            msg = err.decode()
            probe.close()
            _raise(kind, msg)
        if not hasattr(self, "_probes"):
            self._probes = []
        self._probes.append(probe)
        return probe

    def Sweep(self, candidates, prices=None, all_spot=None, detail=False, multi_node=False, replicas=()) -> dict:
        """A whole consolidation sweep of a RESIDENT cluster in one call (ksolve_sweep; disruption/helpers.go:53-155 +
        consolidation.go:159-256 per candidate set): `candidates` = one list of nodes (names, or positions in the problem's
        stateNodes list) per simulation, `prices` the summed candidate prices, `all_spot` whether every candidate is a spot
        node. The probe descriptors (displaced pods, NodePool limits handed back) are built by the host library, every
        simulation is one wavefront of one launch, the verdicts (0 no-op, 1 delete, 2 replace) come back with the replacement
        instance types of the replace commands. detail=True adds where every pod went (tests). multi_node=True: the sets are
        prefixes of MultiNodeConsolidation's binary search, a replace verdict over several candidates is filtered by
        filterOutSameInstanceType (multinodeconsolidation.go:209-246). replicas: Schedulers opened from the same cluster document on
        OTHER devices (options.device): the probes are dealt out over all of them inside the call (ksolve_sweep_replicas), the
        results are those of one device."""
        if not self._session:
            raise RuntimeError("scheduler is closed")
        if candidates and all(isinstance(x, int) for cs in candidates for x in cs):
            # positions in the stateNodes list: handed over as two flat arrays (CSR), not as ten thousand little lists
            off, flat = [0], []
            for cs in candidates:
                flat += cs
                off.append(len(flat))
            doc = {"candidateOff": off, "candidateNodes": flat, "detail": bool(detail), "multiNode": bool(multi_node)}
        else:
            doc = {"candidates": candidates, "detail": bool(detail), "multiNode": bool(multi_node)}
        if prices is not None:      # None: the library takes every candidate's price and capacity type from its own node table
            doc.update(prices=prices, allSpot=all_spot)
        doc = json.dumps(doc).encode()
        if replicas:
            sessions = (ctypes.c_void_p * (1 + len(replicas)))(self._session, *[r._session for r in replicas])
            ptr = self._lib.ksched_sweep_replicas(sessions, 1 + len(replicas), doc)
        else:
            ptr = self._lib.ksched_sweep(self._session, doc)
        try:
            out = json.loads(ctypes.string_at(ptr).decode())
        finally:
            self._lib.ksched_free(ptr)
        if "error" in out:
            _raise(out.get("kind"), out["error"])
        return out

    _SWEEP_TIMINGS = ("descriptors_ms", "sweep_ms", "verdicts_ms", "upload_us", "pack_us", "finalize_us", "download_us", "pods", "bin_evaluations", "node_evaluations",
                      "node_block_steps", "classes", "it_words", "nodes", "node_dead0_us", "req_words", "resources", "devices")

    def SweepArrays(self, candidates, multi_node=False, replicas=(), instance_type_names=None) -> dict:
        """Sweep() without JSON on either side of the host library (ksched_sweep_arrays) — what a cgo caller does: the candidate sets go
        in as one CSR of positions in the problem's stateNodes list, the verdicts come back as arrays (prices and capacity types are the
        session's own). Returns the same document as Sweep() (built here, outside the library's timed region); replacement instance
        types are names when `instance_type_names` (the problem's instanceTypes names, in order) is given, indices otherwise."""
        import numpy as np
        if not self._session:
            raise RuntimeError("scheduler is closed")
        n = len(candidates)
        off = np.zeros(n + 1, dtype=np.uint32)
        off[1:] = np.cumsum([len(c) for c in candidates], dtype=np.int64)
        flat = np.fromiter((x for cs in candidates for x in cs), dtype=np.uint32, count=int(off[n]))
        dec = np.zeros(n, np.int32); ok = np.zeros(n, np.uint8); live = np.zeros(n, np.uint32); status = np.zeros(n, np.int32); refs = np.zeros(n, np.uint64)
        roff = np.zeros(n + 1, np.uint32); spot = np.zeros(n, np.uint8); tv = np.zeros(len(self._SWEEP_TIMINGS), np.float64)
        sessions = (ctypes.c_void_p * (1 + len(replicas)))(self._session, *[r._session for r in replicas])
        cap = max(1 << 16, 64 * n)
        for _ in range(2):
            rits = np.zeros(cap, np.uint32)
            ptr = self._lib.ksched_sweep_arrays(sessions, 1 + len(replicas), n, off.ctypes.data, flat.ctypes.data if len(flat) else None, 1 if multi_node else 0,
                                                dec.ctypes.data, ok.ctypes.data, live.ctypes.data, status.ctypes.data, refs.ctypes.data, roff.ctypes.data, rits.ctypes.data, cap, spot.ctypes.data,
                                                tv.ctypes.data, len(tv))
            if not ptr:
                break
            try:
                err = json.loads(ctypes.string_at(ptr).decode())
            finally:
                self._lib.ksched_free(ptr)
            if err.get("kind") == "capacity" and int(roff[n]) > cap:
                cap = int(roff[n])
                continue
            _raise(err.get("kind"), err["error"])
        repl = []
        for p in np.nonzero(dec == 2)[0].tolist():
            its = rits[roff[p]:roff[p + 1]].tolist()
            repl.append({"probe": p, "instanceTypes": sorted(instance_type_names[i] for i in its) if instance_type_names else its, "capacityType": "spot" if spot[p] else None})
        timings = {k: (float(v) if k.endswith(("_ms", "_us")) else int(v)) for k, v in zip(self._SWEEP_TIMINGS, tv.tolist())}
        timings["probes"] = n
        return {"decisions": dec.tolist(), "allNonPendingPodsScheduled": [bool(x) for x in ok.tolist()], "claims": live.tolist(), "status": status.tolist(),
                "referenceBinEvaluations": [int(x) for x in refs.tolist()], "replacements": repl, "reasons": {}, "timings": timings}

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def Solve(self, repeat: int = 1, want_results=True) -> Results:
        """Runs Solve() on the device; `repeat` re-solves on the same resident inputs and collects every run's timings.
        want_results: False = counters and timings only; True = the full Results; "claims" = the NodeClaims without their
        pod lists (a 10M-pod packing summary does not need 10M uids)."""
        if not self._session:
            raise RuntimeError("scheduler is closed")
        timings = []
        out = None
        for _ in range(max(1, int(repeat))):
            ptr = self._lib.ksched_solve(self._session, _want(want_results))
            try:
                out = json.loads(ctypes.string_at(ptr).decode())
            finally:
                self._lib.ksched_free(ptr)
            if "error" in out:
                _raise(out.get("kind"), out["error"])
            timings += out["timings"]
        out["timings"] = timings
        return Results(out)


    def Assignment(self):
        """The per-pod outputs of the last Solve(want_results="claims") as two numpy arrays over the problem's pods in input order
        (explicit pods, then the groups' pods): assign[i] >= 0 is an index into newNodeClaims, <= -2 the existing node -2 - assign[i],
        -1 unscheduled; slot[i] is the pod's position in that NodeClaim's / node's Pods slice. This is what a caller that keeps
        its pods by position re-hydrates from — no uid text is formatted or parsed for it."""
        import numpy as np
        n = self._lib.ksched_assignment(self._session, None, None, 0)
        assign, slot = np.empty(n, dtype=np.int32), np.empty(n, dtype=np.uint32)
        if n:
            self._lib.ksched_assignment(self._session, assign.ctypes.data, slot.ctypes.data, n)
        return assign, slot

    def PodsByClaim(self, n_claims: int):
        """NodeClaim.Pods for every new NodeClaim of the last Solve(want_results="claims"), as arrays of pod positions in slot order
        (the order the reference appended them in, nodeclaim.go:248)."""
        import numpy as np
        off = np.zeros(n_claims + 1, dtype=np.uint32)
        n = self._lib.ksched_assignment(self._session, None, None, 0)    # the problem's pods: room enough for every claim's list, so one call counts and scatters
        pods = np.empty(max(1, n), dtype=np.uint32)
        total = self._lib.ksched_pods_by_claim(self._session, n_claims, off.ctypes.data, pods.ctypes.data, n)
        return [pods[off[c]:off[c + 1]] for c in range(n_claims)] if total <= n else []

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
    lib.ksched_solve_batch(sessions, n, _want(want_results), outs)
    docs = []
    for i in range(n):   # every document is copied out and released before anything is raised: no leak on the first error
        try:
            docs.append(json.loads(ctypes.string_at(outs[i]).decode()))
        finally:
            lib.ksched_free(outs[i])
    for doc in docs:
        if "error" in doc:
            _raise(doc.get("kind"), doc["error"])
    return [Results(doc) for doc in docs]


MAX_INSTANCE_TYPES = 600                      # scheduling.MaxInstanceTypes, nodeclaimtemplate.go:50
_WELL_KNOWN_LABELS = {"karpenter.sh/nodepool", "topology.kubernetes.io/zone", "topology.kubernetes.io/region", "node.kubernetes.io/instance-type",
                      "kubernetes.io/arch", "kubernetes.io/os", "karpenter.sh/capacity-type", "node.kubernetes.io/windows-build"}   # apis/v1/labels.go:75-84
_SIMULATION_KEYS = ("karpenter.sh/registered", "karpenter.sh/initialized")   # schedulingSimulationKeys, nodeclaimtemplate.go:43-46


def _req_admits(q, value):
    if q.get("gte") is not None or q.get("lte") is not None:
        try:
            v = int(value)
        except ValueError:
            return False
        if (q.get("gte") is not None and v < q["gte"]) or (q.get("lte") is not None and v > q["lte"]):
            return False
    return (value not in q["values"]) if q["complement"] else (value in q["values"])


def ToNodeClaim(claim: dict, problem: dict, max_instance_types: int = MAX_INSTANCE_TYPES) -> dict:
    """NodeClaimTemplate.ToNodeClaim (nodeclaimtemplate.go:109-175) for one NodeClaim of Results: the launch shaping that
    follows Solve(). The instance-type requirement becomes `In` over the options ordered by price and capped at
    MaxInstanceTypes (keeping the claim's minValues for that key), the capacity-type requirement is narrowed to the
    capacity types those options can actually launch with, the scheduling-simulation-only keys are dropped, and custom
    labels with a single possible value become node labels. Returns {"requirements", "labels", "instanceTypes"}.

    Claims with more than `max_instance_types` options need the order the reference's unstable sort produces: solve with
    options.truncateInstanceTypes set and the device returns it (Results.TruncateInstanceTypes, scheduler.go:419-437)."""
    by_name = {t["name"]: t for t in problem["instanceTypes"]}
    reqs = {q["key"]: dict(q) for q in claim["requirements"]}

    def compatible_offerings(name):
        return [o for o in by_name[name]["offerings"] if o.get("available", True)
                and all(r["key"] not in reqs or _req_admits(reqs[r["key"]], r["values"][0]) for r in o["requirements"])]

    names = list(claim["instanceTypes"])
    if len(names) > max_instance_types:
        raise ValueError("more instance type options than MaxInstanceTypes: solve with options.truncateInstanceTypes to get the reference's order")
    names.sort(key=lambda n: min([o["price"] for o in compatible_offerings(n)], default=float("inf")))      # OrderByPrice; ties only matter beyond the cap
    it_req = reqs.get("node.kubernetes.io/instance-type", {})
    reqs["node.kubernetes.io/instance-type"] = {"key": "node.kubernetes.io/instance-type", "complement": False, "values": sorted(names), "gte": None, "lte": None,
                                               "minValues": it_req.get("minValues"), "operator": "In"}
    cts = []
    for n in names:
        for o in compatible_offerings(n):
            ct = next(r["values"][0] for r in o["requirements"] if r["key"] == "karpenter.sh/capacity-type")
            if ct not in cts:
                cts.append(ct)
    if cts:
        old = reqs.get("karpenter.sh/capacity-type")
        keep = [c for c in cts if old is None or _req_admits(old, c)]
        reqs["karpenter.sh/capacity-type"] = {"key": "karpenter.sh/capacity-type", "complement": False, "values": sorted(keep), "gte": None, "lte": None,
                                              "minValues": (old or {}).get("minValues"), "operator": "In"}
    well_known = set(problem.get("wellKnownLabels", [])) | _WELL_KNOWN_LABELS
    labels = {}
    for k, q in reqs.items():
        if k in well_known or k == "kubernetes.io/hostname" or k in _SIMULATION_KEYS:   # WellKnown / Restricted / simulation keys
            continue
        if not q["complement"]:
            if q["values"]:
                labels[k] = sorted(q["values"])[0]    # Requirement.Any() of an In requirement: one of its values (requirement.go:256-259)
            continue                                  # DoesNotExist: no label
        # NotIn / Exists (possibly with Gt / Lt bounds): the reference draws a random integer inside the bounds
        # (requirement.go:260-269); any admissible value is a faithful answer — take the smallest one
        v = q["gte"] if q.get("gte") is not None else 0
        while str(v) in q["values"]:
            v += 1
        if q.get("lte") is None or v <= q["lte"]:
            labels[k] = str(v)
    out = [q for k, q in sorted(reqs.items()) if k not in _SIMULATION_KEYS]
    return {"requirements": out, "labels": labels, "instanceTypes": names}


def SplitComponents(problem: dict, bins: int = 1):
    """The NodePool components of one provisioning batch, from the host library (ksched_split_components): connected
    components of the pods x NodePools graph (pins on karpenter.sh/nodepool, topology selectors), dealt over `bins` devices by pod
    count, largest first. Returns {"components": [{"pools", "pods", "bin", "problem"}], "bins", "binPods"} or {"components": None,
    "reason"} when independence cannot be shown. The Go twin is go/ksolve_components.go."""
    lib = _load_ksched()
    ptr = lib.ksched_split_components(json.dumps(problem).encode(), int(bins))
    try:
        out = json.loads(ctypes.string_at(ptr).decode())
    finally:
        lib.ksched_free(ptr)
    if "error" in out:
        _raise(out.get("kind"), out["error"])
    return out


def device_available() -> bool:
    if not os.path.exists(KSOLVE_LIB):
        return False
    try:
        lib = ctypes.CDLL(KSOLVE_LIB)
        lib.ksolve_device_available.restype = ctypes.c_int
        return bool(lib.ksolve_device_available())
    except OSError:
        return False
