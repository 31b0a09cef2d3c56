"""Runs the functions of tests/test_gpu_parity.py against the host emulation of the solver instead of the device: a check of
the TESTS (fixtures, expectations, sizes) on a machine without a GPU, so that a red GPU run at round end means the
device and not the test. Not part of the product or of the pytest suites."""
import os as _os
_os.environ.setdefault("KSOLVE_TEST_SOLVER_LIB", "1")   # a test tool: may hand a test build of the solver library to NewScheduler(solver_lib=)
import sys, os, inspect, time, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests')); os.chdir(ROOT)
import parity, oracle
oracle.build()
emu = parity.build_emu()
import karpenter_amd.scheduling as ks
ks.KSOLVE_LIB = emu
_orig = ks.Scheduler.__init__
def _init(self, problem, solver_lib=None): _orig(self, problem, solver_lib or emu)
ks.Scheduler.__init__ = _init
import test_gpu_parity as t
import pathlib, tempfile
failed = 0
only = sys.argv[1:]      # optional: substrings of the test names to run
class _Env:              # the one thing the tests ask of pytest's monkeypatch
    def setenv(self, k, v): os.environ[k] = v
for name, fn in sorted(inspect.getmembers(t, inspect.isfunction)):
    if not name.startswith("test_") or name == "test_plain_c_example_on_the_device" or (only and not any(o in name for o in only)):
        continue
    params = inspect.signature(fn).parameters
    marks = [m for m in getattr(fn, "pytestmark", []) if m.name == "parametrize"]
    cases = [()]
    if marks:
        names = [x.strip() for x in marks[0].args[0].split(",")]
        cases = [dict(zip(names, v if isinstance(v, (tuple, list)) else (v,))) for v in marks[0].args[1]]
    for case in cases:
        kw = dict(case) if case else {}
        if "oracle" in params: kw["oracle"] = oracle
        if "tmp_path" in params: kw["tmp_path"] = pathlib.Path(tempfile.mkdtemp())
        if "monkeypatch" in params: kw["monkeypatch"] = _Env()
        t0 = time.time()
        try:
            fn(**kw); print("ok  ", name, case if case else "", round(time.time()-t0, 1), "s", flush=True)
        except Exception as e:
            failed += 1; print("FAIL", name, case, type(e).__name__, str(e)[:300], flush=True); traceback.print_exc()
print("failed:", failed)
