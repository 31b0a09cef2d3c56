"""The N>1 leg of bench.py has torch (its bundled HIP runtime, RCCL) and karpenter_amd/libksolve.so in ONE process. gpurun
boxes have a single GPU, so this test pins the part of that leg a 1-GPU box can show: with torch imported and its device
context live BEFORE the solver library is loaded, a Solve() through the C ABI is still bit-exact against the oracle, and
torch work before and after it is unaffected. Runs in a child process (named zz: last in the -m gpu run)."""
import json
import os
import subprocess
import sys

import pytest

import parity
from karpenter_amd import fixtures as fx

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import json, sys
sys.path.insert(0, sys.argv[1])
lib = sys.argv[2] or None
import torch
on_gpu = torch.cuda.is_available()
dev = "cuda" if on_gpu else "cpu"
a = torch.arange(1 << 20, device=dev, dtype=torch.int64)
before = int(a.sum().item())
from karpenter_amd import fixtures as fx
from karpenter_amd.scheduling import NewScheduler
prob = fx.config2(pods=3000, n_types=144, seed=11)
sched = NewScheduler(prob, solver_lib=lib)
res = [sched.Solve(), sched.Solve()][-1]          # twice: the handle is reused like in bench.py
after = int((a * 2).sum().item())
if on_gpu:
    torch.cuda.synchronize()
print("RESULT " + json.dumps({"on_gpu": on_gpu, "before": before, "after": after, "results": res}))
"""


def run_child(solver_lib=""):
    r = subprocess.run([sys.executable, "-c", CHILD, ROOT, solver_lib], cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    return json.loads(next(l for l in r.stdout.splitlines() if l.startswith("RESULT "))[7:])


def check(out, oracle):
    n = 1 << 20
    assert out["before"] == n * (n - 1) // 2 and out["after"] == n * (n - 1)
    want = oracle.solve(fx.config2(pods=3000, n_types=144, seed=11))
    parity.assert_same_results(out["results"], want)
    assert out["results"]["counters"]["referenceBinEvaluations"] == want["counters"]["binEvaluations"]


@pytest.mark.gpu
def test_solve_next_to_a_live_torch_context(oracle):
    import __graft_entry__
    __graft_entry__.build()
    out = run_child()
    assert out["on_gpu"], "torch sees no GPU on the GPU box"
    check(out, oracle)


def test_child_logic_on_the_host_build(oracle):
    """The same child with the engine's host build (no GPU): keeps the GPU test's own logic honest."""
    check(run_child(parity.build_emu()), oracle)
