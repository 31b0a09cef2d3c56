cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 240 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
timeout 60 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke.log 2>&1; echo "smoke exit $?" >> gpurun_out/smoke.log
timeout 100 python - > gpurun_out/quick.log 2>&1 <<'PY'
from karpenter_amd import fixtures as fx
from karpenter_amd.scheduling import NewScheduler
for label, prob in (("config2 200k", fx.config2(pods=200000)), ("config3 100k", fx.config3(pods=100000, n_types=500, seed=42, anti_affinity_pods=3000))):
    s = NewScheduler(prob)
    r = s.Solve(repeat=3, want_results=False)
    c = r["counters"]
    print(label, "pack ms", [round(t["pack_kernel_ms"], 1) for t in r["timings"]], "pods", c["pods"], "claims", c["claims"], "evals", c["binEvaluations"], "V", c["referenceBinEvaluations"])
PY
tail -5 gpurun_out/pytest_gpu.log; tail -3 gpurun_out/smoke.log; cat gpurun_out/quick.log
