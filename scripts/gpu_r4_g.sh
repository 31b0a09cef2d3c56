# round 4, sixth GPU pass (final build: plain loads for the HBM claim state, pipelined searches over an HBM claim order, sweep descriptors in
# page-locked staging, page-locked uploads, the binary sweep call, the exact 10M batch inside bench.py): parity tests, smoke, the memory
# plans at 1M / 4M pods, the upload with and without hipHostRegister (test-hooks build), PMC / kernel stats of THIS build, the bench line
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4g; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -8 | tee $O/pytest_gpu.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4 | tee $O/smoke.log
timeout 600 python - <<'PY' 2>&1 | tail -12 | tee $O/plans_4m.log
import sys, time, json
sys.path.insert(0, "tests")
import parity
from karpenter_amd import fixtures as fx
from karpenter_amd.scheduling import NewScheduler
for pods in (1_000_000, 4_000_000):
    row = {"pods": pods}
    for eng in ("cursor", "cursor-wide", "cursor-hbm"):
        if eng == "cursor" and pods != 1_000_000: continue
        p = fx.config2(pods=pods, n_types=500, seed=42); p["options"]["engine"] = eng
        s = NewScheduler(p)
        t = time.time(); r = s.Solve(want_results=False); first = time.time() - t
        t = time.time(); r = s.Solve(want_results=False); dt = time.time() - t
        f = s.Solve(want_results=True); d, _ = parity.results_digest(f); s.close()
        row[eng] = {"s": round(dt, 3), "pack_ms": round(r["timings"][0]["pack_kernel_ms"], 1), "plan": r["counters"].get("cursorMemoryPlan"), "claims": r["counters"]["claims"], "slow_sorts": r["counters"]["slowSorts"], "digest": d[:16], "evals": f["counters"]["referenceBinEvaluations"]}
    print(json.dumps(row))
PY
for nr in 0 1; do
  if [ $nr = 1 ]; then export KSOLVE_TEST_NO_HOST_REGISTER=1; else unset KSOLVE_TEST_NO_HOST_REGISTER; fi
  KSOLVE_TEST_SOLVER_LIB=1 timeout 300 python - <<'PY' 2>&1 | tail -3 | tee -a $O/upload_host_register.log
import os, sys, time, json
sys.path.insert(0, "tests")
import parity
from karpenter_amd import fixtures as fx
from karpenter_amd.scheduling import NewScheduler
p = fx.config2(pods=1_000_000, n_types=500, seed=42)
hooks = parity.build_hooks()
rows = []
for i in range(4):
    t = time.time(); s = NewScheduler(p, solver_lib=hooks); dt = time.time() - t
    r = s.Solve(want_results=False); rows.append((round(dt, 4), round(r["timings"][0].get("upload_us", -1000) / 1000, 2))); s.close()
print(json.dumps({"host_register": os.environ.get("KSOLVE_TEST_NO_HOST_REGISTER") is None, "new_scheduler_s, upload_ms": rows}))
PY
done
unset KSOLVE_TEST_NO_HOST_REGISTER
bash scripts/gpu_r4_pmc.sh 2>&1 | tail -30
cp gpurun_out/r4pmc/pmc_traffic.json profiles/round4/pmc_traffic.json
timeout 1800 python bench.py 2>$O/bench.err | tail -1 | tee $O/bench.json
tail -5 $O/bench.err
