# round 4, fifth GPU pass (the build with the cursor engine's plan 2: claim order in HBM too): parity tests, smoke, the configs[1] mix at
# 4M pods on plan 1 against plan 2 (same digest), the EXACT configs[3] batch of 10M pods as one Solve() (27,345 NodeClaims: plan 2),
# PMC / kernel stats of THIS build (-> profiles/round4/pmc_traffic.json), the bench line
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4e; mkdir -p $O
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
timeout 700 python tests/tools/whole_batch_c3.py --pods 10000000 --out $O/whole_batch_c3_10m.json 2>&1 | tail -3 | cut -c1-1500 | tee $O/whole_batch_c3_10m.log
bash scripts/gpu_r4_pmc.sh 2>&1 | tail -30
cp gpurun_out/r4pmc/pmc_traffic.json profiles/round4/pmc_traffic.json
timeout 1800 python bench.py 2>$O/bench.err | tail -1 | tee $O/bench.json
tail -5 $O/bench.err
