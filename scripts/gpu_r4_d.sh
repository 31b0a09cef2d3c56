# round 4, fourth GPU pass (the build with the cursor engine's claim state in HBM above ~3,000 NodeClaims): parity tests, smoke,
# the configs[1] mix at 2M pods on the cursor engine's HBM plan against the general engine (same digest, both timed), PMC / kernel
# stats of THIS build (-> profiles/round4/pmc_traffic.json), the bench line, and a rocprofv3 --marker-trace run (ROCTx phase ranges)
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4d; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -8 | tee $O/pytest_gpu.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4 | tee $O/smoke.log
timeout 600 python - <<'PY' 2>&1 | tail -12 | tee $O/beyond_lds_2m.log
import sys, time, json
sys.path.insert(0, "tests")
import parity
from karpenter_amd import fixtures as fx
from karpenter_amd.scheduling import NewScheduler
for pods in (1_000_000, 2_000_000, 4_000_000):
    row = {"pods": pods}
    for eng in ("auto", "cursor", "general"):
        if eng == "cursor" and pods != 1_000_000: continue
        if eng == "general" and pods > 2_000_000: continue
        p = fx.config2(pods=pods, n_types=500, seed=42); p["options"]["engine"] = eng
        s = NewScheduler(p)
        t = time.time(); r = s.Solve(want_results=False); first = time.time() - t
        t = time.time(); r = s.Solve(want_results=False); dt = time.time() - t
        f = s.Solve(want_results=True); d, _ = parity.results_digest(f); s.close()
        row[eng] = {"first_s": round(first, 3), "s": round(dt, 3), "pack_ms": round(r["timings"][0]["pack_kernel_ms"], 1), "engine": r["counters"].get("engine"), "hbm_state": r["counters"].get("cursorClaimStateInHBM"), "claims": r["counters"]["claims"], "digest": d[:16], "evals": f["counters"]["referenceBinEvaluations"]}
    print(json.dumps(row))
PY
bash scripts/gpu_r4_pmc.sh 2>&1 | tail -30
cp gpurun_out/r4pmc/pmc_traffic.json profiles/round4/pmc_traffic.json
timeout 1800 python bench.py 2>$O/bench.err | tail -1 | tee $O/bench.json
tail -5 $O/bench.err
(cd /tmp && timeout 300 rocprofv3 --marker-trace --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/marker -o m -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --pods 200000 --no-parity-pin --topology-pods 0 --batch-problems 0 --components-pods 0 --beyond-lds-pods 0 --sweep-nodes 20000 --sweep-candidates 2000 --sweep-sample 0 --sweep-topology-sample 0 --sweep-windows 0 --no-host-engine-baseline --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/marker.log 2>&1)
find $O/marker -name "*marker*" | head -5
for f in $(find $O/marker -name "*marker*stats*.csv" | head -1); do cut -c1-160 $f | head -14; cp $f $O/roctx_marker_stats.csv; done
tail -3 $O/marker.log
rm -rf $O/marker
