# round 4, ninth GPU pass: the compact sweep with its probes handed out through a counter, largest first (final build): A/B against the
# one-wavefront kernel inside the test-hooks binary, parity tests, smoke, reduced bench line, kernel stats + PMC of THIS build
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4j; mkdir -p $O
export TMPDIR=/tmp
for mode in compact general; do
  if [ $mode = general ]; then export KSOLVE_TEST_SWEEP_GENERAL=1; else unset KSOLVE_TEST_SWEEP_GENERAL; fi
  S=0; [ $mode = compact ] && S=12
  timeout 300 python tests/tools/sweep_scale.py 100000 10000 $S --repeat 4 --solver-lib tests/emu/libksolve_hooks.so 2>$O/sweep_${mode}.err | tail -1 > $O/sweep_${mode}.json
  timeout 300 python tests/tools/sweep_scale.py 100000 10000 0 --topology --repeat 4 --solver-lib tests/emu/libksolve_hooks.so 2>$O/sweep_${mode}_topology.err | tail -1 > $O/sweep_${mode}_topology.json
done
unset KSOLVE_TEST_SWEEP_GENERAL
timeout 300 python tests/tools/sweep_scale.py 100000 50000 0 --repeat 3 2>$O/sweep_50k.err | tail -1 > $O/sweep_compact_50k_candidates.json
python - <<'PY'
import json
for m in ("compact", "general", "compact_50k_candidates"):
    for t in ("", "_topology"):
        try:
            d = json.load(open(f"gpurun_out/r4j/sweep_{m}{t}.json"))
            print(m + t, d["candidates"], d["decisions"], d["verdict_digest"], {k: d["timings"][k] for k in ("pack_us", "sweep_ms", "descriptors_ms", "verdicts_ms", "upload_us")}, d.get("oracle_checked"))
        except Exception as e:
            print(m + t, "failed", e)
PY
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -8 | tee $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4 | tee $O/smoke.log
KSOLVE_PMC_SKIP_CLASSING_ROWS=1 bash scripts/gpu_r4_pmc.sh 2>&1 | tail -8
cp gpurun_out/r4pmc/pmc_traffic.json profiles/round4/pmc_traffic.json
cp gpurun_out/r4pmc/pmc_traffic.json $O/pmc_traffic.json
cp gpurun_out/r4pmc/rocprofv3_kernel_stats_sweep.csv gpurun_out/r4pmc/rocprofv3_kernel_stats_bench_1m.csv $O/
timeout 900 python bench.py --topology-pods 0 --components-pods 0 --beyond-lds-pods 0 --whole-batch-exact-pods 0 --batch-problems 0 --sweep-sample 12 --sweep-topology-sample 4 2>$O/bench_reduced.err | tail -1 > $O/bench_reduced.json
tail -3 $O/bench_reduced.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r4j/bench_reduced.json"))
print("value", d["value"], "roofline", d["roofline"]["frac"], d["roofline"].get("traffic"))
c = d["config4_sweep"]; print("sweep", c["seconds"], c["value"], c["kernels"]["ksolve_pack_sweep"]); print("multi", c["multi_node"]["seconds"]); print("topo", c["with_topology_pods"]["seconds"], c["with_topology_pods"]["value"])
PY
