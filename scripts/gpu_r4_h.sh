# round 4, seventh GPU pass: the compact consolidation sweep (ksolve_pack_sweep4: four wavefronts per workgroup sharing the read-only
# tables, ScratchSmall working sets, 256 VGPRs -> eight probes per CU instead of four). A/B against the one-wavefront kernel inside the
# SAME binary (the test-hooks build reads KSOLVE_TEST_SWEEP_GENERAL), parity tests, smoke, a reduced bench line, kernel stats and PMC.
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4h; mkdir -p $O
export TMPDIR=/tmp
for mode in compact general; do
  if [ $mode = general ]; then export KSOLVE_TEST_SWEEP_GENERAL=1; else unset KSOLVE_TEST_SWEEP_GENERAL; fi
  S=0; [ $mode = compact ] && S=12
  timeout 300 python tests/tools/sweep_scale.py 100000 10000 $S --repeat 4 --solver-lib tests/emu/libksolve_hooks.so 2>$O/sweep_${mode}.err | tail -1 > $O/sweep_${mode}.json
  timeout 300 python tests/tools/sweep_scale.py 100000 10000 0 --topology --repeat 4 --solver-lib tests/emu/libksolve_hooks.so 2>$O/sweep_${mode}_topology.err | tail -1 > $O/sweep_${mode}_topology.json
done
unset KSOLVE_TEST_SWEEP_GENERAL
python - <<'PY'
import json
for m in ("compact", "general"):
    for t in ("", "_topology"):
        try:
            d = json.load(open(f"gpurun_out/r4h/sweep_{m}{t}.json"))
            print(m + t, d["decisions"], d["verdict_digest"], {k: d["timings"][k] for k in ("pack_us", "sweep_ms", "descriptors_ms", "verdicts_ms", "upload_us")}, d.get("oracle_checked"))
        except Exception as e:
            print(m + t, "failed", e)
PY
# per-probe shader cycles by verdict and phase (profiling build with the test switches: the same binary runs both forms)
for mode in compact general; do
  if [ $mode = general ]; then export KSOLVE_TEST_SWEEP_GENERAL=1; else unset KSOLVE_TEST_SWEEP_GENERAL; fi
  timeout 300 python tests/tools/sweep_probe_costs.py 100000 10000 48 --solver-lib karpenter_amd/variants/libksolve_timers.so 2>$O/probe_costs_${mode}.err | tail -1 > $O/probe_costs_${mode}.json
done
unset KSOLVE_TEST_SWEEP_GENERAL
python - <<'PY'
import json
for m in ("compact", "general"):
    try:
        d = json.load(open(f"gpurun_out/r4h/probe_costs_{m}.json"))
        print(m, d["sweep_timings"]["pack_us"], {k: (v["cycles_mean"], v["cycles_median"], v["cycles_p90"], v["cycles_max"]) for k, v in d["by_decision"].items()})
    except Exception as e:
        print(m, "failed", e)
PY
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -8 | tee $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4 | tee $O/smoke.log
timeout 900 python bench.py --topology-pods 0 --components-pods 0 --beyond-lds-pods 0 --whole-batch-exact-pods 0 --batch-problems 0 --sweep-sample 12 --sweep-topology-sample 4 2>$O/bench_reduced.err | tail -1 > $O/bench_reduced.json
tail -3 $O/bench_reduced.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r4h/bench_reduced.json"))
print("value", d["value"], "roofline", d["roofline"]["frac"], d["roofline"].get("traffic"))
c = d["config4_sweep"]; print("sweep", c["seconds"], c["value"], c["kernels"]["ksolve_pack_sweep"]); print("multi", c["multi_node"]["seconds"]); print("topo", c["with_topology_pods"]["seconds"], c["with_topology_pods"]["value"])
PY
KSOLVE_PMC_SKIP_CLASSING_ROWS=1 bash scripts/gpu_r4_pmc.sh 2>&1 | tail -12
cp gpurun_out/r4pmc/pmc_traffic.json $O/pmc_traffic.json
cp gpurun_out/r4pmc/rocprofv3_kernel_stats_sweep.csv gpurun_out/r4pmc/rocprofv3_kernel_stats_bench_1m.csv $O/
