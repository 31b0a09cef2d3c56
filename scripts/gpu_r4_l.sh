# round 4, last GPU pass: the final build (compact sweep kernel + staged page-locked uploads): NewScheduler / upload / re-hydration profile,
# parity tests, smoke, PMC + kernel stats of THIS build (bench.py quotes `traffic` by source hash), the bench line without the long legs
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4l; mkdir -p $O
export TMPDIR=/tmp
timeout 200 python tests/tools/gpu_rehydrate_profile.py 2>&1 | tail -8 | cut -c1-400 | tee $O/rehydrate_profile.log
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -8 | tee $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4 | tee $O/smoke.log
KSOLVE_PMC_SKIP_CLASSING_ROWS=1 bash scripts/gpu_r4_pmc.sh 2>&1 | tail -6
cp gpurun_out/r4pmc/pmc_traffic.json profiles/round4/pmc_traffic.json
cp gpurun_out/r4pmc/pmc_traffic.json $O/pmc_traffic.json
cp gpurun_out/r4pmc/rocprofv3_kernel_stats_sweep.csv gpurun_out/r4pmc/rocprofv3_kernel_stats_bench_1m.csv $O/
timeout 900 python bench.py --topology-pods 0 --components-pods 0 --beyond-lds-pods 0 --whole-batch-exact-pods 0 --batch-problems 0 --sweep-sample 12 --sweep-topology-sample 4 2>$O/bench_reduced.err | tail -1 > $O/bench_reduced.json
tail -3 $O/bench_reduced.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r4l/bench_reduced.json"))
print("value", d["value"], "roofline", d["roofline"]["frac"], d["roofline"].get("traffic"))
print("e2e", d["end_to_end"])
c = d["config4_sweep"]; print("sweep", c["seconds"], c["value"]); print("multi", c["multi_node"]["seconds"], c["multi_node"].get("first_call_python_s")); print("topo", c["with_topology_pods"]["seconds"], c["with_topology_pods"]["value"])
PY
