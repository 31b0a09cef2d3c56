# round 5, final GPU pass: every GPU parity test, smoke(), PMC + kernel stats of THIS build (bench.py quotes `traffic` and the SQ
# counters by source hash), the default bench line
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5final; mkdir -p $O
export TMPDIR=/tmp
timeout 1700 python -m pytest tests -m gpu -q 2>&1 | tail -8 | tee $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4 | tee $O/smoke.log
KSOLVE_PMC_SKIP_CLASSING_ROWS=1 bash scripts/gpu_r5_pmc.sh 2>&1 | tail -8
mkdir -p profiles/round5
cp gpurun_out/r5pmc/pmc_traffic.json profiles/round5/pmc_traffic.json
cp gpurun_out/r5pmc/pmc_traffic.json gpurun_out/r5pmc/rocprofv3_kernel_stats_sweep.csv gpurun_out/r5pmc/rocprofv3_kernel_stats_bench_1m.csv $O/
timeout 1200 python bench.py 2>$O/bench_default.err | tail -1 > $O/bench_default.json
tail -3 $O/bench_default.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r5final/bench_default.json"))
print("value", d["value"], "ms", d["ms_per_step"], "roofline", d["roofline"]["frac"], d["roofline"].get("traffic"))
print("pack", {k: d["pack_kernel"].get(k) for k in ("avg_kernel_ms", "us_per_pod", "sq_counters", "traffic")})
print("cpu", d.get("cpu_baseline", {}).get("value"), d.get("cpu_baseline_full_size", {}).get("value"), d.get("cpu_baseline_engine_host"))
print("e2e", d["end_to_end"]["pods_per_s_through_the_boundary"])
t = d.get("config2_topology", {}); print("topology", t.get("seconds"), t.get("oracle_pin"))
b = d.get("config1_beyond_lds", {}); print("beyond", b.get("seconds"), b.get("value"), b.get("oracle_pin"))
c = d.get("config3_components", {}); print("components", {k: c.get(k) for k in ("seconds", "value")}, (c.get("whole_batch_exact") or {}).get("seconds"), (c.get("whole_batch") or {}).get("seconds"))
s = d.get("config4_sweep", {}); print("sweep", s.get("seconds"), s.get("value"), s.get("oracle_pin")); print("multi", (s.get("multi_node") or {}).get("seconds"), (s.get("multi_node") or {}).get("oracle_pin")); print("topo sweep", (s.get("with_topology_pods") or {}).get("value"), (s.get("with_topology_pods") or {}).get("oracle_pin"))
print("batched", d.get("batched"))
PY
