# round 6, final pass: every GPU parity test and smoke(), the default bench line (every in-run gate), the PMC / kernel-stats passes (headline, configs[2] leg, sweep kernels, and
# — KSOLVE_PMC_LEGS with "exact" — the exact configs[3] batch), then bench.py once more so that its line quotes the counters of THIS build.
# usage (GPU box): bash scripts/gpu_r6_final.sh [tag]
set -x
cd $GRAFT_REPO_ROOT
T=${1:-r6final}
O=$GRAFT_REPO_ROOT/gpurun_out/$T; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4 | tee $O/smoke.log
KSOLVE_PMC_LEGS="${KSOLVE_PMC_LEGS:-head topo big sweep exact}" timeout 3600 bash scripts/gpu_r6_pmc.sh 2>&1 | tail -14
mkdir -p profiles/round6
cp gpurun_out/r6pmc/pmc_traffic.json profiles/round6/pmc_traffic.json
cp gpurun_out/r6pmc/pmc_traffic.json gpurun_out/r6pmc/rocprofv3_kernel_stats_*.csv $O/
timeout 900 python bench.py 2>$O/bench_default.err | tail -1 > $O/bench_default.json
tail -3 $O/bench_default.err
python - $O <<'PY'
import json, sys
d = json.load(open(sys.argv[1] + "/bench_default.json"))
print("value", d["value"], "ms", d["ms_per_step"], "sq", d["pack_kernel"]["sq_counters"], "traffic", d["roofline"]["traffic"])
t = d.get("config2_topology", {})
print("config2", {k: t.get(k) for k in ("seconds", "engine", "pack_kernel_ms")}, t.get("oracle_pin"), t.get("pack_kernel", {}).get("sq_counters"))
s = d["config4_sweep"]
print("sweep", {k: round(v * 1e3, 3) for k, v in s["seconds"].items()}, s.get("oracle_pin", {}).get("digest_matches_oracle"), s["kernels"]["ksolve_pack_sweep"].get("traffic"), s["kernels"]["ksolve_node_dead0"]["avg_kernel_ms"])
print("exact", {k: d.get("config3_components", {}).get("whole_batch_exact", {}).get(k) for k in ("seconds", "pack_kernel_ms", "cursor_attempts")})
print("e2e", d["end_to_end"]["pods_per_s_through_the_boundary"])
PY
