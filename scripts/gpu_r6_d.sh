# round 6, GPU pass D: the whole default bench line on this build (every in-run gate), then the PMC / kernel-stats passes of
# scripts/gpu_r6_pmc.sh (headline, configs[2] leg: ksolve_pack_topo at 1M pods and ksolve_pack_big at 200k, sweep kernels).
set -x
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r6d; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python bench.py 2>$O/bench_default.err | tail -1 > $O/bench_default.json
tail -3 $O/bench_default.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r6d/bench_default.json"))
print("value", d["value"], "ms", d["ms_per_step"], "dominant", d["roofline"].get("dominant"))
t = d.get("config2_topology", {})
print("config2", {k: t.get(k) for k in ("seconds", "engine", "pack_kernel_ms", "oracle_pin")}, t.get("pack_kernel", {}).get("reference_equivalent"))
print("sweep", {k: v for k, v in d.get("config4_sweep", {}).get("seconds", {}).items()} if isinstance(d.get("config4_sweep", {}).get("seconds"), dict) else d.get("config4_sweep", {}).get("seconds"))
print("exact", d.get("config3_components", {}).get("whole_batch_exact", {}).get("seconds"), d.get("config3_components", {}).get("whole_batch_exact", {}).get("cursor_attempts"))
PY
KSOLVE_PMC_LEGS="head topo big sweep" timeout 3000 bash scripts/gpu_r6_pmc.sh 2>&1 | tail -12
cp gpurun_out/r6pmc/pmc_traffic.json gpurun_out/r6pmc/rocprofv3_kernel_stats_*.csv $O/
