# round 3, final build: counters + kernel stats (scripts/gpu_pmc_traffic.sh), the default bench line quoting them, GPU tests, smoke()
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
bash scripts/gpu_pmc_traffic.sh
cp gpurun_out/r3pmc/pmc_traffic.json profiles/round3/pmc_traffic.json
O=$GRAFT_REPO_ROOT/gpurun_out/r3final
mkdir -p $O
timeout 1500 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -3 $O/bench_default.err; python - $O <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1] + "/bench_default.json") if l.startswith("{")][-1])
print("headline", d["value"], d["ms_per_step"], "roofline", d["roofline"]["avg_kernel_ms"], d["roofline"]["frac"], d["roofline"]["traffic"], "pack traffic", d["pack_kernel"]["traffic"])
print("topology", d["config2_topology"]["seconds"], "components", d["config3_components"]["value"], "sweep", d["config4_sweep"]["value"], d["config4_sweep"]["seconds"])
print("multi-node", json.dumps(d["config4_sweep"].get("multi_node"))[:1800])
PY
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -3 $O/smoke.log
