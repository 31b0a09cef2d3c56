# round 3, last run: the default bench line and the GPU tests on the final tree (same kernel sources as profiles/round3/pmc_traffic.json)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3m
mkdir -p $O
timeout 1500 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -3 $O/bench_default.err; python - $O <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1] + "/bench_default.json") if l.startswith("{")][-1])
print("headline", d["value"], d["ms_per_step"], "roofline", d["roofline"]["avg_kernel_ms"], d["roofline"]["frac"], d["roofline"]["traffic"])
t = d["config2_topology"]; print("topology", t["seconds"], t["largest_pinned_size"]["pods"], t["largest_pinned_size"]["seconds"], t["largest_pinned_size"]["oracle_pin"])
print("sweep", d["config4_sweep"]["value"], d["config4_sweep"]["seconds"]["library_call"], "multi-node", d["config4_sweep"]["multi_node"]["value"], d["config4_sweep"]["multi_node"]["seconds"]["library_call"])
PY
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
