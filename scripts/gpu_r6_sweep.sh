# round 6: the consolidation sweep legs of bench.py alone (10k single-node probes plain and with topology pods, 3,200 multi-node prefixes;
# population pins + re-simulated probes checked in the run) and the sweep GPU tests.   usage (GPU box): bash scripts/gpu_r6_sweep.sh <tag>
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "sweep or resident or consolidation or window or probes" 2>&1 | tail -3 | tee $O/pytest_sweeps.log
timeout 900 python bench.py --steps 1 --warmup 0 --pods 20000 --no-parity-pin --topology-pods 0 --batch-problems 0 --components-pods 0 --beyond-lds-pods 0 --whole-batch-exact-pods 0 --whole-batch-pods 0 --no-host-engine-baseline --no-cpu-baseline 2>$O/bench_sweep.err | tail -1 > $O/bench_sweep.json
tail -2 $O/bench_sweep.err
python - $O <<'PY'
import json, sys
d = json.load(open(sys.argv[1] + "/bench_sweep.json"))
s = d["config4_sweep"]
print("single", {k: round(v * 1e3, 3) for k, v in s["seconds"].items()}, "pin", s.get("oracle_pin", {}).get("digest_matches_oracle"), "dead0_ms", s["kernels"]["ksolve_node_dead0"]["avg_kernel_ms"])
t = s.get("with_topology_pods", {})
print("topology", {k: round(v * 1e3, 3) for k, v in t.get("seconds", {}).items()}, "pin", t.get("oracle_pin", {}).get("digest_matches_oracle"))
m = s.get("multi_node", {})
print("multi", {k: round(v * 1e3, 3) for k, v in m.get("seconds", {}).items()}, "pin", m.get("oracle_pin", {}).get("digest_matches_oracle"))
PY
