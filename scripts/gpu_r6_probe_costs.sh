# round 6: where a single-node probe's cycles go on the device (tests/tools/sweep_probe_costs.py through the -DKSOLVE_PHASE_TIMERS build of
# the CURRENT sources: python -c "import __graft_entry__ as g, os; g.build_ksolve('karpenter_amd/variants/libksolve_timers.so', defines=('-DKSOLVE_PHASE_TIMERS',), force=True)").
# usage (GPU box): bash scripts/gpu_r6_probe_costs.sh <tag>
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python tests/tools/sweep_probe_costs.py 100000 10000 48 --solver-lib $GRAFT_REPO_ROOT/karpenter_amd/variants/libksolve_timers.so 2>$O/probe_costs.err | tail -1 > $O/probe_costs_compact.json
python - $O <<'PY'
import json, sys
d = json.load(open(sys.argv[1] + "/probe_costs_compact.json"))
print(d["sweep_timings"])
for k, v in d["by_decision"].items():
    print(k, v["probes"], "mean", v["cycles_mean"], "median", v["cycles_median"], "p90", v["cycles_p90"], {a: b for a, b in v["phase_mean"].items()})
PY
