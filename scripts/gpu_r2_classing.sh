# classing prepass: kernel stats of the short bench (headline leg only) + a few GPU parity tests that exercise classing
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r2c
mkdir -p $O
if [ -z "$KSOLVE_COOP_PROBE" ]; then timeout 600 python -m pytest tests -m gpu -x -q -k "config1 or edge or scaled or topology_mix or full_size_digest" > $O/pytest_some.log 2>&1; tail -3 $O/pytest_some.log; fi
BENCH="python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --topology-pods 0 --batch-problems 0 --components-pods 0 --no-host-engine-baseline --no-cpu-baseline"
(cd /tmp && KSOLVE_COOP_PROBE=$KSOLVE_COOP_PROBE timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- $BENCH > $O/stats.log 2>&1)
python - $O <<'PY'
import json, sys, glob, csv
O = sys.argv[1]
d = json.loads(open(O + "/stats.log").read().strip().splitlines()[-1])
print("value", d["value"], "stream", d["roofline_stream"], d["phases_ms"])
f = glob.glob(O + "/stats/**/*kernel_stats*.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "ksolve" in r["Name"]: print(r["Name"].split("(")[0], r["Calls"], r["AverageNs"])
PY
