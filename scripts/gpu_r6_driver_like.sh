# round 6: what the driver runs at round end, on the committed tree: every GPU test, smoke(), the bench line with the driver's flags.
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-driver_like}; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | tee $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee $O/smoke.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2>$O/bench.err | tail -1 > $O/bench.json
python - $O <<'PY'
import json, sys
d = json.load(open(sys.argv[1] + "/bench.json"))
s = d["config4_sweep"]
print("value", round(d["value"]), d["ms_per_step"], "roofline", d["roofline"]["frac"], d["roofline"]["traffic"], "config2", d["config2_topology"]["seconds"], "sweep", round(s["seconds"]["library_call"] * 1e3, 3), round(s["multi_node"]["seconds"]["library_call"] * 1e3, 3), "pins", d["parity"]["oracle_pin"]["digest_matches_oracle"], s["oracle_pin"]["digest_matches_oracle"], s["multi_node"]["oracle_pin"]["digest_matches_oracle"])
PY
