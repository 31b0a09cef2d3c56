# round 6, GPU pass A: the spread engine (csrc/topo_engine.h) on the device for the first time — its GPU tests, the configs[2]-shape pins
# on it and on the general engine, kernel stats + SQ counters of the 1M-pod pin.   usage (GPU box): bash scripts/gpu_r6_a.sh
set -x
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r6a; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_spread_engine.py -m gpu -x -q 2>&1 | tail -5 | tee $O/pytest_spread.log
for pin in config3_p200000_t500_s42 config3_p1000000_t500_s42; do
  timeout 300 python tests/tools/gpu_check_pin.py tests/golden/fullsize/$pin.json spread 2>&1 | tail -1 | tee -a $O/pins.log
done
timeout 300 python tests/tools/gpu_check_pin.py tests/golden/fullsize/config3_p200000_t500_s42.json general 2>&1 | tail -1 | tee -a $O/pins.log
CMD="python $GRAFT_REPO_ROOT/tests/tools/gpu_check_pin.py $GRAFT_REPO_ROOT/tests/golden/fullsize/config3_p1000000_t500_s42.json spread"
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o topo -- $CMD > $O/stats.log 2>&1)
find $O/stats -name "*kernel_stats*.csv" | head -1 | xargs -r -I{} cp {} $O/rocprofv3_kernel_stats_config3_1m.csv
cut -c1-160 $O/rocprofv3_kernel_stats_config3_1m.csv | head -8
(cd /tmp && timeout 400 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmc_sq -o sq -- $CMD > $O/pmc_sq.log 2>&1)
(cd /tmp && timeout 400 rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $O/pmc_sq2 -o sq -- $CMD > $O/pmc_sq2.log 2>&1)
python - $O <<'PY'
import csv, sys, glob, json, re
O = sys.argv[1]
out = {}
for tag in ("pmc_sq", "pmc_sq2"):
    for f in glob.glob(f"{O}/{tag}/**/*counter_collection*.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = re.sub(r"^void ", "", r["Kernel_Name"]).split("(")[0]
            if "pack_topo" in k: out[r["Counter_Name"]] = out.get(r["Counter_Name"], 0) + float(r["Counter_Value"])
pods = 1000000
out["per_pod"] = {k: v / pods for k, v in out.items() if k != "SQ_WAVES"}
json.dump(out, open(f"{O}/sq_counters_pack_topo.json", "w"), indent=1)
print(json.dumps(out["per_pod"]))
PY
