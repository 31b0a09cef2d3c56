# round 6: kernel stats + SQ counters of the spread engine on the configs[2]-shape pin at 1M pods -> gpurun_out/$1/   usage (GPU box): bash scripts/gpu_r6_prof.sh <tag> [pin]
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $O
PIN=${2:-config3_p1000000_t500_s42}
export TMPDIR=/tmp
CMD="python $GRAFT_REPO_ROOT/tests/tools/gpu_check_pin.py $GRAFT_REPO_ROOT/tests/golden/fullsize/$PIN.json spread"
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o topo -- $CMD > $O/stats.log 2>&1)
tail -1 $O/stats.log
find $O/stats -name "*kernel_stats*.csv" | head -1 | xargs -r -I{} cp {} $O/rocprofv3_kernel_stats_config3.csv
cut -c1-160 $O/rocprofv3_kernel_stats_config3.csv | head -6
(cd /tmp && timeout 400 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmc_sq -o sq -- $CMD > $O/pmc_sq.log 2>&1)
(cd /tmp && timeout 400 rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $O/pmc_sq2 -o sq -- $CMD > $O/pmc_sq2.log 2>&1)
(cd /tmp && timeout 400 rocprofv3 --pmc SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_INSTS_FLAT SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d $O/pmc_sq3 -o sq -- $CMD > $O/pmc_sq3.log 2>&1)
python - $O $PIN <<'PY'
import csv, sys, glob, json, re
O = sys.argv[1]
pods = int(re.search(r"_p(\d+)_", sys.argv[2]).group(1))
out = {}
for tag in ("pmc_sq", "pmc_sq2", "pmc_sq3"):
    for f in glob.glob(f"{O}/{tag}/**/*counter_collection*.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = re.sub(r"^void ", "", r["Kernel_Name"]).split("(")[0]
            if "pack_topo" in k: out[r["Counter_Name"]] = out.get(r["Counter_Name"], 0) + float(r["Counter_Value"])
out["pods"] = pods
out["per_pod"] = {k: round(v / pods, 2) for k, v in out.items() if k not in ("SQ_WAVES", "pods")}
json.dump(out, open(f"{O}/sq_counters_pack_topo.json", "w"), indent=1)
print(json.dumps(out["per_pod"]), "waves", out.get("SQ_WAVES"))
PY
