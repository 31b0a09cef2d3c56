# round 4, eighth GPU pass: how the pack kernels are fed with instructions (SQC instruction-cache counters, wait / active cycles),
# separate --pmc passes with --kernel-trace only
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4i; mkdir -p $O
CMD="python $GRAFT_REPO_ROOT/tests/tools/gpu_ifetch_probe.py"
(cd /tmp && timeout 300 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH --kernel-trace --output-format csv -d $O/p1 -o a -- $CMD > $O/p1.log 2>&1)
(cd /tmp && timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $O/p2 -o b -- $CMD > $O/p2.log 2>&1)
(cd /tmp && timeout 300 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_IFETCH_LEVEL --kernel-trace --output-format csv -d $O/p3 -o c -- $CMD > $O/p3.log 2>&1)
tail -8 $O/p1.log
python - $O <<'PY'
import csv, sys, glob, collections, json, re
O = sys.argv[1]
acc = collections.defaultdict(float); cnt = collections.Counter(); big = collections.defaultdict(float)
for f in glob.glob(f"{O}/p*/**/*counter_collection*.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"^void ", "", r["Kernel_Name"]).split("(")[0].split("<")[0]
        if "ksolve_pack" not in k: continue
        v = float(r["Counter_Value"])
        acc[(k, r["Counter_Name"])] += v; cnt[(k, r["Counter_Name"])] += 1; big[(k, r["Counter_Name"])] = max(big[(k, r["Counter_Name"])], v)
out = {}
for (k, c), v in sorted(acc.items()):
    out.setdefault(k, {})[c] = {"sum": v, "launches": cnt[(k, c)], "largest_launch": big[(k, c)]}
json.dump(out, open(f"{O}/ifetch_counters.json", "w"), indent=1)
for k, d in out.items():
    print(k, {c: int(x["largest_launch"]) for c, x in d.items()})
PY
rm -rf $O/p1 $O/p2 $O/p3
