# NOTE: as run at commit 361dd55, when the cursor engine still had its frontier window (KSOLVE_FAST_WINDOW; removed after these measurements, profiles/README.md)
# round 3: SQ instruction / cycle counters of the pack kernel with the frontier window off and on (configs[1], 1M pods)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3f
mkdir -p $O
BENCH="python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --topology-pods 0 --batch-problems 0 --components-pods 0 --sweep-nodes 0 --no-host-engine-baseline --no-cpu-baseline"
(cd /tmp && rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u > $O/sq_counters.txt)
for W in 0 1; do
(cd /tmp && KSOLVE_FAST_WINDOW=$W timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/pmc_a$W -o sq -- $BENCH > $O/pmc_a$W.log 2>&1)
(cd /tmp && KSOLVE_FAST_WINDOW=$W timeout 600 rocprofv3 --pmc SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU --kernel-trace --output-format csv -d $O/pmc_b$W -o sq -- $BENCH > $O/pmc_b$W.log 2>&1)
done
python - $O <<'PY'
import csv, sys, glob, collections, json
O = sys.argv[1]
out = {}
for tag in ("a0", "b0", "a1", "b1"):
    fs = glob.glob(f"{O}/pmc_{tag}/**/*counter_collection*.csv", recursive=True)
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in fs:
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            acc[(k, r["Counter_Name"])][0] += float(r["Counter_Value"]); acc[(k, r["Counter_Name"])][1] += 1
    for (k, c), (v, n) in sorted(acc.items()):
        if "pack_fast" in k:
            print(tag, k, c, "launches", n, "per launch", v / n)
            out.setdefault("window" + tag[1], {})[c] = v / n
json.dump(out, open(f"{O}/sq_pack_fast.json", "w"), indent=1)
PY
tail -3 $O/pmc_b1.log
