# round 2, final build: the default bench line, rocprofv3 kernel stats of the bench command (headline leg only), TCC traffic of
# the same command (FETCH_SIZE and WRITE_SIZE in their own --pmc passes, --kernel-trace only)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r2f
mkdir -p $O
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
tail -c 1500 $O/bench_default.json; tail -3 $O/bench_default.err
BENCH="python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --topology-pods 0 --batch-problems 0 --components-pods 0 --no-host-engine-baseline --no-cpu-baseline"
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- $BENCH > $O/stats.log 2>&1)
find $O/stats -name "*kernel_stats*.csv" | head -1 | xargs -r cut -c1-150 | head -16
(cd /tmp && timeout 400 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/pmc_sq -o sq -- $BENCH > $O/pmc_sq.log 2>&1)
(cd /tmp && timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o f -- $BENCH > $O/pmc_fetch.log 2>&1)
(cd /tmp && timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o w -- $BENCH > $O/pmc_write.log 2>&1)
python - $O <<'PY'
import csv, sys, glob, collections, json
O = sys.argv[1]
out = {}
for tag in ("sq", "fetch", "write"):
    fs = glob.glob(f"{O}/pmc_{tag}/**/*counter_collection*.csv", recursive=True)
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in fs:
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            acc[(k, r["Counter_Name"])][0] += float(r["Counter_Value"]); acc[(k, r["Counter_Name"])][1] += 1
    for (k, c), (v, n) in sorted(acc.items()):
        if "ksolve" in k:
            print(tag, k, c, "launches", n, "per launch", v / n)
            out.setdefault(k, {})[c] = {"sum": v, "launches": n, "per_launch": v / n}
json.dump(out, open(f"{O}/pmc_traffic_raw.json", "w"), indent=1)
PY
rm -rf $O/pmc_fetch $O/pmc_write $O/pmc_sq
find $O/stats -name "*kernel_stats*.csv" | head -1 | xargs -r -I{} cp {} $O/rocprofv3_kernel_stats_bench_1m.csv
rm -rf $O/stats
