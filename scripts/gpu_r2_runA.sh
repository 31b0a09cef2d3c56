# round 2, run A: GPU test tier, default bench line, rocprofv3 kernel stats of the bench command, TCC traffic counters
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2a
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r2a
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1
tail -4 $O/pytest_gpu.log
timeout 1200 python bench.py > $O/bench_default.json 2> $O/bench_default.err
tail -c 1500 $O/bench_default.json; tail -3 $O/bench_default.err
# kernel stats of the same command, legs other than the headline switched off so the trace stays small
BENCH="python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --topology-pods 0 --batch-problems 0 --components-pods 0 --no-host-engine-baseline --no-cpu-baseline"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- $BENCH > $O/stats.log 2>&1)
tail -2 $O/stats.log
find $O/stats -name "*kernel_stats*.csv" | head -1 | xargs -r head -20
# HBM traffic of the pack kernel: FETCH_SIZE and WRITE_SIZE in separate passes (TCC slots), --kernel-trace only
(cd /tmp && timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o f -- $BENCH > $O/pmc_fetch.log 2>&1)
(cd /tmp && timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o w -- $BENCH > $O/pmc_write.log 2>&1)
python - $O <<'PY'
import csv, sys, glob, collections, json
O = sys.argv[1]
out = {}
for tag in ("fetch", "write"):
    fs = glob.glob(f"{O}/pmc_{tag}/**/*counter_collection*.csv", recursive=True)
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in fs:
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            acc[(k, r["Counter_Name"])][0] += float(r["Counter_Value"]); acc[(k, r["Counter_Name"])][1] += 1
    for (k, c), (v, n) in sorted(acc.items()):
        if "ksolve" in k:
            print(tag, k, c, "sum", v, "launches", n, "per launch", v / n)
            out.setdefault(k, {})[c] = {"sum": v, "launches": n, "per_launch": v / n}
json.dump(out, open(f"{O}/pmc_traffic_raw.json", "w"), indent=1)
PY
