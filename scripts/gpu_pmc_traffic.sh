# HBM traffic of the bench's kernels for THIS build: TCC FETCH_SIZE and WRITE_SIZE in their own rocprofv3 --pmc passes (kernel-trace
# only — gpurun refuses counter collection combined with the other trace domains), FETCH_SIZE doubled on gfx950 as
# /opt/skills/guides/MI355X_MICROARCH.md prescribes. Writes profiles/round3/pmc_traffic.json with a hash of the kernel sources:
# bench.py quotes a traffic figure only when that hash equals the hash of the sources it runs. Also the rocprofv3 kernel stats
# of the same command. usage (on the GPU box): bash scripts/gpu_pmc_traffic.sh
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3pmc
mkdir -p $O
BENCH="python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --topology-pods 0 --batch-problems 0 --components-pods 0 --sweep-nodes 0 --no-host-engine-baseline --no-cpu-baseline"
(cd /tmp && timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- $BENCH > $O/stats.log 2>&1)
find $O/stats -name "*kernel_stats*.csv" | head -1 | xargs -r -I{} cp {} $O/rocprofv3_kernel_stats_bench_1m.csv
cut -c1-160 $O/rocprofv3_kernel_stats_bench_1m.csv | head -14
(cd /tmp && timeout 500 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o f -- $BENCH > $O/pmc_fetch.log 2>&1)
(cd /tmp && timeout 500 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o w -- $BENCH > $O/pmc_write.log 2>&1)
(cd /tmp && timeout 500 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/pmc_sq -o sq -- $BENCH > $O/pmc_sq.log 2>&1)
python - $O <<'PY'
import csv, sys, glob, collections, json, re, os
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import bench
O = sys.argv[1]
acc = collections.defaultdict(lambda: [0.0, 0])
for tag in ("fetch", "write", "sq"):
    for f in glob.glob(f"{O}/pmc_{tag}/**/*counter_collection*.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = re.sub(r"^void ", "", r["Kernel_Name"]).split("(")[0].split("<")[0]
            a = acc[(k, r["Counter_Name"])]
            a[0] += float(r["Counter_Value"]); a[1] += 1
kernels = {}
for (k, c), (v, n) in sorted(acc.items()):
    if "ksolve" in k:
        kernels.setdefault(k, {})[c] = {"per_launch": v / n, "launches": n}
for k, d in kernels.items():
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        d["traffic_bytes_per_launch"] = int(2 * d["FETCH_SIZE"]["per_launch"] * 1024 + d["WRITE_SIZE"]["per_launch"] * 1024)
out = {"source_sha": bench.source_sha(), "pods": 1000000, "types": 500, "command": "bench.py --steps 3 --warmup 1 (headline leg only)",
       "units": "FETCH_SIZE / WRITE_SIZE in KB per launch (rocprofv3 --pmc, separate passes); traffic = 2 x FETCH_SIZE + WRITE_SIZE (gfx950)", "kernels": kernels}
json.dump(out, open(f"{O}/pmc_traffic.json", "w"), indent=1)
for k in ("ksolve_pack_fast", "ksolve_row_hash_coop2"):
    print(k, {c: (round(v["per_launch"], 1) if isinstance(v, dict) else v) for c, v in kernels.get(k, {}).items()})
PY
rm -rf $O/pmc_fetch $O/pmc_write $O/pmc_sq $O/stats
