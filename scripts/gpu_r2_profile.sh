# round 2 evidence: rocprofv3 kernel stats of the bench command (headline leg only), SQ counters and TCC traffic of the same command in their own passes
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r2p
mkdir -p $O
BENCH="python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --topology-pods 0 --batch-problems 0 --components-pods 0 --no-host-engine-baseline --no-cpu-baseline"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- $BENCH > $O/stats.log 2>&1)
tail -1 $O/stats.log | cut -c1-600
find $O/stats -name "*kernel_stats*.csv" | head -1 | xargs -r cut -c1-150 | head -24
(cd /tmp && timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/pmc_sq -o sq -- $BENCH > $O/pmc_sq.log 2>&1)
(cd /tmp && timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o f -- $BENCH > $O/pmc_fetch.log 2>&1)
(cd /tmp && timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o w -- $BENCH > $O/pmc_write.log 2>&1)
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
json.dump(out, open(f"{O}/pmc_raw.json", "w"), indent=1)
PY
python - $O <<'PY'
import json, sys
O = sys.argv[1]
raw = json.load(open(O + "/pmc_raw.json"))
k = raw["ksolve_pack_fast"]
fetch_kb, write_kb = k["FETCH_SIZE"]["per_launch"], k["WRITE_SIZE"]["per_launch"]
out = {"kernel": "ksolve_pack_fast",
       "workload": "bench.py --steps 3 --warmup 1 (configs[1]: 1M pods x 500 types), rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes with --kernel-trace only (scripts/gpu_r2_profile.sh)",
       "fetch_kb_per_launch": fetch_kb, "write_kb_per_launch": write_kb,
       "traffic_bytes_per_launch": int(2 * fetch_kb * 1024 + write_kb * 1024),
       "note": "FETCH_SIZE doubled (gfx950: the counter tallies 128-byte requests at 64 bytes, MI355X_MICROARCH.md); WRITE_SIZE as reported"}
json.dump(out, open(O + "/pmc_pack_traffic.json", "w"), indent=1)
print(out)
PY
# lone-wave cost model, cursor-engine per-path counters, the resident-cluster consolidation sweep
(cd tests/tools/ubench && for b in branch_cost lds_latency; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $b $b.hip; done && ./branch_cost && ./lds_latency) > $O/ubench.log 2>&1; cat $O/ubench.log
bash scripts/gpu_fast_phases.sh | grep -v upload_us > $O/fast_phases.log 2>&1; tail -12 $O/fast_phases.log
timeout 900 python tests/tools/consolidation_sweep.py 10000 256 6 --types 500 --out $O/consolidation_sweep_10k_256.json > $O/sweep.log 2>&1
tail -c 1500 $O/sweep.log
