# Round 5: HBM traffic (TCC FETCH_SIZE / WRITE_SIZE, separate rocprofv3 --pmc passes with --kernel-trace only; FETCH_SIZE doubled on
# gfx950 as /opt/skills/guides/MI355X_MICROARCH.md prescribes) and wave counts of (a) the headline leg's kernels and (b) the kernels of
# the consolidation sweep (ksolve_pack_sweep, ksolve_node_dead0) for THIS build -> gpurun_out/r5pmc/pmc_traffic.json (copy it to
# profiles/round5/; bench.py quotes a traffic figure only when the source hash matches), plus rocprofv3 kernel stats of both commands
# and the classing kernel at 1M / 2M / 4M rows (beyond the 256 MiB Infinity Cache).   usage (GPU box): bash scripts/gpu_r5_pmc.sh
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5pmc
mkdir -p $O
HEAD="python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --topology-pods 0 --batch-problems 0 --components-pods 0 --beyond-lds-pods 0 --whole-batch-exact-pods 0 --whole-batch-pods 0 --sweep-nodes 0 --no-host-engine-baseline --no-cpu-baseline"
SWEEP="python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --pods 20000 --no-parity-pin --topology-pods 0 --batch-problems 0 --components-pods 0 --beyond-lds-pods 0 --sweep-sample 0 --sweep-topology-sample 0 --sweep-windows 0 --no-host-engine-baseline --no-cpu-baseline"
(cd /tmp && timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_head -o bench -- $HEAD > $O/stats_head.log 2>&1)
find $O/stats_head -name "*kernel_stats*.csv" | head -1 | xargs -r -I{} cp {} $O/rocprofv3_kernel_stats_bench_1m.csv
[ "${KSOLVE_PMC_LEGS:-head sweep}" = head ] || (cd /tmp && timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_sweep -o bench -- $SWEEP > $O/stats_sweep.log 2>&1)
find $O/stats_sweep -name "*kernel_stats*.csv" | head -1 | xargs -r -I{} cp {} $O/rocprofv3_kernel_stats_sweep.csv
cut -c1-150 $O/rocprofv3_kernel_stats_bench_1m.csv | head -8; cut -c1-150 $O/rocprofv3_kernel_stats_sweep.csv | head -10
for leg in ${KSOLVE_PMC_LEGS:-head sweep}; do
  CMD="$HEAD"; [ $leg = sweep ] && CMD="$SWEEP"
  (cd /tmp && timeout 500 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_${leg}_fetch -o f -- $CMD > $O/pmc_${leg}_fetch.log 2>&1)
  (cd /tmp && timeout 500 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_${leg}_write -o w -- $CMD > $O/pmc_${leg}_write.log 2>&1)
  (cd /tmp && timeout 500 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmc_${leg}_sq -o sq -- $CMD > $O/pmc_${leg}_sq.log 2>&1)
  (cd /tmp && timeout 500 rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $O/pmc_${leg}_sq2 -o sq -- $CMD > $O/pmc_${leg}_sq2.log 2>&1)
done
python - $O <<'PY'
import csv, sys, glob, collections, json, re, os
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import bench
O = sys.argv[1]
def collect(leg):
    acc = collections.defaultdict(lambda: [0.0, 0])
    big = collections.defaultdict(float)
    for tag in ("fetch", "write", "sq", "sq2"):
        for f in glob.glob(f"{O}/pmc_{leg}_{tag}/**/*counter_collection*.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                k = re.sub(r"^void ", "", r["Kernel_Name"]).split("(")[0].split("<")[0]
                a = acc[(k, r["Counter_Name"])]
                a[0] += float(r["Counter_Value"]); a[1] += 1
                big[(k, r["Counter_Name"])] = max(big[(k, r["Counter_Name"])], float(r["Counter_Value"]))
    kernels = {}
    for (k, c), (v, n) in sorted(acc.items()):
        if "ksolve" in k:
            kernels.setdefault(k, {})[c] = {"per_launch": v / n, "launches": n, "largest_launch": big[(k, c)]}
    for k, d in kernels.items():
        if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
            d["traffic_bytes_per_launch"] = int(2 * d["FETCH_SIZE"]["per_launch"] * 1024 + d["WRITE_SIZE"]["per_launch"] * 1024)
            # the command's launches of one kernel differ in size (a warm-up sweep of 64 probes before the 10k-probe one): the
            # largest launch is the one the bench line times
            d["traffic_bytes_largest_launch"] = int(2 * d["FETCH_SIZE"]["largest_launch"] * 1024 + d["WRITE_SIZE"]["largest_launch"] * 1024)
    return kernels
out = {"source_sha": bench.source_sha(), "pods": 1000000, "types": 500, "command": "bench.py --steps 3 --warmup 1 (headline leg only); sweep_kernels: the configs[4] leg (100k nodes; a warm-up sweep of 64 probes, then the 10k single-node probes the bench line times = largest_launch)",
       "units": "FETCH_SIZE / WRITE_SIZE in KB per launch (rocprofv3 --pmc, separate passes); traffic = 2 x FETCH_SIZE + WRITE_SIZE (gfx950); per_launch = mean over the launches of that kernel in the command",
       "kernels": collect("head"), "sweep_kernels": collect("sweep")}
if os.environ.get("KSOLVE_PMC_LEGS", "head sweep") == "head":
    # the headline leg alone was re-measured: the sweep kernels' counters are carried over from the committed file, which says on which
    # build they were taken (a build that differs from this one in files no sweep kernel instantiates — stated by whoever ran this)
    prev = json.load(open(os.path.join(os.environ["GRAFT_REPO_ROOT"], "profiles", "round5", "pmc_traffic.json")))
    out["sweep_kernels"] = prev.get("sweep_kernels", {})
    out["sweep_kernels_measured_on_source_sha"] = prev.get("sweep_kernels_measured_on_source_sha", prev.get("source_sha"))
    out["sweep_kernels_note"] = os.environ.get("KSOLVE_PMC_SWEEP_NOTE", "")
json.dump(out, open(f"{O}/pmc_traffic.json", "w"), indent=1)
for grp, names in (("kernels", ("ksolve_pack_fast", "ksolve_row_hash_coop2")), ("sweep_kernels", ("ksolve_pack_sweep", "ksolve_node_dead0"))):
    for k in names:
        print(k, {c: (round(v["per_launch"], 1) if isinstance(v, dict) else v) for c, v in out[grp].get(k, {}).items()})
PY
[ -n "$KSOLVE_PMC_SKIP_CLASSING_ROWS" ] || { timeout 600 python tests/tools/gpu_classing_rows.py > $O/classing_rows.json 2> $O/classing_rows.err; cat $O/classing_rows.json; }
rm -rf $O/pmc_*_fetch $O/pmc_*_write $O/pmc_*_sq $O/pmc_*_sq2 $O/stats_head $O/stats_sweep
