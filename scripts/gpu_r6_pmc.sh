# Round 6: HBM traffic (TCC FETCH_SIZE / WRITE_SIZE, separate rocprofv3 --pmc passes with --kernel-trace only; FETCH_SIZE doubled on
# gfx950 as /opt/skills/guides/MI355X_MICROARCH.md prescribes), wave counts and SQ instruction / cycle counters of THIS build for
#   kernels            the headline leg (bench.py: ksolve_pack_fast_g0r1, ksolve_row_hash_coop2, ...)
#   topology_kernels   the configs[2] leg at 1M pods: the spread engine's ksolve_pack_topo — and, on the 200k-pod pin, the general / BIG
#                      engine's ksolve_pack_big (the kernel the spread engine replaces there)
#   sweep_kernels      the consolidation sweep (ksolve_pack_sweep4, ksolve_node_dead0)
#   exact_kernels      KSOLVE_PMC_LEGS contains "exact": the exact configs[3] batch (10M pods as ONE Solve(): ksolve_pack_fast_g2r4)
# -> gpurun_out/r6pmc/pmc_traffic.json (copy it to profiles/round6/; bench.py quotes a figure only when the source hash matches), plus
# rocprofv3 kernel stats of the commands.   usage (GPU box): bash scripts/gpu_r6_pmc.sh      KSOLVE_PMC_LEGS="head topo big sweep exact"
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r6pmc
mkdir -p $O
LEGS=${KSOLVE_PMC_LEGS:-head topo big sweep}
B="python $GRAFT_REPO_ROOT/bench.py --batch-problems 0 --components-pods 0 --beyond-lds-pods 0 --whole-batch-exact-pods 0 --whole-batch-pods 0 --no-host-engine-baseline --no-cpu-baseline"
declare -A CMD
CMD[head]="$B --steps 3 --warmup 1 --topology-pods 0 --sweep-nodes 0"
CMD[topo]="$B --steps 1 --warmup 0 --pods 20000 --no-parity-pin --topology-pods 1000000 --sweep-nodes 0"
CMD[big]="python $GRAFT_REPO_ROOT/tests/tools/gpu_check_pin.py $GRAFT_REPO_ROOT/tests/golden/fullsize/config3_p200000_t500_s42.json general"
CMD[sweep]="$B --steps 1 --warmup 0 --pods 20000 --no-parity-pin --topology-pods 0 --sweep-sample 0 --sweep-topology-sample 0 --sweep-windows 0"
CMD[exact]="python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --pods 20000 --no-parity-pin --topology-pods 0 --batch-problems 0 --components-pods 10000000 --beyond-lds-pods 0 --whole-batch-pods 0 --sweep-nodes 0 --no-host-engine-baseline --no-cpu-baseline"
for leg in $LEGS; do
  C="${CMD[$leg]}"
  (cd /tmp && timeout 700 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$leg -o bench -- $C > $O/stats_$leg.log 2>&1)
  find $O/stats_$leg -name "*kernel_stats*.csv" | head -1 | xargs -r -I{} cp {} $O/rocprofv3_kernel_stats_$leg.csv
  cut -c1-150 $O/rocprofv3_kernel_stats_$leg.csv | head -6
  (cd /tmp && timeout 700 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_${leg}_fetch -o f -- $C > $O/pmc_${leg}_fetch.log 2>&1)
  (cd /tmp && timeout 700 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_${leg}_write -o w -- $C > $O/pmc_${leg}_write.log 2>&1)
  (cd /tmp && timeout 700 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmc_${leg}_sq -o sq -- $C > $O/pmc_${leg}_sq.log 2>&1)
  (cd /tmp && timeout 700 rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $O/pmc_${leg}_sq2 -o sq -- $C > $O/pmc_${leg}_sq2.log 2>&1)
done
python - $O "$LEGS" <<'PY'
import csv, sys, glob, collections, json, re, os
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import bench
O, legs = sys.argv[1], sys.argv[2].split()
def collect(leg, want=None):
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
        if "ksolve" in k and (want is None or any(k.startswith(w) for w in want)):
            kernels.setdefault(k, {})[c] = {"per_launch": v / n, "launches": n, "largest_launch": big[(k, c)]}
    for k, d in kernels.items():
        if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
            d["traffic_bytes_per_launch"] = int(2 * d["FETCH_SIZE"]["per_launch"] * 1024 + d["WRITE_SIZE"]["per_launch"] * 1024)
            d["traffic_bytes_largest_launch"] = int(2 * d["FETCH_SIZE"]["largest_launch"] * 1024 + d["WRITE_SIZE"]["largest_launch"] * 1024)
    return kernels
out = {"source_sha": bench.source_sha(), "pods": 1000000, "types": 500, "legs": legs,
       "commands": "head: bench.py --steps 3 --warmup 1 (headline leg only); topo: bench.py --topology-pods 1000000 (the configs[2] leg; its 20k-pod headline problem is not the timed one); "
                   "big: tests/tools/gpu_check_pin.py config3 200k pods, general engine; sweep: the configs[4] leg (100k nodes; a warm-up sweep of 64 probes, then the 10k single-node probes = largest_launch); "
                   "exact: bench.py --whole-batch-exact-pods 10000000 (the exact configs[3] batch as ONE Solve())",
       "units": "FETCH_SIZE / WRITE_SIZE in KB per launch (rocprofv3 --pmc, separate passes); traffic = 2 x FETCH_SIZE + WRITE_SIZE (gfx950); per_launch = mean over the launches of that kernel in the command; SQ_WAVE_CYCLES / SQ_WAIT_ANY / SQ_ACTIVE_INST_ANY count in units of four shader cycles"}
prev = {}
try:
    prev = json.load(open(os.path.join(os.environ["GRAFT_REPO_ROOT"], "profiles", "round6", "pmc_traffic.json")))
except (OSError, ValueError):
    pass
for leg, group, want in (("head", "kernels", None), ("sweep", "sweep_kernels", None), ("exact", "exact_kernels", ("ksolve_pack_fast",))):
    if leg in legs:
        out[group] = collect(leg, want)
    elif group in prev:
        out[group] = prev[group]; out[group + "_measured_on_source_sha"] = prev.get(group + "_measured_on_source_sha", prev.get("source_sha"))
    else:
        out[group] = {}
tk = {}
if "topo" in legs: tk.update(collect("topo", ("ksolve_pack_topo",)))
if "big" in legs: tk.update(collect("big", ("ksolve_pack_big",)))
if not tk and "topology_kernels" in prev:
    tk = prev["topology_kernels"]; out["topology_kernels_measured_on_source_sha"] = prev.get("topology_kernels_measured_on_source_sha", prev.get("source_sha"))
out["topology_kernels"] = tk
if "ksolve_pack_big" in tk: tk["ksolve_pack_big"]["pods"] = 200000
json.dump(out, open(f"{O}/pmc_traffic.json", "w"), indent=1)
for grp in ("kernels", "topology_kernels", "sweep_kernels", "exact_kernels"):
    for k, d in out.get(grp, {}).items():
        if "pack" in k or "row_hash_coop2" in k or "dead0" in k:
            print(grp, k, {c: (round(v["per_launch"], 1) if isinstance(v, dict) else v) for c, v in d.items()})
PY
rm -rf $O/pmc_*_fetch $O/pmc_*_write $O/pmc_*_sq $O/pmc_*_sq2 $O/stats_*
