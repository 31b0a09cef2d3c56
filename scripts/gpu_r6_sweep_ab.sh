# round 6: what the register spills of ksolve_pack_sweep4 cost. The shipped kernel (__launch_bounds__(256, 2): 256 VGPRs, 103 spilled) against
# the measurement build -DKSOLVE_SWEEP4_NO_SPILL (one wavefront per SIMD: 341 VGPRs, none spilled; scripts/build_unit_variant.sh) on the
# configs[4] legs of bench.py: kernel time from the line, TCC traffic of the kernel from two rocprofv3 --pmc passes each.
# usage (GPU box): bash scripts/gpu_r6_sweep_ab.sh <tag>
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $O
export TMPDIR=/tmp KSOLVE_BENCH_TEST_HOOK=1
B="python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --pods 20000 --no-parity-pin --topology-pods 0 --batch-problems 0 --components-pods 0 --beyond-lds-pods 0 --whole-batch-exact-pods 0 --whole-batch-pods 0 --no-host-engine-baseline --no-cpu-baseline"
VARIANTS="${KSOLVE_AB_VARIANTS:-product sweep4_nospill}"
for v in $VARIANTS; do
  L=""; [ $v != product ] && L="--solver-lib $GRAFT_REPO_ROOT/karpenter_amd/variants/libksolve_$v.so"
  timeout 600 $B $L 2>$O/bench_$v.err | tail -1 > $O/bench_$v.json
  (cd /tmp && timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_${v}_fetch -o f -- $B $L --sweep-sample 0 --sweep-topology-sample 0 --sweep-windows 0 > $O/pmc_${v}_fetch.log 2>&1)
  (cd /tmp && timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_${v}_write -o w -- $B $L --sweep-sample 0 --sweep-topology-sample 0 --sweep-windows 0 > $O/pmc_${v}_write.log 2>&1)
done
python - $O "$VARIANTS" <<'PY'
import json, sys, glob, csv
O = sys.argv[1]
out = {}
for v in sys.argv[2].split():
    d = json.load(open(f"{O}/bench_{v}.json"))
    s = d["config4_sweep"]
    row = {"single_node": {k: round(x * 1e3, 3) for k, x in s["seconds"].items()}, "pin": s.get("oracle_pin", {}).get("digest_matches_oracle")}
    t = s.get("with_topology_pods", {}); m = s.get("multi_node", {})
    if t: row["with_topology_pods"] = {k: round(x * 1e3, 3) for k, x in t.get("seconds", {}).items()}
    if m: row["multi_node"] = {k: round(x * 1e3, 3) for k, x in m.get("seconds", {}).items()}
    for tag in ("fetch", "write"):
        big = 0.0
        for f in glob.glob(f"{O}/pmc_{v}_{tag}/**/*counter_collection*.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if "ksolve_pack_sweep4" in r["Kernel_Name"]: big = max(big, float(r["Counter_Value"]))
        row[tag + "_kb_largest_launch"] = big
    row["traffic_bytes_largest_launch"] = int((2 * row["fetch_kb_largest_launch"] + row["write_kb_largest_launch"]) * 1024)
    out[v] = row
json.dump(out, open(f"{O}/sweep_spill_ab.json", "w"), indent=1)
print(json.dumps(out))
PY
