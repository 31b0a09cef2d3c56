# round 4, experiment pass (measurement only): the general engine with the topology groups' descriptors / small state in LDS
# (karpenter_amd/variants/libksolve_ldstopo.so, built from profiles/round4/experiments/lds_topology_state.patch) against the product
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4m; mkdir -p $O
timeout 120 python tests/tools/gpu_exp_ldstopo.py 60000 200000 2>$O/exp.err | tee $O/exp.jsonl
(cd /tmp && timeout 60 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $O/p -o a -- python $GRAFT_REPO_ROOT/tests/tools/gpu_exp_ldstopo.py 60000 > $O/pmc.log 2>&1)
python - $O <<'PY'
import csv, glob, sys, collections, json
O = sys.argv[1]
rows = collections.defaultdict(dict)
for f in glob.glob(f"{O}/p/**/*counter_collection*.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "ksolve_pack" in r["Kernel_Name"] and "fast" not in r["Kernel_Name"]:
            rows[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
out = [dict(dispatch=k, **v) for k, v in sorted(rows.items())]
json.dump(out, open(f"{O}/pmc_by_dispatch.json", "w"), indent=1)
for o in out: print(o)
PY
rm -rf $O/p
