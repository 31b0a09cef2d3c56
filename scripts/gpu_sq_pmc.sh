# SQ counters of the pack kernel (one pass, 8 SQ slots; --kernel-trace only) on configs[1] at 1M pods
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/r2/sq_pmc
(cd /tmp && timeout 900 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $out -o pmc -- python $GRAFT_REPO_ROOT/tests/tools/gpu_engines_cmp.py ${1:-1000000} 500 --no-general > $out.log 2>&1)
tail -3 $out.log
f=$(find $out -name "*counter_collection*.csv" | head -1)
python - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for r in rows:
    if "ksolve_pack" in r["Kernel_Name"]:
        acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in acc.items():
    print(k, dict(v))
PY
