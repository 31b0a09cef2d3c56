# round 5, GPU pass D: the cursor engine's fast loop with the next pod's order reads pipelined behind the current pod's writes:
# pins, reduced bench, SQ counters, kernel stats, the exact configs[3] batch at 10M pods (plan 2, four rows of class slots)
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5d; mkdir -p $O
export TMPDIR=/tmp
for pin in config2_p200000_t500_s42 config2_p1000000_t500_s42; do
  for eng in auto cursor-wide cursor-hbm; do timeout 300 python tests/tools/gpu_check_pin.py tests/golden/fullsize/$pin.json $eng 2>&1 | tail -1 | tee -a $O/pins.log; done
done
timeout 900 python bench.py --steps 5 --topology-pods 0 --components-pods 0 --beyond-lds-pods 0 --whole-batch-exact-pods 0 --whole-batch-pods 0 --batch-problems 0 --sweep-nodes 0 --no-cpu-baseline --no-host-engine-baseline 2>$O/bench_reduced.err | tail -1 > $O/bench_reduced.json
tail -3 $O/bench_reduced.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r5d/bench_reduced.json"))
print("value", d["value"], "ms_per_step", d["ms_per_step"], "pack ms", d["pack_kernel"]["avg_kernel_ms"])
PY
HEAD="python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --topology-pods 0 --batch-problems 0 --components-pods 0 --beyond-lds-pods 0 --whole-batch-exact-pods 0 --whole-batch-pods 0 --sweep-nodes 0 --no-host-engine-baseline --no-cpu-baseline"
(cd /tmp && timeout 400 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_sq -o sq -- $HEAD > $GRAFT_REPO_ROOT/$O/pmc_sq.log 2>&1)
(cd /tmp && timeout 400 rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_sq2 -o sq -- $HEAD > $GRAFT_REPO_ROOT/$O/pmc_sq2.log 2>&1)
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob("gpurun_out/r5d/pmc_sq*/**/*counter_collection*.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "pack_fast" in r["Kernel_Name"]:
            a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
print({k: round(v[0] / v[1]) for k, v in acc.items()})
PY
rm -rf $O/pmc_sq $O/pmc_sq2
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "edge or cursor or batch or population" 2>&1 | tail -5 | tee $O/pytest_subset.log
