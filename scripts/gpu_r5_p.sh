# round 5, GPU pass P: the fast loop at 179 instructions per pod (no per-lane sentinel behind the order read, scalar minimum in the sampled-position test) — the three full-size pins it
# touches, PMC + kernel stats of the HEADLINE leg of this build (the sweep kernels' counters stay those of the final pass: this build
# differs from that one in csrc/fast_engine.h only, which no sweep kernel instantiates), the headline bench line
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5p; mkdir -p $O
export TMPDIR=/tmp
timeout 120 python tests/tools/gpu_check_pin.py tests/golden/fullsize/config2_p1000000_t500_s42.json auto 2>&1 | tail -1 | tee -a $O/pins.log
timeout 120 python tests/tools/gpu_check_pin.py tests/golden/fullsize/config4_p1000000_t1000_s42_x16.json auto 2>&1 | tail -1 | tee -a $O/pins.log
timeout 120 python tests/tools/gpu_check_pin.py tests/golden/fullsize/config2_p2000000_t500_s42.json auto 2>&1 | tail -1 | tee -a $O/pins.log
for eng in auto cursor-hbm; do timeout 200 python tests/tools/gpu_check_pin.py tests/golden/fullsize/config4_p2000000_t1000_s42_x16.json $eng 2>&1 | tail -1 | tee -a $O/pins.log; done   # the new pin of the x16 shape at 2M pods (5,498 claims, four rows of class slots, plans 1 and 2)
KSOLVE_PMC_LEGS=head KSOLVE_PMC_SKIP_CLASSING_ROWS=1 KSOLVE_PMC_SWEEP_NOTE="sweep kernels measured in the final pass (profiles/round5/final/); this build differs from that one in karpenter_amd/csrc/fast_engine.h only, which no sweep kernel instantiates" timeout 400 bash scripts/gpu_r5_pmc.sh 2>&1 | tail -4
cp gpurun_out/r5pmc/pmc_traffic.json profiles/round5/pmc_traffic.json
cp gpurun_out/r5pmc/pmc_traffic.json gpurun_out/r5pmc/rocprofv3_kernel_stats_bench_1m.csv $O/
timeout 200 python bench.py --steps 5 --topology-pods 0 --components-pods 0 --beyond-lds-pods 0 --whole-batch-exact-pods 0 --whole-batch-pods 0 --batch-problems 0 --sweep-nodes 0 --no-cpu-baseline 2>$O/bench_reduced.err | tail -1 > $O/bench_reduced.json
tail -2 $O/bench_reduced.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r5p/bench_reduced.json"))
print("value", d["value"], "ms", d["ms_per_step"], "pack", d["pack_kernel"]["avg_kernel_ms"], d["pack_kernel"].get("sq_counters"), d["roofline"].get("traffic"))
print(d.get("cpu_baseline_engine_host"))
PY
