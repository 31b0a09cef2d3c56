# phase counters of the general / BIG engine (profiling build) on the BASELINE configs[2] shape
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2
timeout 900 python - <<'PY' 2>&1 | tee gpurun_out/r2/c3_phases.log
import sys, time, json, os
from karpenter_amd import fixtures as fx
from karpenter_amd.scheduling import NewScheduler
names = ["queue","class_fetch","sort","scan","rec_load","can_add","commit","new_claim","dead_mark","try_sched","total","ca_pre","ca_merge","ca_total","ca_filter","f_ballots","f_combine","s_stage","s_headroom","s_select","x20","merge_reached","col_resets","full_filters"]
for label, prob in (("config3 100k", fx.config3(pods=100000, n_types=500, seed=42)), ("config3 200k", fx.config3(pods=200000, n_types=500, seed=42))):
    s = NewScheduler(prob, solver_lib=os.path.abspath("karpenter_amd/variants/libksolve_timers.so"))
    r = s.Solve(repeat=1, want_results=False)
    c = r["counters"]; pc = c["phaseCycles"]
    print(label, "pack ms", [t["pack_kernel_ms"] for t in r["timings"]], "pods", c["pods"], "claims", c["claims"], "evals", c["binEvaluations"], "slow", c["slowSorts"])
    for n, v in zip(names, pc): print("%-12s %14d cycles  %9.0f /pod" % (n, v, v / c["pods"]))
PY
