# pack-kernel time of a variant build of the device library on configs[1] (timing experiments; the results of an ablated build are not checked)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2
for lib in "$@"; do
timeout 600 python - "$lib" <<'PY' 2>&1 | tee -a gpurun_out/r2/variant_time.log
import sys, os
from karpenter_amd import fixtures as fx
from karpenter_amd.scheduling import NewScheduler
lib = os.path.abspath(sys.argv[1])
s = NewScheduler(fx.config2(pods=1000000), solver_lib=lib)
r = s.Solve(repeat=3, want_results=False)
c = r["counters"]
print(os.path.basename(lib), c["engine"], "claims", c["claims"], "pack ms", [round(t["pack_kernel_ms"], 1) for t in r["timings"]], "steps", c["phaseCycles"][21])
PY
done
