cd $GRAFT_REPO_ROOT
timeout 1500 python - <<'PY'
import glob, os
from karpenter_amd import fixtures as fx
from karpenter_amd.scheduling import NewScheduler
prob = fx.config2(pods=200000)
libs = sorted(glob.glob("karpenter_amd/variants/libksolve_*.so"))
res = {}
for rnd in range(2):
    for lib in libs:
        s = NewScheduler(prob, solver_lib=os.path.abspath(lib))
        r = s.Solve(repeat=2, want_results=False)
        res.setdefault(os.path.basename(lib), []).extend(round(t["pack_kernel_ms"]) for t in r["timings"])
        s.close()
for k, v in res.items(): print(k, v)
PY
