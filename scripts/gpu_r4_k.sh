# round 4, tenth GPU pass: A/B of two builds of the compact sweep kernel (rare probe paths inlined / as real calls), same box
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4k; mkdir -p $O
export KSOLVE_TEST_SOLVER_LIB=1
for v in prev outl prev outl; do
  timeout 200 python tests/tools/sweep_scale.py 100000 10000 0 --repeat 4 --solver-lib karpenter_amd/variants/libksolve_$v.so 2>>$O/err.log | tail -1 >> $O/sweep_$v.jsonl
done
for v in prev outl; do
  timeout 200 python tests/tools/sweep_scale.py 100000 10000 0 --topology --repeat 4 --solver-lib karpenter_amd/variants/libksolve_$v.so 2>>$O/err.log | tail -1 >> $O/sweep_${v}_topology.jsonl
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r4k/*.jsonl")):
    for line in open(f):
        d = json.loads(line)
        print(f.split("/")[-1], d["verdict_digest"], {k: round(d["timings"][k], 1) for k in ("pack_us", "sweep_ms", "upload_us")})
PY
