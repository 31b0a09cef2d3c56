# round 4, second GPU pass: parity tests (500k pin, hooks build), smoke, the default bench line (invariants at 1M, topology sweep leg,
# by-position re-hydration, sweep-kernel rooflines), then the PMC / kernel-stats passes of scripts/gpu_r4_pmc.sh
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4b; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 | tee $O/pytest_gpu.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4 | tee $O/smoke.log
timeout 1800 python bench.py 2>$O/bench.err | tail -1 | tee $O/bench.json
tail -5 $O/bench.err
bash scripts/gpu_r4_pmc.sh 2>&1 | tail -30
