# round 4, first GPU pass: parity tests (sweep verdicts now judged by oracle/consolidation.hpp), smoke, bench line
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4a; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 | tee $O/pytest_gpu.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4 | tee $O/smoke.log
timeout 1500 python bench.py --steps 3 --warmup 1 2>$O/bench.err | tail -1 | tee $O/bench.json
