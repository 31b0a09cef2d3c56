# classing kernel coop2 (one round trip per block, parallel leaders) against coop1 / plain at 1M rows, then the GPU parity tests
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r2d
mkdir -p $O
timeout 400 python tests/tools/gpu_classing_ab.py 1000000 > $O/classing_ab.log 2>&1; tail -12 $O/classing_ab.log
timeout 600 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log
