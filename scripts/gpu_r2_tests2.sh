# the GPU parity tests of the build with volume requirement alternatives, complement minValues and the coop2 classing kernel
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r2e
mkdir -p $O
timeout 700 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log | cut -c1-400
