# round 2: the GPU test tier and the default bench line on one box
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r2/pytest_gpu.log 2>&1
tail -5 gpurun_out/r2/pytest_gpu.log
timeout 1200 python bench.py > gpurun_out/r2/bench_default.json 2> gpurun_out/r2/bench_default.err
tail -c 3000 gpurun_out/r2/bench_default.json; tail -5 gpurun_out/r2/bench_default.err
