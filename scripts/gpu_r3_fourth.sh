# round 3, final build: TCC / SQ counters + rocprofv3 kernel stats of the headline leg (scripts/gpu_pmc_traffic.sh), then the GPU parity tests
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
bash scripts/gpu_pmc_traffic.sh
O=$GRAFT_REPO_ROOT/gpurun_out/r3g
mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log
