# round 3: phase counters of the general / BIG engine on the configs[2] shape (profiling build), then the GPU parity tests incl. the
# multi-node consolidation windows
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3i
mkdir -p $O
bash scripts/gpu_c3_phases.sh > $O/c3_phases.log 2>&1; grep -v "^\[" $O/c3_phases.log | tail -28
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
