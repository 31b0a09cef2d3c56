# round 5, GPU pass R (what is left of the budget): more of the GPU parity tests that run the cursor engine on the last build
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5r; mkdir -p $O
export TMPDIR=/tmp
timeout 50 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "cursor_engine_moves or config4_components or full_size_properties or cancel or reference_known or cross_feature" 2>&1 | tail -6 | tee $O/pytest_subset.log
