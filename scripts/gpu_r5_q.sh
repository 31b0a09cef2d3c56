# round 5, GPU pass Q (the round's last GPU minute): the GPU parity tests that drive the cursor engine's small problems through the
# last build's fast loop (n in 13..49 claims: pdqsort's other paths on every move; n <= 12; batched launches; the wave pair)
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5q; mkdir -p $O
export TMPDIR=/tmp
timeout 85 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "config1_5000 or config2_scaled or edge_cases or two_wavefront or batched_launch" 2>&1 | tail -6 | tee $O/pytest_subset.log
