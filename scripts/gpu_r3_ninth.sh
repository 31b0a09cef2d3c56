# round 3: the 300k-pod pin of the configs[2] shape (made offline by the single-thread oracle) against the device
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3l
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "full_size_digest" > $O/pytest_fullsize.log 2>&1; tail -4 $O/pytest_fullsize.log
