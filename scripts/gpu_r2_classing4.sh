# classing kernel: 60 rows per block (8 blocks per CU) against 64 (7), then the parity tests that exercise small and ragged row counts
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r2d
mkdir -p $O
timeout 300 python tests/tools/gpu_classing_ab.py 1000000 coop2@60 coop2@64 coop2@60 coop2@64 > $O/classing_ab3.log 2>&1; tail -9 $O/classing_ab3.log
timeout 200 python -m pytest tests -m gpu -x -q -k "row_hash or config2_scaled or edge or config1 or known_answers or volume" > $O/pytest_gpu3.log 2>&1; tail -2 $O/pytest_gpu3.log | cut -c1-200
