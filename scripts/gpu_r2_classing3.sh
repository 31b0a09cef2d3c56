# classing kernel without the (all-nil) minValues tables, and its sensitivity to wavefronts per CU (unused LDS per block)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r2d
mkdir -p $O
timeout 400 python tests/tools/gpu_classing_ab.py 1000000 --digest-once coop2 coop2+2000 coop2+6000 coop2+11000 coop2+19000 coop2 coop1 > $O/classing_ab2.log 2>&1; tail -12 $O/classing_ab2.log
timeout 300 python -m pytest tests -m gpu -x -q -k "row_hash or config2_scaled or edge or daemons or topology_mix" > $O/pytest_gpu2.log 2>&1; tail -3 $O/pytest_gpu2.log
