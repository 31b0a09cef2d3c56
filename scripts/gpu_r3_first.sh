# round 3, first GPU run: parity tests, the configs[4] sweep at size, the default bench line
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3a
mkdir -p $O
nproc > $O/nproc.txt; free -g | head -2 >> $O/nproc.txt
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -5 $O/pytest_gpu.log
timeout 900 python tests/tools/sweep_scale.py 20000 2000 16 > $O/sweep_20k.json 2> $O/sweep_20k.err; tail -c 1500 $O/sweep_20k.json; tail -3 $O/sweep_20k.err
timeout 1500 python tests/tools/sweep_scale.py 100000 10000 16 > $O/sweep_100k.json 2> $O/sweep_100k.err; tail -c 1500 $O/sweep_100k.json; tail -3 $O/sweep_100k.err
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 3000 $O/bench_default.json; tail -3 $O/bench_default.err
