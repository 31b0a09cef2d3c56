# round 3, second GPU run: the configs[2] pin at 300k / 500k pods made by the oracle with its candidate fan-out on the box's host
# cores (ORACLE_THREADS), then the default bench line with the new legs (configs[2] at 1M, configs[4] sweep)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3b
mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
T0=$(date +%s)
ORACLE_THREADS=128 PIN_OUT_DIR=$O timeout 900 python tests/golden/make_fullsize_digests.py config3 300000 500 42 > $O/pin300k.log 2>&1; tail -2 $O/pin300k.log
T1=$(date +%s); echo "pin 300k took $((T1-T0)) s"
if [ $((T1-T0)) -lt 420 ]; then
  ORACLE_THREADS=128 PIN_OUT_DIR=$O timeout 1300 python tests/golden/make_fullsize_digests.py config3 500000 500 42 > $O/pin500k.log 2>&1; tail -2 $O/pin500k.log
  T2=$(date +%s); echo "pin 500k took $((T2-T1)) s"
fi
cp $O/config3_p*.json tests/golden/fullsize/ 2>/dev/null
timeout 1200 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 2500 $O/bench_default.json; tail -5 $O/bench_default.err
