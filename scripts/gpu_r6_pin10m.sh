# round 6: the oracle pin of the EXACT configs[3] batch at full size (10M pods x 1000 types x 16 NodePools as ONE Solve(); ~1e11 bin
# evaluations) made on the GPU box's host cores — ORACLE_THREADS fans the in-flight scan out as the reference's parallelizeUntil does
# (oracle/scheduler.hpp add_to_inflight_parallel; same Results and counters as the sequential scan, tests/test_oracle_scheduling.py) —
# while the GPU runs the parity tests and smoke() of this build beside it.  usage (GPU box): bash scripts/gpu_r6_pin10m.sh [pods] [tag]
cd $GRAFT_REPO_ROOT
PODS=${1:-10000000}
T=${2:-pin10m}
O=$GRAFT_REPO_ROOT/gpurun_out/$T; mkdir -p $O
export TMPDIR=/tmp
NP=$(nproc); MEM=$(free -g | awk '/^Mem:/{print $7}')
echo "nproc $NP, available memory $MEM GB" | tee $O/box.log
lscpu | egrep "Model name|Thread|Core|Socket" | tee -a $O/box.log
if [ "$MEM" -lt 90 ]; then echo "not enough memory for the 10M oracle run"; exit 0; fi
TH=$((NP - 8)); [ $TH -gt 128 ] && TH=128; [ $TH -lt 1 ] && TH=1
(ORACLE_TIMING=1 ORACLE_THREADS=$TH PIN_OUT_DIR=$O timeout ${PIN_TIMEOUT:-2700} python tests/golden/make_fullsize_digests.py config4 $PODS 1000 42 16 > $O/oracle.log 2>&1; echo "oracle rc $?" >> $O/oracle.log) &
OP=$!
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 | tee $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $O/smoke.log
wait $OP
tail -3 $O/oracle.log | cut -c1-300
ls -la $O
