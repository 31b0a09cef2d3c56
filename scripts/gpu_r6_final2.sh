# round 6, last pass: the complete final pass (scripts/gpu_r6_final.sh) on the last sources, THEN — with the GPU minutes that would otherwise
# be lost — a second attempt at the oracle pin of the exact configs[3] batch at 10M pods on the box's host cores: one socket this time
# (64 threads bound to NUMA node 0: the first attempt spread 128 threads over both sockets and did not finish in 45 minutes), with a progress
# line every 250k pods.   usage (GPU box): bash scripts/gpu_r6_final2.sh [tag]
cd $GRAFT_REPO_ROOT
T=${1:-final2}
bash scripts/gpu_r6_final.sh $T 2>&1 | tail -40
O=$GRAFT_REPO_ROOT/gpurun_out/${T}_pin10m; mkdir -p $O
lscpu | egrep "NUMA|Socket|Core|Thread" > $O/box.log
BIND=""
if command -v numactl >/dev/null 2>&1; then BIND="numactl --cpunodebind=0 --membind=0"; else BIND="taskset -c 0-63"; fi
echo "bind: $BIND" >> $O/box.log
ORACLE_TIMING=1 ORACLE_PROGRESS=250000 ORACLE_THREADS=${PIN_THREADS:-64} PIN_OUT_DIR=$O timeout ${PIN_TIMEOUT:-4200} $BIND python tests/golden/make_fullsize_digests.py config4 ${PIN_PODS:-10000000} 1000 42 16 > $O/oracle.log 2>&1
echo "oracle rc $?" >> $O/oracle.log
tail -5 $O/oracle.log | cut -c1-300
ls -la $O
