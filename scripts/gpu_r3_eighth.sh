# round 3: BASELINE configs[4] at full size WITH topology pods in the cluster (two fifths of the default pool's pod templates carry
# zonal or hostname spread constraints): 100k nodes, ~2M bound pods (all of them pod rows of the resident base), 10k single-node
# probes, six of them re-simulated by the oracle over the whole cluster
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3k
mkdir -p $O
timeout 1000 python tests/tools/sweep_scale.py 100000 10000 6 --topology > $O/sweep_topology_100k.log 2>&1; tail -2 $O/sweep_topology_100k.log | cut -c1-1500
