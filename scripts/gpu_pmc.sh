# HBM traffic of the pack kernel from the TCC counters (separate passes: FETCH_SIZE and WRITE_SIZE do not fit together,
# /opt/skills/guides/MI355X_MICROARCH.md "rocprofv3 PMC slots"); --kernel-trace only, no other trace domains.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$c -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --batch-problems 0 > $GRAFT_REPO_ROOT/gpurun_out/pmc_$c.log 2>&1)
  ls gpurun_out/pmc_$c
  f=$(find gpurun_out/pmc_$c -name "*counter_collection*.csv" | head -1)
  head -1 $f
  grep ksolve_pack $f | head -3
done
