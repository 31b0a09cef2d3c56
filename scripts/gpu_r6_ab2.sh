cd $GRAFT_REPO_ROOT
O=gpurun_out/r6y; mkdir -p $O
export TMPDIR=/tmp KSOLVE_BENCH_TEST_HOOK=1 KSOLVE_TEST_SOLVER_LIB=1
for v in product topo_maxilp product; do
  L=karpenter_amd/libksolve.so; [ $v != product ] && L=karpenter_amd/variants/libksolve_$v.so
  KSOLVE_LIB=$GRAFT_REPO_ROOT/$L timeout 300 python tests/tools/gpu_check_pin.py tests/golden/fullsize/config3_p1000000_t500_s42.json spread 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['pack_kernel_ms'], d['digest_matches'])" | tee -a $O/ab_topo.log
done
B="python bench.py --steps 1 --warmup 0 --pods 20000 --no-parity-pin --topology-pods 0 --batch-problems 0 --components-pods 0 --beyond-lds-pods 0 --whole-batch-exact-pods 0 --whole-batch-pods 0 --no-host-engine-baseline --no-cpu-baseline --sweep-windows 0 --sweep-topology-sample 0"
for v in product sweep4_maxilp product; do
  L=""; [ $v != product ] && L="--solver-lib $GRAFT_REPO_ROOT/karpenter_amd/variants/libksolve_$v.so"
  timeout 300 $B $L 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); s=d['config4_sweep']['seconds']; print('$v', round(s['pack_kernel']*1e3,3), round(s['library_call']*1e3,3))" | tee -a $O/ab_sweep.log
done
