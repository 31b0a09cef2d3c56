# round 6: code-generation variants of the cursor engine's translation unit (scripts/build_unit_variant.sh ksolve_pack_fast <tag> -mllvm ...)
# on the headline leg alone (1M pods, digest checked against the pin in every run).   usage (GPU box): bash scripts/gpu_r6_fast_ab.sh <tag> <variant>...
cd $GRAFT_REPO_ROOT
T=$1; shift
O=$GRAFT_REPO_ROOT/gpurun_out/$T; mkdir -p $O
export TMPDIR=/tmp KSOLVE_BENCH_TEST_HOOK=1
B="python bench.py --steps 6 --warmup 2 --topology-pods 0 --batch-problems 0 --components-pods 0 --beyond-lds-pods 0 --whole-batch-exact-pods 0 --whole-batch-pods 0 --sweep-nodes 0 --no-host-engine-baseline --no-cpu-baseline"
for v in product "$@" product; do
  L=""; [ $v != product ] && L="--solver-lib $GRAFT_REPO_ROOT/karpenter_amd/variants/libksolve_$v.so"
  timeout 300 $B $L 2>$O/bench_$v.err | tail -1 > $O/bench_$v.json
  python -c "
import json,sys
d=json.load(open('$O/bench_$v.json')); print('$v', round(d['value']), round(d['ms_per_step'],2), d['parity']['oracle_pin'].get('digest_matches_oracle') if d['parity'].get('oracle_pin') else None)" | tee -a $O/ab.log
done
