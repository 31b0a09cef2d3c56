# round 6, passes o..: the sweep legs of bench.py on the current build + the TCC traffic of ksolve_pack_sweep4 (product only).
# usage (GPU box): bash scripts/gpu_r6_o.sh <tag>
cd $GRAFT_REPO_ROOT
T=${1:-r6o}
export TMPDIR=/tmp
bash scripts/gpu_r6_sweep.sh $T 2>&1 | tail -6
KSOLVE_AB_VARIANTS=product bash scripts/gpu_r6_sweep_ab.sh $T 2>&1 | tail -1
