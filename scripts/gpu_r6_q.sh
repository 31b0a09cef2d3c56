# round 6, pass q: sweep legs + traffic of the current build, then the per-probe phase breakdown on the profiling build of the same sources.
cd $GRAFT_REPO_ROOT
T=${1:-r6q}
export TMPDIR=/tmp
bash scripts/gpu_r6_o.sh $T
bash scripts/gpu_r6_probe_costs.sh $T
