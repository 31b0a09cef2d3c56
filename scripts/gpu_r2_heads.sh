# ready-heads iteration: digest + pack time of the cursor engine on configs[1] (200k with the general engine beside it, 1M alone), phase counters
set -x
mkdir -p gpurun_out/r2
python tests/tools/gpu_engines_cmp.py 200000 500 > gpurun_out/r2/cmp_200k.log 2>&1
cat gpurun_out/r2/cmp_200k.log
python tests/tools/gpu_engines_cmp.py 1000000 500 --no-general > gpurun_out/r2/cmp_1m.log 2>&1
cat gpurun_out/r2/cmp_1m.log
bash scripts/gpu_fast_phases.sh | grep -v "upload_us"
