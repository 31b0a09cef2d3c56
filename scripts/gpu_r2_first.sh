# round 2, first GPU contact of the cursor engine: parity against the general engine on the device + timing
set -x
mkdir -p gpurun_out/r2
python tests/tools/gpu_engines_cmp.py 20000 144 > gpurun_out/r2/cmp_20k.log 2>&1
python tests/tools/gpu_engines_cmp.py 200000 500 > gpurun_out/r2/cmp_200k.log 2>&1
python tests/tools/gpu_engines_cmp.py 1000000 500 --no-general > gpurun_out/r2/cmp_1m.log 2>&1
cat gpurun_out/r2/cmp_*.log
