# round 2 iteration loop: parity of the cursor engine vs the general engine on the device, SQ counters, phase counters
set -x
mkdir -p gpurun_out/r2
python tests/tools/gpu_engines_cmp.py 200000 500 > gpurun_out/r2/cmp_200k.log 2>&1
cat gpurun_out/r2/cmp_200k.log
bash scripts/gpu_sq_pmc.sh
bash scripts/gpu_fast_phases.sh | grep -v "^config2 200k\|upload_us"
