# Profiling build of the device library with the per-phase shader-clock counters compiled in (the product build
# compiles them out, wave.h). Used by scripts/gpu_quick.sh; never loaded by the product.
cd "$(dirname "$0")/.." && mkdir -p karpenter_amd/variants && \
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DKSOLVE_PHASE_TIMERS -o karpenter_amd/variants/libksolve_timers.so karpenter_amd/csrc/ksolve.hip
