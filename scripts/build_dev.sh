# builds the product library and the phase-timer variant; exits non-zero when either fails
set -e
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -o karpenter_amd/libksolve.so.tmp karpenter_amd/csrc/ksolve.hip &
p1=$!
mkdir -p karpenter_amd/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DKSOLVE_PHASE_TIMERS -o karpenter_amd/variants/libksolve_timers.so.tmp karpenter_amd/csrc/ksolve.hip &
p2=$!
g++ -O2 -std=c++17 -fPIC -shared -o karpenter_amd/libksched.so.tmp karpenter_amd/host/ksched.cpp -ldl
wait $p1; wait $p2
mv karpenter_amd/libksched.so.tmp karpenter_amd/libksched.so
mv karpenter_amd/libksolve.so.tmp karpenter_amd/libksolve.so
mv karpenter_amd/variants/libksolve_timers.so.tmp karpenter_amd/variants/libksolve_timers.so
echo built
