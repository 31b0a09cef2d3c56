#!/bin/bash
# Runs the device-algorithm tests with the host flattener and the host emulation of the engine built under AddressSanitizer.
# The engine headers are the device's own source (tests/emu compiles them for the host), so an out-of-bounds access found
# here is an out-of-bounds access of the GPU kernel. Usage: bash scripts/asan_check.sh  (≈12 min; restores the normal builds)
set -e
cd "$(dirname "$0")/.."
ASAN=$(ls /usr/lib/gcc/x86_64-linux-gnu/*/libasan.so | head -1)
STD=/usr/lib/x86_64-linux-gnu/libstdc++.so.6     # python does not link libstdc++: preload it so ASan can intercept __cxa_throw
FLAGS="-O1 -g -fsanitize=address -fsanitize-recover=address -fno-omit-frame-pointer -std=c++17 -fPIC -shared"
python -c "import __graft_entry__ as g; g.build()"; python -c "import sys; sys.path.insert(0, 'tests'); import parity; parity.build_emu()"
cp karpenter_amd/libksched.so /tmp/libksched.normal.so; cp tests/emu/libksolve_emu.so /tmp/libksolve_emu.normal.so
restore() { cp /tmp/libksched.normal.so karpenter_amd/libksched.so; cp /tmp/libksolve_emu.normal.so tests/emu/libksolve_emu.so; }
trap restore EXIT
g++ $FLAGS -o karpenter_amd/libksched.so karpenter_amd/host/ksched.cpp -ldl
g++ $FLAGS -pthread -o tests/emu/libksolve_emu.so tests/emu/ksolve_emu.cpp
LD_PRELOAD="$ASAN $STD" ASAN_OPTIONS=detect_leaks=0:halt_on_error=0 python -m pytest tests/test_device_algorithm.py tests/test_device_topology.py \
  tests/test_device_fuzz_all.py tests/test_cursor_engine.py tests/test_spread_engine.py tests/test_disruption.py tests/test_reference_known_answers.py -q -s -p no:cacheprovider > /tmp/asan_pytest.log 2>&1 || true
tail -1 /tmp/asan_pytest.log
echo "AddressSanitizer reports: $(grep -c 'ERROR: AddressSanitizer' /tmp/asan_pytest.log)"
grep SUMMARY /tmp/asan_pytest.log | sort | uniq -c
