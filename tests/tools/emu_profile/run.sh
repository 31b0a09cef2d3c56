#!/bin/bash
# Function profile of the DEVICE ALGORITHM on the host (TEST TOOL): the emulation (tests/emu/ksolve_emu.cpp — the kernels' own
# source with a loop-over-lanes Wave) compiled with -finstrument-functions, a tiny enter/exit hook that keeps calls / self / inclusive
# cycles per function, one Solve() of a fixture. Host cycles: a wave-wide operation is a loop over 64 lanes here, so it weighs more than
# on the device — what the table is good for is WHICH functions a pod goes through and how often (DESIGN.md §8.2 quotes it for the
# configs[2] shape: a third of the general engine's work per pod in filter_core, a sixth in finish_record, a sixth in the scan).
#   usage: bash tests/tools/emu_profile/run.sh [pods=60000] [fixture=config3]     -> /tmp/emu_profile/prof_out.txt (sorted by self time)
set -e
cd "$(dirname "$0")/../../.."
P=${1:-60000}; F=${2:-config3}
O=/tmp/emu_profile; mkdir -p $O
g++ -O2 -std=c++17 -fPIC -c -o $O/prof_hooks.o tests/tools/emu_profile/prof_hooks.cpp
# lambdas (operator()), the Wave helpers and the smallest accessors are left out: instrumenting them measures the hook
g++ -O2 -std=c++17 -fPIC -c -pthread -finstrument-functions \
  "-finstrument-functions-exclude-file-list=wave.h,/usr/include,/usr/lib,json_mini,reqalg.h,ksp.h" \
  "-finstrument-functions-exclude-function-list=operator,lds_get,lds_put,fast_uniform,ctz64,popc64,RA,RAV,lo32,hi32,claim_at,topo_has,mix64,mask_of,head_,c_f0,c_f1,at" \
  -o $O/emu_prof.o tests/emu/ksolve_emu.cpp
g++ -shared -pthread -Wl,-Bsymbolic -o $O/libksolve_emu_prof.so $O/emu_prof.o $O/prof_hooks.o -ldl     # -Bsymbolic: glibc has no-op hooks of the same names
KSOLVE_TEST_SOLVER_LIB=1 python - "$P" "$F" <<'PY'
import ctypes, sys
sys.path.insert(0, ".")
from karpenter_amd import fixtures as fx
from karpenter_amd.scheduling import NewScheduler
pods, fixture = int(sys.argv[1]), sys.argv[2]
prob = getattr(fx, fixture)(pods=pods)
lib = ctypes.CDLL("/tmp/emu_profile/libksolve_emu_prof.so")
s = NewScheduler(prob, solver_lib="/tmp/emu_profile/libksolve_emu_prof.so")
lib.prof_start(); r = s.Solve(want_results=False); lib.prof_stop(b"/tmp/emu_profile/prof_out.txt")
print(r["counters"]["engine"], r["counters"]["pods"], "pods ->", r["counters"]["claims"], "NodeClaims; profile: /tmp/emu_profile/prof_out.txt")
s.close()
PY
head -25 $O/prof_out.txt
