# A measurement variant of the device library that differs from the product in the spread engine's translation unit only
# (csrc/ksolve_pack_topo.hip compiled with the given -D flags, the other units' product objects linked as they are: run
# `python __graft_entry__.py` first). -> karpenter_amd/variants/libksolve_<tag>.so; never loaded by the product.
# usage: bash scripts/build_topo_variant.sh timers -DKSOLVE_PHASE_TIMERS
cd "$(dirname "$0")/.." && mkdir -p karpenter_amd/variants
tag=$1; shift
obj=karpenter_amd/csrc/_obj/product
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -c karpenter_amd/csrc/ksolve_pack_topo.hip -o /tmp/ksolve_pack_topo_$tag.o && \
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o karpenter_amd/variants/libksolve_$tag.so $obj/ksolve.o $obj/ksolve_pack_general.o $obj/ksolve_pack_batch.o $obj/ksolve_pack_sweep4.o $obj/ksolve_pack_fast.o /tmp/ksolve_pack_topo_$tag.o
