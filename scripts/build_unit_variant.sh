# A measurement variant of the device library that differs from the product in ONE translation unit (csrc/<unit>.hip compiled with the
# given -D flags, the other units' product objects linked as they are: run `python __graft_entry__.py` first).
# -> karpenter_amd/variants/libksolve_<tag>.so; never loaded by the product.
# usage: bash scripts/build_unit_variant.sh ksolve_pack_sweep4 sweep4_nospill -DKSOLVE_SWEEP4_NO_SPILL
cd "$(dirname "$0")/.." && mkdir -p karpenter_amd/variants
unit=$1; tag=$2; shift; shift
obj=karpenter_amd/csrc/_obj/product
objs=""
for u in ksolve ksolve_pack_general ksolve_pack_batch ksolve_pack_sweep4 ksolve_pack_fast ksolve_pack_topo; do
  if [ $u = $unit ]; then objs="$objs /tmp/${unit}_$tag.o"; else objs="$objs $obj/$u.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -c karpenter_amd/csrc/$unit.hip -o /tmp/${unit}_$tag.o && \
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o karpenter_amd/variants/libksolve_$tag.so $objs
