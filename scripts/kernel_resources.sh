# Register / spill / scratch / LDS usage of every kernel in a built libksolve.so, from the code object notes.
# usage: bash scripts/kernel_resources.sh karpenter_amd/libksolve.so
so=$(readlink -f $1)
tmp=$(mktemp -d)
cd $tmp
/opt/rocm/lib/llvm/bin/llvm-objcopy --dump-section .hip_fatbin=$tmp/fat.bin $so
/opt/rocm/lib/llvm/bin/clang-offload-bundler --list --type=o --input=$tmp/fat.bin
/opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$tmp/fat.bin --output=$tmp/dev.co --unbundle
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $tmp/dev.co | grep -E "\.name:|sgpr_count|sgpr_spill|vgpr_count|vgpr_spill|private_segment_fixed|group_segment_fixed" | paste - - - - - - - | grep pack | sed 's/  */ /g'
