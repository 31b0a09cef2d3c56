# Register / spill / scratch / LDS usage of every pack kernel in a built libksolve.so, from the code object notes. The library is linked
# from several translation units (csrc/*.hip), so its .hip_fatbin section holds one offload bundle per unit: each is unbundled on its own.
# usage: bash scripts/kernel_resources.sh karpenter_amd/libksolve.so
so=$(readlink -f $1)
tmp=$(mktemp -d)
cd $tmp
/opt/rocm/lib/llvm/bin/llvm-objcopy --dump-section .hip_fatbin=$tmp/fat.bin $so
python3 - <<'PY'
data = open("fat.bin", "rb").read()
magic = b"__CLANG_OFFLOAD_BUNDLE__"
starts = []
i = data.find(magic)
while i >= 0:
    starts.append(i)
    i = data.find(magic, i + 1)
for n, a in enumerate(starts):
    b = starts[n + 1] if n + 1 < len(starts) else len(data)
    open(f"bundle{n}.bin", "wb").write(data[a:b])
PY
for b in bundle*.bin; do
  /opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$b --output=$b.co --unbundle 2>/dev/null
  /opt/rocm/lib/llvm/bin/llvm-readelf --notes $b.co | grep -E "\.name:|sgpr_count|sgpr_spill|vgpr_count|vgpr_spill|private_segment_fixed|group_segment_fixed" | paste - - - - - - - | grep pack | sed 's/  */ /g'
done
