#!/bin/bash
# tools/kernel_resources.sh <object.o> -- VGPR / SGPR / spill / LDS / scratch of every kernel in a hipcc object (gfx950 code object metadata)
obj=$(readlink -f $1)
tmp=$(mktemp -d)
L=/opt/rocm/lib/llvm/bin
$L/llvm-objcopy -O binary --only-section=.hip_fatbin $obj $tmp/fat.bin
$L/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$tmp/fat.bin --output=$tmp/dev.co --unbundle
$L/llvm-readelf --notes $tmp/dev.co | awk '
/\.name:/ {name=$2}
/\.vgpr_count:/ {v=$2} /\.sgpr_count:/ {s=$2} /\.vgpr_spill_count:/ {vs=$2} /\.sgpr_spill_count:/ {ss=$2}
/\.group_segment_fixed_size:/ {l=$2} /\.private_segment_fixed_size:/ {p=$2}
/\.wavefront_size:/ {printf "%-90s vgpr %3d sgpr %3d vspill %3d sspill %3d lds %6d scratch %4d\n", name, v, s, vs, ss, l, p}'
if [ -n "$2" ]; then cp $tmp/dev.co $2; fi
rm -rf $tmp
