#!/bin/bash
# Developer tool: registers / LDS / spills of every kernel in a built object or library (code-object metadata).
#   tools/kernel_resources.sh obia_amd/csrc/slic_sweep.o [name-filter]
f=$(realpath "$1"); filt=${2:-.}
LLVM=${ROCM_PATH:-/opt/rocm}/lib/llvm/bin
tmp=$(mktemp -d); cp "$f" $tmp/in.o; (cd $tmp && $LLVM/llvm-objdump --offloading in.o >/dev/null 2>&1)
for co in $tmp/*amdgcn*; do
  $LLVM/llvm-readelf --notes "$co" | awk '
    /\.group_segment_fixed_size:/ {lds=$2} /\.name:/ {name=$2} /\.sgpr_count:/ {sg=$2} /\.sgpr_spill_count:/ {ss=$2}
    /\.vgpr_count:/ {vg=$2} /\.vgpr_spill_count:/ {vs=$2} /\.private_segment_fixed_size:/ {pv=$2}
    /\.wavefront_size:/ {printf "%s vgpr %3s sgpr %3s lds %6s scratch %4s spill v%s s%s\n", name, vg, sg, lds, pv, vs, ss}'
done | while read name rest; do printf "%-70s %s\n" "$(echo $name | c++filt | sed 's/(.*//; s/^void //; s/obia:://')" "$rest"; done | grep -E "$filt"
rm -rf $tmp
