#!/bin/bash
# Register / scratch / LDS use of the kernels in a shipped object, from the code object's own notes:
#   tools/kres.sh fdeflate_amd/csrc/build/inflate_seg3.hip.o
# (the .hip_fatbin section holds a clang offload bundle; the gfx950 code object is unbundled from it and read with llvm-readelf --notes)
set -e
O=$1
T=$(mktemp -d)
/opt/rocm/llvm/bin/llvm-objcopy -O binary --only-section=.hip_fatbin "$O" $T/fat.bin
TGT=$(/opt/rocm/llvm/bin/clang-offload-bundler --list --type=o --input=$T/fat.bin | grep gfx950 | head -1)
/opt/rocm/llvm/bin/clang-offload-bundler --type=o --targets=$TGT --input=$T/fat.bin --output=$T/dev.o --unbundle
/opt/rocm/llvm/bin/llvm-readelf --notes $T/dev.o | grep -E "\.name:|\.vgpr_count|\.vgpr_spill_count|\.sgpr_spill_count|\.private_segment_fixed_size|\.group_segment_fixed_size" \
  | sed 's/^ *//' | awk '/^\.group_segment_fixed_size/{g=$0; next} /^\.name:/{if (line) print line; line=$0"  "g; next} {line=line"  "$0} END{print line}' | grep -v "^\.name: *[a-z_]*$" | sort
# (the notes list a kernel's keys in alphabetical order: .group_segment_fixed_size comes BEFORE .name and belongs to the name behind it)
rm -rf $T
