#!/bin/bash
# Builds variants of the library with other LZ-window kernel parameters (for A/B runs on the GPU box):
#   tools/lzsweep.sh "LIT DIST RANGE IMG RING WAVES tag" ...   ->  fdeflate_amd/libfdeflate_hip_lz_<tag>.so
cd "$(dirname "$0")/../fdeflate_amd/csrc" || exit 1
for cfg in "$@"; do
  set -- $cfg
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -DFDH_LZ_LIT_BITS=$1 -DFDH_LZ_DIST_BITS=$2 \
    -DFDH_LZ_RANGE=$3 -DFDH_LZ_WARM=$3 -DFDH_LZ_IMG=$4 -DFDH_LZ_RING=$5 -DFDH_LZ_WAVES_PER_CU=$6 -shared -o ../libfdeflate_hip_lz_$7.so \
    fdeflate_hip.cpp stream_decompressor.cpp multi_gpu.cpp inflate.hip deflate_ultrafast.hip deflate_stored.hip deflate_general.hip png_filter.hip -ldl 2>&1 \
    | grep -E "error|static assertion"
done
ls ../*.so
