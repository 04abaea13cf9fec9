#!/bin/bash
# Builds the product library and the copy with the LZ-window kernel's per-phase clocks (tools/lztime.py).
set -e
cd "$(dirname "$0")/../fdeflate_amd/csrc"
make
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -DFDH_LZ_DEBUG -shared \
    -o ../libfdeflate_hip_lzdebug.so fdeflate_hip.cpp stream_decompressor.cpp multi_gpu.cpp inflate.hip inflate_seg3.hip deflate_ultrafast.hip deflate_stored.hip deflate_general.hip png_filter.hip -ldl 2>&1 | grep -E "error" && exit 1
ls -la ../*.so
