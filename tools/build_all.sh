#!/bin/bash
# Builds the product library, the instrumented (-DFDH_DEBUG_TILES) copy used by tools/segdiag.py,
# and prints the register / LDS / scratch use of the kernel named in $1 (default: segments).
set -e
cd "$(dirname "$0")/../fdeflate_amd/csrc"
make   # set -e: a failing make ends the script with its status
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -DFDH_DEBUG_TILES -DFDH_DEBUG_GEN -shared \
    -o ../libfdeflate_hip_debug.so fdeflate_hip.cpp stream_decompressor.cpp multi_gpu.cpp inflate.hip deflate_ultrafast.hip deflate_stored.hip deflate_general.hip png_filter.hip -ldl 2>&1 | grep -E "error" && exit 1
K=${1:-inflate_segments_kernel}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c inflate.hip -o /tmp/inflate.o -Rpass-analysis=kernel-resource-usage 2>&1 \
    | grep -A10 "Function Name: .*$K" | grep -E "VGPRs:|Scratch|Spill|Occupancy|LDS" || true
ls -la ../*.so
