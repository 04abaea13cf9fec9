#!/bin/bash
# Builds the product library (make: one object per source file), the instrumented copy used by tools/segdiag.py
# (-DFDH_DEBUG_TILES -DFDH_DEBUG_GEN: every translation unit again, into build_dbg/), and prints the register /
# LDS / scratch use of the kernel named in $1 (default: the landing decoder) from its translation unit.
set -e
cd "$(dirname "$0")/../fdeflate_amd/csrc"
make   # set -e: a failing make ends the script with its status
SRCS="fdeflate_hip.cpp stream_decompressor.cpp multi_gpu.cpp inflate.hip inflate_seg3.hip deflate_ultrafast.hip deflate_stored.hip deflate_general.hip png_filter.hip"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -DFDH_DEBUG_TILES -DFDH_DEBUG_GEN -shared \
    -o ../libfdeflate_hip_debug.so $SRCS -ldl 2>&1 | grep -E "error" && exit 1
K=${1:-inflate_seg3_kernel}
F=inflate.hip
case "$K" in *seg3*) F=inflate_seg3.hip;; *deflate_ultrafast*) F=deflate_ultrafast.hip;; *deflate_parse*|*deflate_write*) F=deflate_general.hip;; *png*) F=png_filter.hip;; esac
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c $F -o /tmp/fdh_res.o -Rpass-analysis=kernel-resource-usage 2>&1 \
    | grep -A10 "Function Name: .*$K" | grep -E "VGPRs:|Scratch|Spill|Occupancy|LDS" || true
ls -la ../*.so
