#!/bin/bash
# Instrumented copy of the library for tools/s3time.py: only the landing decoder's translation unit is rebuilt
# (-DFDH_S3_DEBUG: per-stream phase clocks), the other objects are the product's.
set -e
cd "$(dirname "$0")/../fdeflate_amd/csrc"
make
mkdir -p build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -DFDH_S3_DEBUG -DFDH_S2_DEBUG -c -o build/inflate_seg3.dbg.o inflate_seg3.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../libfdeflate_hip_debug.so $(ls build/*.hip.o build/*.cpp.o | grep -v inflate_seg3.hip.o) build/inflate_seg3.dbg.o -ldl
ls -la ../libfdeflate_hip_debug.so
