#!/bin/bash
# A variant of the library for A/B runs: one translation unit rebuilt with extra flags, the others from build/.
#   tools/variant.sh <name> <file.hip> [flags...]   ->  ab/lib_<name>.so
set -e
name=$1; src=$2; shift; shift
cd "$(dirname "$0")/../fdeflate_amd/csrc"
make -s
mkdir -p ../../ab build_var
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function "$@" -c -o build_var/$name.o $src
objs=$(ls build/*.hip.o build/*.cpp.o | grep -v "build/$src.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../../ab/lib_$name.so $objs build_var/$name.o -ldl
ls -la ../../ab/lib_$name.so
