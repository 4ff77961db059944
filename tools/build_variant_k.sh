#!/bin/bash
# tools/build_variant_k.sh <tag> <extra cflags...>: a diagnostic build of kernels.hip linked with the current objects
# -> vectorlite_amd/libvl_<tag>.so (use with VL_LIB_PATH)
set -e
TAG=$1; shift
cd "$(dirname "$0")/.."
O=vectorlite_amd/csrc/_obj
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function -Iinclude "$@" -c vectorlite_amd/csrc/kernels.hip -o /tmp/kernels_$TAG.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o vectorlite_amd/libvl_$TAG.so /tmp/kernels_$TAG.o $(ls $O/*.o | grep -v "/kernels.o$") -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
echo built vectorlite_amd/libvl_$TAG.so
