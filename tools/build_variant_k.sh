#!/bin/bash
# tools/build_variant_k.sh <tag> <extra cflags...>: a diagnostic build of kernels.hip linked with the current objects
# -> vectorlite_amd/libvl_<tag>.so (use with VL_LIB_PATH)
set -e
TAG=$1; shift
cd "$(dirname "$0")/.."
O=vectorlite_amd/csrc/_obj
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function -Iinclude "$@" -c vectorlite_amd/csrc/kernels.hip -o /tmp/kernels_$TAG.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o vectorlite_amd/libvl_$TAG.so /tmp/kernels_$TAG.o $O/hnsw.o $O/mfma_scan.o $O/shard.o $O/flat_index.o $O/hnsw_index.o $O/shard_comm.o $O/vlc_loader.o $O/c_api.o -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
echo built vectorlite_amd/libvl_$TAG.so
