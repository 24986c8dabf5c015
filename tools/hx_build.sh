#!/bin/bash
# Developer aid: build a variant of pair_hx_kernels.hip as genie2_amd/lib/abl/libgenie_hx_<tag>.so (select with GENIE_HIP_LIB).
# usage: tools/hx_build.sh <tag> [-DHX_ABL=512 ...]
set -e
cd "$(dirname "$0")/.."
TAG=$1; shift
mkdir -p genie2_amd/lib/abl
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -DGENIE_BUILD "$@" -c genie2_amd/csrc/pair_hx_kernels.hip -o genie2_amd/lib/abl/hx_$TAG.o
OBJS=""
for f in pair_kernels pair_wl_kernels pair_fused_kernels single_kernels train_kernels train_layout_kernels probe_kernels genie_train genie_api; do OBJS="$OBJS genie2_amd/lib/$f.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o genie2_amd/lib/abl/libgenie_hx_$TAG.so genie2_amd/lib/abl/hx_$TAG.o $OBJS
rm genie2_amd/lib/abl/hx_$TAG.o
echo genie2_amd/lib/abl/libgenie_hx_$TAG.so
