#!/bin/bash
# Developer aid: build ablation variants of the hx kernels (-DHX_ABL=n) as genie2_amd/lib/abl/libgenie_abl<n>.so
# (selected at run time with GENIE_HIP_LIB).  Usage: tools/abl_build.sh 1 2 4
set -e
cd "$(dirname "$0")/.."
python -c "import __graft_entry__ as g; g.build()" > /dev/null
mkdir -p genie2_amd/lib/abl
# GH=1 tools/abl_build.sh n...  builds the row-GEMM ablations (-DGH_ABL=n in single_kernels.hip) instead
SRC=pair_hx_kernels; DEF=HX_ABL; KEEP="genie2_amd/lib/single_kernels.o"
if [ -n "$GH" ]; then SRC=single_kernels; DEF=GH_ABL; KEEP="genie2_amd/lib/pair_hx_kernels.o"; fi
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -DGENIE_BUILD -D$DEF=$n -c genie2_amd/csrc/$SRC.hip -o genie2_amd/lib/abl/hx_$n.o 2>/dev/null &
done
wait
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o genie2_amd/lib/abl/libgenie_abl$n.so genie2_amd/lib/abl/hx_$n.o genie2_amd/lib/pair_kernels.o genie2_amd/lib/pair_wl_kernels.o $KEEP genie2_amd/lib/genie_api.o
  rm genie2_amd/lib/abl/hx_$n.o
done
ls -la genie2_amd/lib/abl
