#!/bin/bash
# Developer aid: build ablation variants of the hx kernels (-DHX_ABL=n) as genie2_amd/lib/abl/libgenie_abl<n>.so
# (selected at run time with GENIE_HIP_LIB).  Usage: tools/abl_build.sh 1 2 4
set -e
cd "$(dirname "$0")/.."
python -c "import __graft_entry__ as g; g.build()" > /dev/null
mkdir -p genie2_amd/lib/abl
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -DGENIE_BUILD -DHX_ABL=$n -c genie2_amd/csrc/pair_hx_kernels.hip -o genie2_amd/lib/abl/hx_$n.o 2>/dev/null &
done
wait
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o genie2_amd/lib/abl/libgenie_abl$n.so genie2_amd/lib/abl/hx_$n.o genie2_amd/lib/pair_kernels.o genie2_amd/lib/pair_wl_kernels.o genie2_amd/lib/single_kernels.o genie2_amd/lib/genie_api.o
  rm genie2_amd/lib/abl/hx_$n.o
done
ls -la genie2_amd/lib/abl
