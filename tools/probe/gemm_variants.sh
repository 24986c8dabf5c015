#!/bin/bash
# Developer probe: builds train_kernels.hip with each GM_EXP knock-out into its own small library next to gemm_bench and links a copy
# of the probe against it.  Run here (needs hipcc, no GPU): tools/probe/gemm_variants.sh ; then on the GPU box run tools/probe/gv/run.sh
cd "$(dirname "$0")/../.." || exit 1
mkdir -p tools/probe/gv
for e in ${GM_VARIANTS:-0 1 2 4 8 3 7}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DGENIE_BUILD -DGM_EXP=$e -Wno-unused-function -Wno-unused-value \
      -I genie2_amd/csrc genie2_amd/csrc/train_kernels.hip -o tools/probe/gv/libgm_$e.so 2>&1 | grep -v "warning" | grep -v "^$"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -I genie2_amd/csrc tools/probe/gemm_bench.hip -L tools/probe/gv -lgm_$e -Wl,-rpath,'$ORIGIN' \
      -o tools/probe/gv/bench_$e 2>&1 | grep -v "warning" | grep -v "^$"
done
cat > tools/probe/gv/run.sh <<'EOS'
#!/bin/bash
cd "$(dirname "$0")"
for b in bench_*; do echo "== $b (GM_EXP bits: 1 no MFMA, 2 no global loads, 4 no result store, 8 hi piece only)"; timeout -k 10 60 ./$b ${1:-3} || exit 1; done
EOS
chmod +x tools/probe/gv/run.sh
ls tools/probe/gv
