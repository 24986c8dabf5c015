#!/bin/bash
# GPU box: per-kernel time of one training step (rocprofv3 --kernel-trace --stats).  usage: tools/train_profile.sh <outdir> [N] [B] [fast_math]
OUT=${1:-gpurun_out/trainprof}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/raw" -- python3 tools/train_bench.py ${2:-128} ${3:-2} ${4:-0} > "$OUT/run.log" 2>&1 || exit 1
cp "$OUT"/raw/*/*_kernel_stats.csv "$OUT/kernel_stats.csv"
rm -rf "$OUT/raw"
head -40 "$OUT/kernel_stats.csv"
