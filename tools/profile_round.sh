#!/bin/bash
# GPU box: the round's evidence in one call -- bench JSON (both arithmetics), rocprofv3 --kernel-trace --stats of the
# same command, and the PMC passes (tools/pmc_profile.sh).  Usage: tools/profile_round.sh <tag>   (outputs under gpurun_out/<tag>/)
TAG=${1:-round}
OUT=gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
timeout -k 10 400 python3 bench.py --steps 40 --warmup 5 > "$OUT/bench_hx.json" 2> "$OUT/bench_hx.err" || echo "bench hx failed"
timeout -k 10 400 python3 bench.py --steps 20 --warmup 5 --math f32 --no-cpu-baseline > "$OUT/bench_f32.json" 2> "$OUT/bench_f32.err" || echo "bench f32 failed"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > "$OUT/stats.log" 2>&1 || echo "rocprof stats failed"
find "$OUT/stats" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats.csv" \;
bash tools/pmc_profile.sh "$OUT/pmc" > /dev/null 2>&1
cp "$OUT/pmc/summary.txt" "$OUT/pmc_summary.txt" 2>/dev/null
head -20 "$OUT/kernel_stats.csv"
