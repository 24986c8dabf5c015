#!/bin/bash
# GPU box: the rocprofv3 evidence of a round.  usage: tools/profile_round.sh <tag>   (e.g. r02)
#   1. --kernel-trace --stats of the bench command            -> gpurun_out/prof_<tag>/stats  (per-kernel average durations)
#   2. --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes -> gpurun_out/prof_<tag>/traffic.json (HBM bytes per launch, FETCH doubled
#      as MI355X_MICROARCH.md prescribes for gfx950)
TAG=${1:-r02}
OUT=gpurun_out/prof_$TAG
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
CMD="python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extra-legs --profile-steps 1"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- $CMD > "$OUT/stats.log" 2>&1 || exit 1
echo "stats done"
timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pass_fetch" -- $CMD > "$OUT/fetch.log" 2>&1 || exit 1
echo "fetch done"
timeout -k 10 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pass_write" -- $CMD > "$OUT/write.log" 2>&1 || exit 1
echo "write done"
python3 tools/pmc_traffic.py "$OUT" > "$OUT/traffic_summary.txt" 2>&1
cp "$OUT"/stats/*/*_kernel_stats.csv "$OUT/rocprofv3_kernel_stats.csv" 2>/dev/null
rm -rf "$OUT/stats" "$OUT/pass_fetch" "$OUT/pass_write"      # raw traces are tens of MB: only the summaries travel back
cat "$OUT/traffic_summary.txt" | head -30
