#!/bin/bash
# GPU box: one rocprofv3 counter pass of the bench.  Usage: tools/pmc_one.sh <outdir> "<counters>"
OUT=$1; CNT=$2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc $CNT --output-format csv -d "$OUT/pass1" -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --profile-steps 1 > "$OUT/pass1.log" 2>&1 || echo "pass failed"
python3 tools/pmc_summary.py "$OUT" > "$OUT/summary.txt" 2>&1
