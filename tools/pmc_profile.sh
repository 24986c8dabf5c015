#!/bin/bash
# GPU box: per-kernel PMC counters for the bench workload, one rocprofv3 pass per counter group
# (counters only with --kernel-trace; see the gpurun rules).  Usage: tools/pmc_profile.sh <outdir> [bench args]
set -u
OUT=${1:-gpurun_out/pmc}; shift || true
ARGS=${@:---steps 2 --warmup 1 --no-cpu-baseline --profile-steps 1}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
i=0
while read -r GROUP; do
  [ -z "$GROUP" ] && continue
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $GROUP --output-format csv -d "$OUT/pass$i" -- python3 bench.py $ARGS > "$OUT/pass$i.log" 2>&1 || echo "pass $i failed"
done <<'GROUPS'
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA
SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAVES SQ_ACTIVE_INST_VALU
FETCH_SIZE GRBM_GUI_ACTIVE
WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
GROUPS
python3 tools/pmc_summary.py "$OUT" > "$OUT/summary.txt" 2>&1
tail -60 "$OUT/summary.txt"
