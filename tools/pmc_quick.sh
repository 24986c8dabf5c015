#!/bin/bash
# one PMC pass (SQ group) over the bench workload; usage: tools/pmc_quick.sh <outdir>
OUT=${1:-gpurun_out/pmcq}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA --output-format csv -d "$OUT/pass1" -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs --profile-steps 1 > "$OUT/pass1.log" 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_INSTS_SALU --output-format csv -d "$OUT/pass2" -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs --profile-steps 1 > "$OUT/pass2.log" 2>&1
python3 tools/pmc_summary.py "$OUT" > "$OUT/summary.txt" 2>&1
rm -rf "$OUT/pass1" "$OUT/pass2"      # raw counter CSVs: tens of MB
python3 - "$OUT/summary.txt" <<'PY'
import re, sys
txt = open(sys.argv[1]).read()
for blk in txt.split('== ')[1:]:
    name = blk.split()[0]
    if not any(k in name for k in ('trimul', 'transition', 'ipa_attn', 'gemm_rows', 'ipa_bias', 'pair_init', 'pair_fused')):
        continue
    c = {m.group(1): float(m.group(2)) for m in re.finditer(r'^\s+(\S+)\s+([\d.]+)\s*$', blk, re.M)}
    us = float(re.search(r'avg_us=([\d.]+)', blk).group(1))
    cyc = c.get('GRBM_GUI_ACTIVE', 0) / 8
    waves = c.get('SQ_WAVES', 1)
    wc = c.get('SQ_WAVE_CYCLES', 0) * 4
    print(f"{name:28s} {us:8.1f}us clk={cyc/us/1e3:.2f}GHz occ={wc/(1024*cyc):.2f}w/simd mfma_busy={c.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/(1024*cyc):.2f} "
          f"per-wave: life={wc/waves/1e3:.1f}K wait_any={c.get('SQ_WAIT_ANY',0)*4/waves/1e3:.1f}K wait_inst={c.get('SQ_WAIT_INST_ANY',0)*4/waves/1e3:.1f}K "
          f"active={c.get('SQ_ACTIVE_INST_ANY',0)*4/waves/1e3:.1f}K valu={c.get('SQ_INSTS_VALU',0)/waves:.0f} mfma={c.get('SQ_INSTS_MFMA',0)/waves:.0f} lds={c.get('SQ_INSTS_LDS',0)/waves:.0f} ldsconf={c.get('SQ_LDS_BANK_CONFLICT',0)/max(c.get('SQ_LDS_IDX_ACTIVE',1),1):.2f}")
PY
