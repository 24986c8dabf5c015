"""One line per kernel from tools/pmc_profile.sh's summary.txt (clock, occupancy, MFMA-busy share, per-wave cycle and instruction counts,
LDS bank-conflict share).  python3 tools/pmc_digest.py gpurun_out/pmc_r03/summary.txt > profiles/r03_pmc_digest_hx.txt"""
import re
import sys

txt = open(sys.argv[1]).read()
for blk in txt.split('== ')[1:]:
    name = blk.split()[0]
    if not any(k in name for k in ('trimul', 'transition', 'ipa_attn', 'gemm_rows', 'ipa_bias', 'pair_init', 'pair_fused', 'ipa_prep', 'struct_rows')):
        continue
    c = {m.group(1): float(m.group(2)) for m in re.finditer(r'^\s+(\S+)\s+([\d.]+)\s*$', blk, re.M)}
    us = float(re.search(r'avg_us=([\d.]+)', blk).group(1))
    cyc = c.get('GRBM_GUI_ACTIVE', 0) / 8
    waves = c.get('SQ_WAVES', 1)
    wc = c.get('SQ_WAVE_CYCLES', 0) * 4
    print(f"{name:28s} {us:8.1f}us clk={cyc/us/1e3:.2f}GHz occ={wc/(1024*cyc):.2f}w/simd mfma_busy={c.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/(1024*cyc):.2f} "
          f"per-wave: life={wc/waves/1e3:.1f}K wait_any={c.get('SQ_WAIT_ANY',0)*4/waves/1e3:.1f}K wait_inst={c.get('SQ_WAIT_INST_ANY',0)*4/waves/1e3:.1f}K "
          f"active={c.get('SQ_ACTIVE_INST_ANY',0)*4/waves/1e3:.1f}K valu={c.get('SQ_INSTS_VALU',0)/waves:.0f} mfma={c.get('SQ_INSTS_MFMA',0)/waves:.0f} "
          f"lds={c.get('SQ_INSTS_LDS',0)/waves:.0f} ldsconf={c.get('SQ_LDS_BANK_CONFLICT',0)/max(c.get('SQ_LDS_IDX_ACTIVE',1),1):.2f}")
