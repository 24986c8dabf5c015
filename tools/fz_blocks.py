"""Developer aid: per-basic-block instruction mix of one kernel in /tmp/fz_regs.s (written by tools/fz_regs.sh).
    python tools/fz_blocks.py ILb0ELb1ELb1E [dump-block-label]"""
import re
import sys
txt = open('/tmp/fz_regs.s').read()
want = sys.argv[1] if len(sys.argv) > 1 else 'ILb0ELb1ELb1E'
i = txt.index('_Z12k_pair_fused' + want + 'Ev9FusedArgs:')
j = txt.index('.Lfunc_end', i)
blocks, cur, name = [], [], 'entry'
for l in txt[i:j].split('\n'):
    s = l.strip()
    m = re.match(r'^(\.LBB\d+_\d+):', s)
    if m:
        blocks.append((name, cur)); cur = []; name = m.group(1) + (' LOOP' if 'Loop Header' in s else '')
    elif s and not s.startswith(';'):
        cur.append(s)
blocks.append((name, cur))
cnt = lambda b, p: sum(1 for x in b if x.startswith(p))
for n, b in blocks:
    m = cnt(b, 'v_mfma')
    if m or len(b) > 40:
        print(f'{n:18s} n={len(b):5d} mfma={m:3d} s_nop={cnt(b, "s_nop"):3d} valu={cnt(b, "v_") - m:4d} (mov={cnt(b, "v_mov")}, accvgpr={cnt(b, "v_accvgpr")}) ds={cnt(b, "ds_"):3d} '
              f'vmem={cnt(b, "buffer_"):3d} waitcnt={cnt(b, "s_waitcnt"):3d} barrier={cnt(b, "s_barrier")} scratch={cnt(b, "scratch_")}')
if len(sys.argv) > 2:
    for n, b in blocks:
        if n.split()[0] == sys.argv[2]:
            print('\n'.join(b))
