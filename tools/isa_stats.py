"""Per-basic-block instruction mix of the kernels in a hipcc .s file (VALU is what competes with
f32 MFMA for the FP32 lanes).   python tools/isa_stats.py file.s [substring-of-kernel-name]"""
import sys
from collections import Counter

path = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else ''
kern = None
blocks = {}
order = []
for line in open(path):
    s = line.strip()
    if s.startswith('_Z') and ':' in s:
        kern = s.split(':')[0]
        blocks[(kern, 'entry')] = Counter()
        order.append((kern, 'entry'))
        continue
    if kern is None or want not in kern:
        continue
    if s.startswith('.LBB') and ':' in s:
        key = (kern, s.split(':')[0])
        blocks[key] = Counter()
        order.append(key)
        cur = key
        continue
    if not s or s[0] in ';.' or not order or order[-1][0] != kern:
        continue
    op = s.split()[0]
    c = blocks[order[-1]]
    if op.startswith('v_mfma'):
        c['mfma'] += 1
    elif op.startswith('v_'):
        c['valu'] += 1
        c['valu:' + op] += 1
    elif op.startswith('ds_'):
        c['lds'] += 1
    elif op.split('_')[0] in ('buffer', 'global', 'scratch', 'flat'):
        c['vmem'] += 1
    elif op.startswith('s_'):
        c['salu'] += 1
for key in order:
    c = blocks[key]
    if c['mfma'] or c['valu'] > 15:
        top = ', '.join(f'{k[5:]}={v}' for k, v in c.most_common() if k.startswith('valu:'))[:160]
        print(f'{key[0][:40]} {key[1]:10s} mfma={c["mfma"]:4d} valu={c["valu"]:4d} lds={c["lds"]:3d} vmem={c["vmem"]:3d} salu={c["salu"]:3d} | {top}')
