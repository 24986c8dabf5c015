#!/bin/bash
# GPU box: kernel trace of the bench; per-launch durations of the single-track kernels grouped by grid size, and the
# timeline (start offset, duration, gap to the previous kernel's end) of one structure layer
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/trace_gemm; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extra-legs --profile-steps 1 > $OUT/log.txt 2>&1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/trace_gemm/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
d = collections.defaultdict(list)
names = ('gemm_rows', 'layernorm', 'ipa_prep', 'bb_update', 'struct_rows', 'ipa_attn')
for r in rows:
    if any(n in r['Kernel_Name'] for n in names):
        key = (r['Kernel_Name'].split('(')[0][:40], r.get('Grid_Size_X', r.get('Grid_Size', '?')), r.get('Grid_Size_Y', ''))
        d[key].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in sorted(d.items()):
    v.sort()
    print(k, 'n', len(v), 'min %.1f med %.1f max %.1f us' % (v[0], v[len(v) // 2], v[-1]))
# timeline of the last full structure net in the trace: from the last k_ipa_bias on
ib = [i for i, r in enumerate(rows) if 'k_ipa_bias' in r['Kernel_Name']]
if ib:
    i0 = ib[-2] if len(ib) > 1 else ib[-1]
    t0 = int(rows[i0]['Start_Timestamp']); prev_end = t0
    for r in rows[i0:i0 + 40]:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        print('%9.1f us  dur %7.1f  gap %6.1f  %s grid %s' % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3,
              r['Kernel_Name'].split('(')[0][:36], r.get('Grid_Size_X', r.get('Grid_Size', '?'))))
        prev_end = e
PY
