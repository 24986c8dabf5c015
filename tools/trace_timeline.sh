#!/bin/bash
# GPU box: rocprofv3 kernel trace of a few reverse-loop steps; start offset / duration / queue of every kernel of the last structure net
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/trace_tl; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extra-legs --profile-steps 1 > $OUT/log.txt 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/trace_tl/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
ib = [i for i, r in enumerate(rows) if 'k_ipa_bias' in r['Kernel_Name']]
i0 = ib[-3]
t0 = int(rows[i0]['Start_Timestamp'])
for r in rows[i0:i0 + 60]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print('%9.1f .. %9.1f us  dur %7.1f  q %s  %s' % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, r.get('Queue_Id', '?'), r['Kernel_Name'].split('(')[0][:40]))
PY
rm -rf $OUT
