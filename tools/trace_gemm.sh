#!/bin/bash
# GPU box: kernel trace of the bench, per-launch durations of k_gemm_rows* grouped by grid size
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/trace_gemm; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --profile-steps 1 > $OUT/log.txt 2>&1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/trace_gemm/**/*kernel_trace.csv', recursive=True)[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if 'gemm_rows' in r['Kernel_Name'] or 'layernorm' in r['Kernel_Name'] or 'ipa_prep' in r['Kernel_Name'] or 'bb_update' in r['Kernel_Name']:
        key = (r['Kernel_Name'].split('(')[0], r.get('Grid_Size_X', r.get('Grid_Size', '?')), r.get('Grid_Size_Y', ''))
        d[key].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in sorted(d.items()):
    v.sort()
    print(k, 'n', len(v), 'min %.1f med %.1f max %.1f us' % (v[0], v[len(v) // 2], v[-1]))
PY
