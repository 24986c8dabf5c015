#!/bin/bash
# GPU box: rocprofv3 kernel trace of a few reverse-loop steps; per (kernel, grid) average durations -> gpurun_out/<tag>_trace_by_grid.txt
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/trace_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/raw" -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extra-legs --profile-steps 1 > "$OUT/log.txt" 2>&1 || { tail -5 "$OUT/log.txt"; exit 1; }
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
f = glob.glob(out + '/raw/**/*kernel_trace.csv', recursive=True)[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    name = r['Kernel_Name'].split('(')[0][:48]
    grid = (r.get('Grid_Size_X') or r.get('Grid_Size'), r.get('Grid_Size_Y'), r.get('Grid_Size_Z'), r.get('Workgroup_Size_X') or r.get('Workgroup_Size'))
    agg[(name, grid, r.get('LDS_Block_Size'), r.get('VGPR_Count'))].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
with open(out + '_by_grid.txt', 'w') as fh:
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        fh.write(f'{k[0]:48s} grid {str(k[1]):34s} lds {k[2]:>7s} vgpr {k[3]:>4s} calls {len(v):4d} avg {sum(v)/len(v):8.1f} us min {min(v):8.1f} total {sum(v)/1e3:8.2f} ms\n')
print(open(out + '_by_grid.txt').read()[:6000])
PY
rm -rf "$OUT/raw"
