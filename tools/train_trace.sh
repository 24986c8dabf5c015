#!/bin/bash
# GPU box: kernel trace of one training step, grouped by (kernel, grid): which GEMM shapes the time goes to.
# usage: tools/train_trace.sh <outdir> [N] [B] [fast_math]
OUT=${1:-gpurun_out/traintrace}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf "$OUT"; mkdir -p "$OUT"
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d "$OUT/raw" -- python3 tools/train_bench.py ${2:-256} ${3:-2} ${4:-0} > "$OUT/run.log" 2>&1 || exit 1
python3 - "$OUT" <<'PY'
import csv, glob, collections, sys, re
out = sys.argv[1]
f = glob.glob(out + '/raw/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
# the last step only: from the last k_pair_features launch on
st = [i for i, r in enumerate(rows) if 'k_pair_features(' in r['Kernel_Name'] and 'bwd' not in r['Kernel_Name']]
rows = rows[st[-1]:]
d = collections.defaultdict(list)
for r in rows:
    n = r['Kernel_Name']
    m = re.search(r'lambda\(long long\)#(\d+)', n)
    short = n.split('(')[0][:44] if 'k_ew' not in n else 'k_ew#' + '/'.join(re.findall(r'#(\d+)', n)[-2:])
    key = (short, r.get('Grid_Size_X', '?'), r.get('Grid_Size_Y', ''), r.get('Grid_Size_Z', ''))
    d[key].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
tot = sum(sum(v) for v in d.values())
with open(out + '/by_grid.txt', 'w') as fo:
    for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
        line = '%-46s grid %-22s calls %4d avg %8.1f us total %7.2f ms' % (k[0], ','.join(k[1:]), len(v), sum(v) / len(v), sum(v) / 1e3)
        fo.write(line + '\n')
    fo.write('kernel sum %.2f ms over %d launches\n' % (tot / 1e3, sum(len(v) for v in d.values())))
PY
rm -rf "$OUT/raw"
head -50 "$OUT/by_grid.txt"
