"""Summarise rocprofv3 --pmc passes: per kernel name, mean of each counter per dispatch."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for path in glob.glob(os.path.join(out, 'pass*', '**', '*counter_collection.csv'), recursive=True):
    for r in csv.DictReader(open(path)):
        name = r['Kernel_Name'].split('(')[0].replace('void ', '')
        acc[name][r['Counter_Name']].append(float(r['Counter_Value']))
dur = defaultdict(list)
for path in glob.glob(os.path.join(out, 'pass1', '**', '*kernel_trace.csv'), recursive=True):
    for r in csv.DictReader(open(path)):
        name = r['Kernel_Name'].split('(')[0].replace('void ', '')
        dur[name].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
names = sorted(acc, key=lambda n: -sum(dur.get(n, [0])))
for n in names:
    if not n.startswith('k_'):
        continue
    c = {k: sum(v) / len(v) for k, v in acc[n].items()}
    d = sum(dur[n]) / max(len(dur[n]), 1)
    print(f'== {n}  dispatches={len(dur[n])} avg_us={d:.1f}')
    for k in sorted(c):
        print(f'   {k:28s} {c[k]:16.1f}')
    if 'FETCH_SIZE' in c:
        print(f'   HBM read  (2x FETCH_SIZE KB, gfx950 correction) {2 * c["FETCH_SIZE"] / 1024:10.1f} MB')
    if 'WRITE_SIZE' in c:
        print(f'   HBM write (WRITE_SIZE KB)                       {c["WRITE_SIZE"] / 1024:10.1f} MB')
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in c and 'GRBM_GUI_ACTIVE' in c:
        pass
