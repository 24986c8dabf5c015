"""Build <dir>/traffic.json (HBM bytes per launch and kernel) from the FETCH_SIZE / WRITE_SIZE passes of tools/profile_round.sh,
and <dir>/kernel_stats.csv (name, calls, average ns) from the --stats pass.  FETCH_SIZE is doubled (gfx950: the counter tallies
128-B requests at 64 B, MI355X_MICROARCH.md section HBM); both counters are in KB."""
import csv
import glob
import hashlib
import json
import os
import sys
from collections import defaultdict

out = sys.argv[1]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(n):
    return n.split('(')[0].replace('void ', '').strip()


def counters(sub, name):
    acc = defaultdict(list)
    for path in glob.glob(os.path.join(out, sub, '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(path)):
            if r['Counter_Name'] == name:
                acc[short(r['Kernel_Name'])].append(float(r['Counter_Value']))
    return {k: sum(v) / len(v) for k, v in acc.items()}


def durations(sub):
    acc = defaultdict(list)
    for path in glob.glob(os.path.join(out, sub, '**', '*kernel_trace.csv'), recursive=True):
        for r in csv.DictReader(open(path)):
            acc[short(r['Kernel_Name'])].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
    return acc


fetch, write = counters('pass_fetch', 'FETCH_SIZE'), counters('pass_write', 'WRITE_SIZE')
dur = durations('stats')
sys.path.insert(0, ROOT)
from bench import kernels_sha  # noqa: E402  (one definition of "the sources these numbers belong to")
blob = {'source': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over bench.py --steps 4 --warmup 2; FETCH_SIZE doubled (gfx950)',
        'kernels_sha': kernels_sha(), 'hx': {}}
for k in sorted(set(fetch) | set(write)):
    if not k.startswith('k_'):
        continue
    blob['hx'][k] = {'hbm_read_bytes': 2 * fetch.get(k, 0.0) * 1024, 'hbm_write_bytes': write.get(k, 0.0) * 1024,
                     'avg_us_unprofiled_pass': (sum(dur[k]) / len(dur[k]) / 1e3) if k in dur else None, 'launches': len(dur.get(k, []))}
json.dump(blob, open(os.path.join(out, 'traffic.json'), 'w'), indent=1)
with open(os.path.join(out, 'kernel_stats.csv'), 'w') as fh:
    fh.write('kernel,calls,avg_ns,total_ns\n')
    for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
        fh.write('"%s",%d,%.1f,%d\n' % (k, len(v), sum(v) / len(v), sum(v)))
tot = sum(sum(v) for v in dur.values())
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1]))[:14]:
    t = blob['hx'].get(k, {})
    rd, wr = t.get('hbm_read_bytes', 0) / 1e6, t.get('hbm_write_bytes', 0) / 1e6
    print(f'{k:34s} calls {len(v):5d} avg {sum(v) / len(v) / 1e3:9.1f} us  share {sum(v) / tot:5.1%}  HBM read {rd:8.1f} MB write {wr:8.1f} MB')
