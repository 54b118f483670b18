#!/usr/bin/env python3
"""Timeline of the last complete step in a rocprofv3 --kernel-trace CSV (bench.py --inflight 1): start / end / duration of every
kernel relative to the step's first kernel, its queue (= engine lane), and the step's span against the union of busy time.
    python tools/micro/step_timeline.py <rocprof dir> [first-kernel-substring]"""
import csv, glob, re, sys
d = sys.argv[1]
first = sys.argv[2] if len(sys.argv) > 2 else None
rows = []
for f in glob.glob(d + '/**/*kernel_trace.csv', recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
if first is None:
    first = 'input_s2d' if any('input_s2d' in r['Kernel_Name'] for r in rows[-400:]) else 'stem_planar'
idx = [i for i, r in enumerate(rows) if first in r['Kernel_Name']]
a, b = idx[-3], idx[-2]
want = sys.argv[3] if len(sys.argv) > 3 else None      # pick the last step that contains a kernel with this substring
if want:
    for k in range(len(idx) - 2, 0, -1):
        if any(want in r['Kernel_Name'] for r in rows[idx[k - 1]:idx[k]]):
            a, b = idx[k - 1], idx[k]
            break
st = rows[a:b]
t0 = int(st[0]['Start_Timestamp'])
iv = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in st)
cs, ce, un = iv[0][0], iv[0][1], 0
for s, e in iv[1:]:
    if s > ce:
        un += ce - cs; cs, ce = s, e
    else:
        ce = max(ce, e)
un += ce - cs
print('kernels %d  span %.1f us  busy (union) %.1f us  sum of durations %.1f us' % (len(st), (max(e for s, e in iv) - t0) / 1e3, un / 1e3, sum(e - s for s, e in iv) / 1e3))
for r in st:
    s, e = int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0
    n = re.sub(r'^_ZN2lp\d+', '', r['Kernel_Name'])
    n = re.sub(r'EvNS_8ConvArgs.*', '', n)[:52]
    print('%8.1f %8.1f %6.1f  q%-2s %s' % (s / 1e3, e / 1e3, (e - s) / 1e3, r['Queue_Id'], n))
