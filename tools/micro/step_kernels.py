#!/usr/bin/env python3
"""Kernels of ONE steady-state step (in issue order) from a rocprofv3 --kernel-trace of bench.py --inflight 1 --single-lane 1:
mean duration of every dispatch position over the last `steps` steps.   python tools/micro/step_kernels.py <trace dir> <steps>"""
import csv, glob, re, sys
rows = []
for f in glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True):
    rows += [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], int(r['LDS_Block_Size']), int(r['VGPR_Count']),
              int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X'])), int(r['Workgroup_Size_X'])) for r in csv.DictReader(open(f))]
rows.sort()
steps = int(sys.argv[2])
g = [i for i, r in enumerate(rows) if 'greedy_kernel' in r[2]]
# bench times its K steps twice (both one-in-flight here), then 5 NMS timings: take the K steps before those
end = g[-6]
start = g[-6 - steps]
seg = rows[start + 1:end + 1]
per = len(seg) // steps
assert per * steps == len(seg), (len(seg), steps)
tot = 0.0
for k in range(per):
    d = [(seg[s * per + k][1] - seg[s * per + k][0]) / 1e3 for s in range(steps)]
    gap = [(seg[s * per + k][0] - seg[s * per + k - 1][1]) / 1e3 for s in range(steps) if s * per + k > 0]
    r = seg[k]
    n = re.sub(r'^_ZN2lp\d+', '', r[2])[:58]
    tot += sum(d) / len(d)
    print('%3d %-58s %7.1f us  gap %5.1f  lds %6d vgpr %3d wgs %5d x %d' % (k, n, sum(d) / len(d), sum(gap) / max(1, len(gap)), r[3], r[4], r[5], r[6]))
span = (seg[-1][1] - seg[0][0]) / 1e3 / steps
print('sum of durations per step %.1f us; span per step %.1f us' % (tot, span))
