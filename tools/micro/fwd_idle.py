#!/usr/bin/env python3
"""Where the device sits idle in the steady-state steps of a rocprofv3 --kernel-trace of bench.py: per step the busy time (union of all
kernel intervals, any queue), the idle time, and the kernels in front of which the device was idle for more than 1 us.
    python tools/micro/fwd_idle.py <trace dir> <steps>"""
import csv, glob, re, sys, collections
rows = []
for f in glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True):
    rows += [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], int(r['Queue_Id'])) for r in csv.DictReader(open(f))]
rows.sort()
steps = int(sys.argv[2])
g = [i for i, r in enumerate(rows) if 'greedy_kernel' in r[2]]
seg = rows[g[-6 - steps] + 1:g[-6] + 1]
t0, t1 = seg[0][0], max(r[1] for r in seg)
busy, cur_end, idle_at = 0, seg[0][0], collections.Counter()
for s, e, n, q in seg:
    if s > cur_end:
        if s - cur_end > 1000:
            idle_at[re.sub(r'^_ZN2lp\d+', '', n)[:40]] += s - cur_end
        cur_end = s
    if e > cur_end:
        busy += e - cur_end
        cur_end = e
span = t1 - t0
print('per step: span %.1f us, device busy (any kernel) %.1f us, idle %.1f us; sum of kernel durations %.1f us; queues %s' % (
    span / 1e3 / steps, busy / 1e3 / steps, (span - busy) / 1e3 / steps, sum(e - s for s, e, _, _ in seg) / 1e3 / steps,
    dict(collections.Counter(q for _, _, _, q in seg))))
for n, v in idle_at.most_common(12):
    print('   idle before %-40s %6.1f us per step' % (n, v / 1e3 / steps))
