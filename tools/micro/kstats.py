#!/usr/bin/env python3
"""Per-kernel summary of a rocprofv3 --kernel-trace CSV: calls, mean / total duration (us) per kernel, and -- the figure
bench.py's roofline.avg_launch_ms has to agree with -- the mean duration of the 3x3 convolution launches of the TIMED steps:
the last steps * launches_per_step dispatches of conv3x3_pipe_kernel / stem_planar_kernel / conv_mfma_kernel<.., KS = 3, ..> (everything before
them is warm-up and the autotuner trying every kernel variant, which a --stats summary lumps in).

    python tools/micro/kstats.py <rocprof output dir> [--steps K --launches 47]
"""
import argparse
import csv
import glob
import re
import collections

ap = argparse.ArgumentParser()
ap.add_argument('dir')
ap.add_argument('--steps', type=int, default=0)
ap.add_argument('--launches', type=int, default=47)
a = ap.parse_args()
acc = collections.defaultdict(list)
conv3 = []
for f in glob.glob(a.dir + '/**/*kernel_trace.csv', recursive=True):
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
    for r in rows:
        n = r['Kernel_Name']
        us = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        short = re.sub(r'^_ZN2lp\d+', '', n)[:70]
        acc[short].append(us)
        if 'conv3x3_pipe' in n or 'stem_planar_kernel' in n or 'stem2_fused_kernel' in n or 'pw_s2_fused_kernel' in n or re.search(r'conv_mfma_kernelI\w+?Li\dELi3E', n):
            conv3.append(us)
tot = sum(sum(v) for v in acc.values())
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    print('%-72s calls %6d  mean %8.1f us  total %10.1f us  %5.1f%%' % (k, len(v), sum(v) / len(v), sum(v), 100 * sum(v) / tot))
if a.steps and conv3:
    tail = conv3[-a.steps * a.launches:]
    print('3x3 convolution launches of the last %d steps: %d dispatches, mean duration %.2f us, total %.1f us per step'
          % (a.steps, len(tail), sum(tail) / len(tail), sum(tail) / a.steps))
