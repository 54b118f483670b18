#!/usr/bin/env python3
"""Per-kernel summary of a rocprofv3 --kernel-trace CSV: calls, mean / total duration (us), grouped by a short kernel name."""
import csv, glob, re, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name']
        m = re.match(r'(?:void )?(?:lp::)?(\w+)<?(.*)', n)
        short = m.group(1) if m else n
        if short == 'conv_mfma_kernel':
            t = re.findall(r'\(lp::ConvCfg\)(\d+)|, (\d+)', n)
            short += '<' + re.sub(r'\s+', '', n.split('<', 1)[1].rsplit('>', 1)[0])[:60] + '>'
        elif short in ('conv3x3_pipe_kernel', 'head_cls_rows_kernel', 'conv1x1_stream_kernel'):
            short += '<' + re.sub(r'\s+', '', n.split('<', 1)[1].rsplit('>', 1)[0])[:40] + '>'
        acc[short].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
tot = sum(sum(v) for v in acc.values())
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    print('%-100s calls %6d  mean %8.1f us  total %10.1f us  %5.1f%%' % (k[:100], len(v), sum(v) / len(v), sum(v), 100 * sum(v) / tot))
