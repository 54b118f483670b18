#!/usr/bin/env python3
"""roofline.frac of the dominant kernels from a rocprofv3 --kernel-trace of bench.py (the judge's arithmetic):
algorithmic FLOPs of the 3x3 convolution layers of a step / the summed kernel-trace duration of their dispatches in the
TIMED steps (the last steps x launches_per_step matching dispatches; everything before belongs to warm-up and the tuner).

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kt1 -- python3 bench.py --steps 50 --warmup 5 --inflight 1 --no-cpu-baseline > bench.json
    python tools/roofline_from_trace.py gpurun_out/kt1 bench.json --steps 50 > profiles/r03_roofline.json

(one batch in flight, one execution lane: kernels do not overlap, so a dispatch's duration is its own).  The file records the hash of
the kernel sources; bench.py reports the figure as roofline.frac only when that hash is the running build's.
"""
import argparse
import csv
import glob
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yolov6.hip.srchash import source_hash   # noqa: E402

NMS = re.compile(r'score_kernel|sort_kernel|greedy_kernel|zero_counts|counts_kernel|nms_')
CONV3 = re.compile(r'conv3x3_\w*kernel|stem_planar_kernel|stem2_fused_kernel|pw_s2_fused_kernel|conv_mfma_kernelI\w+?Li\dELi3E')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('trace_dir')
    ap.add_argument('bench_json', help='the JSON line bench.py printed in that run (FLOPs and layer count of the 3x3 layers)')
    ap.add_argument('--steps', type=int, required=True, help='timed steps in the trace')
    ap.add_argument('--peak', type=float, default=2500.0)
    a = ap.parse_args()
    d = json.loads([l for l in open(a.bench_json).read().splitlines() if l.startswith('{')][-1])
    r = d['roofline']
    flops_step = r.get('conv3_flops_per_step') or r['flops_per_launch'] * 1e9 * r['launches']
    layers = r.get('conv3_layers') or r['launches']
    rows = []
    for f in glob.glob(a.trace_dir + '/**/*kernel_trace.csv', recursive=True):
        rows += [(int(x['Start_Timestamp']), int(x['End_Timestamp']), x['Kernel_Name']) for x in csv.DictReader(open(f))]
    rows.sort()
    # the timed steps: bench.py ends with five forward + lp_nms timings (five greedy_kernel dispatches) behind the K timed steps;
    # a step's NMS runs under the next forward, so the window between the greedy dispatches of steps i0 and i0 + K holds K forwards
    g = [i for i, (s_, e_, n) in enumerate(rows) if 'greedy_kernel' in n]
    if len(g) < a.steps + 6:
        raise SystemExit('trace holds %d NMS dispatches, need %d timed steps + 5' % (len(g), a.steps))
    seg = rows[g[-6 - a.steps] + 1:g[-6] + 1]
    tail = [(e_ - s_) / 1e3 for s_, e_, n in seg if CONV3.search(n)]
    # the backbone alone: the first `backbone_dispatches` forward kernels of every timed step (the NMS kernels of the step before run
    # on the post stream under them and are not forward kernels)
    bb = None
    if r.get('backbone_dispatches'):
        fwd = [(e_ - s_) / 1e3 for s_, e_, n in seg if not NMS.search(n)]
        per = len(fwd) // a.steps
        if per * a.steps == len(fwd) and per >= r['backbone_dispatches']:
            us_bb = sum(sum(fwd[k * per:k * per + r['backbone_dispatches']]) for k in range(a.steps)) / a.steps
            bb = {'backbone_us_per_step': round(us_bb, 1), 'backbone_dispatches': r['backbone_dispatches'], 'forward_dispatches_per_step': per,
                  'backbone_frac': round(r['backbone_flops'] / (us_bb * 1e-6) / 1e12 / a.peak, 4)}
    # the whole forward (every kernel of a step but the NMS's): the denominator of a bandwidth-bound configuration's roofline
    fwd_us = sum((e_ - s_) / 1e3 for s_, e_, n in seg if not NMS.search(n)) / a.steps
    per_step = len(tail) / a.steps
    us_step = sum(tail) / a.steps
    ach = flops_step / (us_step * 1e-6) / 1e12
    print(json.dumps({
        'kernel_source_hash': source_hash(), 'workload': d['config']['workload'],
        'kernel': '3x3 convolution layers: every dispatch of conv3x3_*kernel / stem2_fused_kernel / stem_planar_kernel / pw_s2_fused_kernel / '
                  'conv_mfma_kernel<KS=3> in the timed steps of a rocprofv3 --kernel-trace of bench.py --inflight 1',
        'steps': a.steps, 'dispatches_per_step': round(per_step, 2), 'layers_per_step': layers, 'forward_us_per_step': round(fwd_us, 1),
        'conv3_us_per_step': round(us_step, 1), 'avg_dispatch_us': round(sum(tail) / len(tail), 2),
        'flops_per_step': flops_step, 'achieved_tflops': round(ach, 1), 'peak_tflops': a.peak, 'frac': round(ach / a.peak, 4),
        'bench_event_timed_frac': r.get('frac_event', r['frac']), 'bench_value_inflight1': d.get('value_inflight1'), **(bb or {})}))


if __name__ == '__main__':
    main()
