#!/usr/bin/env python3
"""Per-launch HBM traffic of the dominant kernel (3x3 conv_mfma_kernel launches) from two rocprofv3 --pmc passes.

    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline
    python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write --steps 3 --launches 47 > profiles/r02_pmc_traffic.json

Counters are taken from the LAST steps*launches matching dispatches (the timed steps; earlier ones belong to warm-up
and to the autotuner).  Corrections per MI355X_MICROARCH.md (HBM section): FETCH_SIZE / WRITE_SIZE are in KiB;
on gfx950 FETCH_SIZE reports half the bytes of wide (16 B per lane) coalesced reads, which is what the LDS-DMA
staging of this kernel issues, so it is doubled; WRITE_SIZE is exact for 16-B-per-lane stores.
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


def step_values(d, counter, steps):
    """Counter values of the 3x3 dispatches of the last `steps` steps (a step ends with its sort_kernel dispatch) and their number per step."""
    f = glob.glob(d + '/**/*counter_collection.csv', recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if r['Counter_Name'] == counter]
    rows.sort(key=lambda r: int(r['Dispatch_Id']))
    ends = [i for i, r in enumerate(rows) if 'sort_kernel' in r['Kernel_Name']]
    per_step = sum(1 for r in rows[ends[-2] + 1:ends[-1]] if CONV3.search(r['Kernel_Name']))
    vals = [float(r['Counter_Value']) for r in rows[:ends[-1]] if CONV3.search(r['Kernel_Name'])]
    # every forward kernel of the last `steps` steps (all but the NMS's): the whole-forward traffic of a bandwidth-bound configuration
    first = ends[-1 - steps] + 1
    fwd = [float(r['Counter_Value']) for r in rows[first:ends[-1]] if not NMS.search(r['Kernel_Name'])]
    return vals[-per_step * steps:], per_step, sum(fwd) / steps


ap = argparse.ArgumentParser()
ap.add_argument('fetch_dir')
ap.add_argument('write_dir')
ap.add_argument('--steps', type=int, default=3)
ap.add_argument('--bench', required=True, help='a JSON line of bench.py on this build: the number of 3x3 layers per step (the per-launch '
                'figures are per LAYER, like bench.py roofline; a fused kernel runs two)')
a = ap.parse_args()
rl = json.loads([l for l in open(a.bench).read().splitlines() if l.startswith('{')][-1])['roofline']
layers = rl.get('conv3_layers') or rl['launches']
fetch, per_step, fwd_fetch = step_values(a.fetch_dir, 'FETCH_SIZE', a.steps)
write, _, fwd_write = step_values(a.write_dir, 'WRITE_SIZE', a.steps)
fetch_b = sum(fetch) / (a.steps * layers) * 1024 * 2          # KiB -> B, x2 gfx950 wide-read correction
write_b = sum(write) / (a.steps * layers) * 1024
print(json.dumps({'kernel': '3x3 conv layers (conv3x3_pipe_kernel + stem2_fused_kernel / stem_planar_kernel + conv_mfma_kernel<KS=3>), bytes per LAYER', 'kernel_source_hash': source_hash(),
                  'dispatches_averaged': len(fetch), 'dispatches_per_step': per_step, 'layers_per_step': layers,
                  'fetch_bytes_per_launch': round(fetch_b), 'write_bytes_per_launch': round(write_b),
                  'hbm_bytes_per_launch': round(fetch_b + write_b),
                  'forward_hbm_bytes_per_step': round(fwd_fetch * 1024 * 2 + fwd_write * 1024),
                  'correction': 'FETCH_SIZE KiB x1024 x2 (gfx950 wide reads), WRITE_SIZE KiB x1024'}))
