#!/usr/bin/env python3
"""Micro-benchmark of one conv layer of the HIP engine (for rocprofv3 --pmc runs and variant A/B tests).

    LP_AUTOTUNE=0 LP_TUNE_CFG128=3 LP_TUNE_NBUF=1 python tools/conv_bench.py --cin 256 --cout 256 --hw 40 --batch 32
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from yolov6.hip import abi  # noqa: E402
from yolov6.hip.runtime import Engine  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--cin', type=int, default=256)
ap.add_argument('--cout', type=int, default=256)
ap.add_argument('--k', type=int, default=3)
ap.add_argument('--s', type=int, default=1)
ap.add_argument('--hw', type=int, default=40, help='feature-map height = width at the conv input')
ap.add_argument('--batch', type=int, default=32)
ap.add_argument('--sl', type=int, default=5, help='log2 downscale of the conv input inside the dummy graph (lower it for big maps: the network input is hw << sl)')
ap.add_argument('--iters', type=int, default=50)
ap.add_argument('--dtype', default='f16')
ap.add_argument('--variant', default='', help='force a kernel variant: cfg,nbuf (cfg 0..5 = tiles A..F, 16 / 17 = streaming 1x1 with 64 / 128 couts per wave)')
ap.add_argument('--stamps', action='store_true', help='needs LP_HIP_LIB=yolo-lp_amd/libyololp_hip_stamps.so (make stamps)')
args = ap.parse_args()

dt = {'f16': torch.float16, 'bf16': torch.bfloat16, 'f32': torch.float32}[args.dtype]
eng = Engine(dt, 'cuda:0')
eng.autotune = False
sl = args.sl
src = eng.tensor(args.cin, sl)
g = torch.Generator().manual_seed(0)
w = torch.randn(args.cout, args.cin, args.k, args.k, generator=g) * (2.0 / (args.cin * args.k * args.k)) ** 0.5
dst = eng.conv([src], w, torch.zeros(args.cout), args.k, args.s, abi.LP_ACT_RELU, sl)
eng.finish()
H = W = args.hw << sl
stamps = None
if args.stamps:
    import ctypes
    stamps = torch.zeros(1 << 20, dtype=torch.int64, device='cuda:0')     # (rows of 8: per-wave totals; from 1 << 19: prologue detail)
    eng.lib.lpdbg_set_stamps(ctypes.c_void_p(stamps.data_ptr()))
eng.bind(args.batch, H, W)
if args.variant:
    eng.set_variant(1, *[int(v) for v in args.variant.split(',')])
eng.tensor_view(src).copy_(torch.randn(args.batch, args.cin, args.hw, args.hw, generator=g).to('cuda:0', dt))
ops = eng.profile(torch.zeros(args.batch, 3, H, W, device='cuda:0', dtype=dt), reps=args.iters)
o = ops[1]
print('%s k%d s%d %d->%d @%dx%d B%d variant %s: %.1f us  %.1f TFLOP/s  %.2f TB/s (algorithmic)' % (
    args.dtype, args.k, args.s, args.cin, args.cout, args.hw, args.hw, args.batch, o['variant'], o['ms'] * 1e3,
    o['flops'] / o['ms'] / 1e9, o['bytes'] / o['ms'] / 1e9))

if stamps is not None:
    st = stamps.view(-1, 8).cpu()
    clk = st[st[:, 7] == 2].double()
    if clk.shape[0]:          # pipelined 3x3 kernel: shader clocks / 100 MHz reference ticks per wave
        ghz = (clk[:, 0] / clk[:, 1].clamp(min=1)) * 0.1
        print('in-kernel clock: median %.3f GHz (min %.3f, max %.3f) over %d waves; wave lifetime median %.1f us, max %.1f us'
              % (ghz.median(), ghz.min(), ghz.max(), clk.shape[0], clk[:, 1].median() / 100, clk[:, 1].max() / 100))
        us = lambda col: (clk[:, col] / (ghz * 1e3)).median().item()    # noqa: E731
        print('per wave (median): prologue %.2f us, epilogues %.2f us, drain of the last stores %.2f us, chunk loop %.2f us'
              % (us(2), us(3), us(4), ((clk[:, 0] - clk[:, 2] - clk[:, 3] - clk[:, 4]) / (ghz * 1e3)).median().item()))
        p2 = stamps[1 << 19:].view(-1, 8).cpu().double()
        p2 = p2[p2[:, 7] == 5]
        if p2.shape[0]:
            g = ghz.median().item() * 1e3
            print('prologue in detail (median us): index math %.2f, tile decode + request of chunks 0, 1 %.2f, bias + chunk tables %.2f, wait for my pieces %.2f, barrier + first reads %.2f'
                  % tuple((p2[:, k] / g).median().item() for k in range(5)))
        p3 = stamps[(1 << 19) + (1 << 18):].view(-1, 8).cpu().double()
        p3 = p3[p3[:, 7] == 6]
        if p3.shape[0]:
            tot = p3[:, :4].sum(1)
            print('chunk loop (stamped once per chunk: perturbs the read pipeline), share of the clocks: MFMA steps + reads %.1f %%, wait for my DMA pieces %.1f %%, barrier %.1f %%, issue of the next pieces %.1f %%'
                  % tuple((100 * p3[:, k] / tot).median().item() for k in range(4)))
        sys.exit(0)
    st = st[st[:, 7] == 1].double()
    n = st.shape[0]
    tot = st[:, 5].mean()
    print('waves %d, chunks/wave %d, mean wave lifetime %.0f cycles (s_memtime ticks)' % (n, int(st[0, 6]), tot))
    # streaming 1x1 kernel: columns are act + LDS staging / load issue / wait for next stage / LDS read + stores / MFMA
    for name, col in (('top barrier | staging', 0), ('DMA issue', 1), ('DMA wait (vmcnt)', 2), ('data barrier | stores', 3), ('compute', 4)):
        print('  %-26s %9.0f ticks  %5.1f%%   per chunk %7.0f' % (name, st[:, col].mean(), 100 * st[:, col].mean() / tot, st[:, col].mean() / st[0, 6]))
