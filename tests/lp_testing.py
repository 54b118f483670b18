"""Shared helpers of the test-suite (own code; nothing here comes from the reference)."""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

SEG = (13, 44, 68, 105, 142, 179, 216, 253, 290)


def synth_pred(B, N, seed, frac_hot=0.2, obj_one=True, extent=560.0):
    """Random [B,N,290] head output: clustered boxes (IoUs straddle the threshold) and sparse high scores.
    Same construction as tests/golden/make_golden.py::synth_pred."""
    g = torch.Generator().manual_seed(seed)
    p = torch.rand(B, N, 290, generator=g) * 0.2
    centers = torch.rand(B, N // 8 + 1, 2, generator=g) * extent + 40
    cxy = centers[:, torch.arange(N) % (N // 8 + 1)] + torch.randn(B, N, 2, generator=g) * 6
    wh = torch.rand(B, N, 2, generator=g) * 60 + 30
    p[..., 0:2], p[..., 2:4] = cxy, wh
    p[..., 4] = 1.0 if obj_one else torch.rand(B, N, generator=g) * 0.5 + 0.5
    p[..., 5:13] = cxy.repeat(1, 1, 4) + torch.randn(B, N, 8, generator=g) * 20
    hot = torch.rand(B, N, generator=g) < frac_hot
    for a, b in zip(SEG[:-1], SEG[1:]):
        cls = torch.randint(0, b - a, (B, N), generator=g)
        val = torch.rand(B, N, generator=g) * 0.7 + 0.3
        cur = p[..., a:b]
        cur.scatter_(2, cls[..., None], torch.where(hot, val, cur.gather(2, cls[..., None])[..., 0])[..., None])
    return p.half().float()


def nhwc(t, cs=None):
    """NCHW torch tensor -> NHWC with channels zero-padded to ``cs``."""
    t = t.permute(0, 2, 3, 1).contiguous()
    if cs is not None and cs != t.shape[-1]:
        t = torch.nn.functional.pad(t, (0, cs - t.shape[-1]))
    return t


def rel_err(a, b):
    """max |a-b| / max(1, max|b|)"""
    return float((a.double() - b.double()).abs().max() / max(1.0, float(b.double().abs().max())))
