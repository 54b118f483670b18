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


def unpack_lists(z, prefix, width):
    """Inverse of tests/golden/make_golden_metric.py::pack: per batch, per image float32 tensors [n, width]."""
    rows, lens, batch = z[prefix + '_rows'], z[prefix + '_len'].tolist(), z[prefix + '_batch'].tolist()
    out, o, i = [], 0, 0
    for b in batch:
        cur = []
        for _ in range(b):
            cur.append(rows[o:o + lens[i]].reshape(-1, width).float().clone())
            o += lens[i]
            i += 1
        out.append(cur)
    return out


def synth_metric_batch(seed, B, max_pred, max_tgt, hw=640.0):
    """Seeded detections [n,28] / labels [m,20] per image with IoUs spread over the metric's bins (same recipe as the
    golden generator, kept separate on purpose)."""
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.rand(*s, generator=g)
    pb, tb = [], []
    for _ in range(B):
        n = 0 if r(1).item() < 0.1 else int(torch.randint(1, max_pred + 1, (1,), generator=g))
        m = 0 if r(1).item() < 0.1 else int(torch.randint(1, max_tgt + 1, (1,), generator=g))
        cxy, wh = r(n, 2) * (hw - 120) + 60, r(n, 2) * 80 + 20
        box = torch.cat([cxy - wh / 2, cxy + wh / 2], 1)
        cor = torch.cat([box[:, :2], box[:, 2:3], box[:, 1:2], box[:, 2:], box[:, 0:1], box[:, 3:4]], 1) + (r(n, 8) - 0.5) * 4
        cls = torch.randint(0, 24, (n, 8), generator=g).float()
        pred = torch.cat([box, cor, r(n, 8), cls], 1)
        if n > 0 and m > 0:
            src = torch.randint(0, n, (m,), generator=g)
            tbox = box[src] + (r(m, 4) - 0.5) * wh[src].repeat(1, 2) * r(m, 1) * 0.9
            tcor = cor[src] + (r(m, 8) - 0.5) * wh[src].mean(1, keepdim=True) * r(m, 1) * 0.5
            tcls = cls[src].clone()
            flip = r(m) < 0.3
            tcls[flip, 0] = (tcls[flip, 0] + 1) % 24
            tgt = torch.cat([tcls, tbox, tcor], 1)
        else:
            tgt = torch.cat([torch.randint(0, 24, (m, 8), generator=g).float(), r(m, 2) * 300, r(m, 2) * 300 + 320, r(m, 8) * hw], 1)
        pb.append(pred.float())
        tb.append(tgt.float())
    return pb, tb
