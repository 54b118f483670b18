"""Anchor-free anchor points for the eval decode (host-side mirror).

Only the ``is_eval=True`` branch is on the hot path (reference
yolov6/assigners/anchor_generator.py:11-31): anchor centres ``(x+off, y+off)``
in grid units, row-major (y outer, x inner), plus a per-anchor stride column.
The training branch (anchor boxes for the assigners) is out of scope.
"""
import torch


def generate_anchors(feats, fpn_strides, grid_cell_size=5.0, grid_cell_offset=0.5, device='cpu', is_eval=False,
                     mode='af'):
    assert feats is not None
    if not is_eval:
        raise NotImplementedError('training-time anchors are outside the inference hot path')
    rep = 1 if mode == 'af' else 3          # 'ab' (anchor-based) repeats every point 3x
    points, strides = [], []
    for feat, stride in zip(feats, fpn_strides):
        h, w = feat.shape[2:]
        xs = torch.arange(end=w, device=device) + grid_cell_offset
        ys = torch.arange(end=h, device=device) + grid_cell_offset
        gy, gx = torch.meshgrid(ys, xs, indexing='ij')
        pts = torch.stack([gx, gy], axis=-1).to(torch.float).reshape([-1, 2])
        st = torch.full((h * w, 1), stride, dtype=torch.float, device=device)
        points.append(pts.repeat(rep, 1) if rep > 1 else pts)
        strides.append(st.repeat(rep, 1) if rep > 1 else st)
    return torch.cat(points), torch.cat(strides)
