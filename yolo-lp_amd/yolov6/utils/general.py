"""Box / corner decode helpers (host-side mirror of reference
yolov6/utils/general.py:29-66, :93-115).  Operation order is kept as in the
reference (e.g. ``(x1+x2)/2``, not ``a+(rb-lt)/2``) so fp32 results agree."""
import glob
import os
from pathlib import Path

import torch


def increment_name(path):
    """``runs/exp`` -> ``runs/exp1``, ``runs/exp2`` ... if the path exists."""
    path = Path(path)
    if path.exists():
        stem, suffix = (path.with_suffix(''), path.suffix) if path.is_file() else (path, '')
        for n in range(1, 9999):
            cand = f'{stem}{n}{suffix}'
            if not os.path.exists(cand):
                break
        path = Path(cand)
    return path


def find_latest_checkpoint(search_dir='.'):
    found = glob.glob(f'{search_dir}/**/last*.pt', recursive=True)
    return max(found, key=os.path.getctime) if found else ''


def dist2bbox(distance, anchor_points, box_format='xyxy'):
    """ltrb distances from an anchor point -> xyxy or xywh box."""
    lt, rb = torch.split(distance, 2, -1)
    x1y1 = anchor_points - lt
    x2y2 = anchor_points + rb
    if box_format == 'xyxy':
        return torch.cat([x1y1, x2y2], -1)
    if box_format == 'xywh':
        return torch.cat([(x1y1 + x2y2) / 2, x2y2 - x1y1], -1)


def bbox2dist(anchor_points, bbox, reg_max):
    x1y1, x2y2 = torch.split(bbox, 2, -1)
    return torch.cat([anchor_points - x1y1, x2y2 - anchor_points], -1).clip(0, reg_max - 0.01)


def dist2cor(distance, anchor_points):
    """Eight distances -> four corners (TL, BL, BR, TR):
    ``[ax-d0, ay-d1, ax-d2, ay+d3, ax+d4, ay+d5, ax+d6, ay-d7]``."""
    ax, ay = torch.split(anchor_points, 1, -1)
    d = torch.split(distance, 1, -1)
    return torch.cat([ax - d[0], ay - d[1], ax - d[2], ay + d[3], ax + d[4], ay + d[5], ax + d[6], ay - d[7]], -1)


def xywh2xyxy(x):
    """[cx, cy, w, h] -> [x1, y1, x2, y2]."""
    y = x.clone() if isinstance(x, torch.Tensor) else x.copy()
    y[..., 0] = x[..., 0] - x[..., 2] / 2
    y[..., 1] = x[..., 1] - x[..., 3] / 2
    y[..., 2] = x[..., 0] + x[..., 2] / 2
    y[..., 3] = x[..., 1] + x[..., 3] / 2
    return y


def box_iou(box1, box2):
    """Pairwise IoU of two xyxy sets -> [N, M]."""
    area1 = (box1[:, 2] - box1[:, 0]) * (box1[:, 3] - box1[:, 1])
    area2 = (box2[:, 2] - box2[:, 0]) * (box2[:, 3] - box2[:, 1])
    inter = (torch.min(box1[:, None, 2:], box2[:, 2:]) - torch.max(box1[:, None, :2], box2[:, :2])).clamp(0).prod(2)
    return inter / (area1[:, None] + area2 - inter)
