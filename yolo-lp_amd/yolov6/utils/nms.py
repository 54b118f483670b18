"""Class-agnostic NMS of the LP head output (host-side mirror of reference
yolov6/utils/nms.py:21-130).

``non_max_suppression(prediction[B,N,290]) -> list of [n_i, 28]`` rows
``[xyxy, 8 corner coords, 8 head confidences, 8 head arg-max indices (as float)]``
in descending-score order.  Reference quirks are kept on purpose:
no candidate pre-filter, the keep-mask averages ``ad4`` twice and omits ``ad5``,
``classes`` / ``agnostic`` / ``multi_label`` are accepted and ignored, and the
input tensor is multiplied by its objectness column IN PLACE.

GPU tensors are processed by the HIP kernels (score + arg-max wavefront
reductions, compaction, stable sort, greedy IoU suppression) through the C ABI;
there is no eager fallback on a GPU.  CPU tensors run the torch code below,
whose greedy step replaces ``torchvision.ops.nms`` (not installed here).
"""
import os
import time

import numpy as np
import torch

torch.set_printoptions(linewidth=320, precision=5, profile='long')
np.set_printoptions(linewidth=320, formatter={'float_kind': '{:11.5g}'.format})
os.environ['NUMEXPR_MAX_THREADS'] = str(min(os.cpu_count(), 8))

#: column ranges of the eight classification heads in a 290-wide row
HEAD_SLICES = ((13, 44), (44, 68), (68, 105), (105, 142), (142, 179), (179, 216), (216, 253), (253, 290))
MAX_NMS = 30000      # rows handed to the greedy step at most
TIME_LIMIT = 10.0    # seconds; the CPU loop gives up after this, like the reference


def xywh2xyxy(x):
    """[cx, cy, w, h] -> [x1, y1, x2, y2] for an [n, 4] tensor / array."""
    y = x.clone() if isinstance(x, torch.Tensor) else np.copy(x)
    y[:, 0] = x[:, 0] - x[:, 2] / 2
    y[:, 1] = x[:, 1] - x[:, 3] / 2
    y[:, 2] = x[:, 0] + x[:, 2] / 2
    y[:, 3] = x[:, 1] + x[:, 3] / 2
    return y


def greedy_nms(boxes, scores, iou_threshold):
    """Greedy IoU suppression with ``torchvision.ops.nms`` CPU semantics: stable
    descending sort, fp32 areas, suppress later boxes whose IoU is strictly
    greater than the threshold (compared in double).  Returns kept indices (int64)."""
    b = boxes.detach().cpu().float().numpy()
    s = scores.detach().cpu().float().numpy()
    order = np.argsort(-s, kind='stable')
    x1, y1, x2, y2 = b[:, 0], b[:, 1], b[:, 2], b[:, 3]
    areas = (x2 - x1) * (y2 - y1)
    dead = np.zeros(order.size, dtype=bool)
    keep = []
    for pos in range(order.size):
        if dead[pos]:
            continue
        i = order[pos]
        keep.append(i)
        later = order[pos + 1:]
        iw = np.maximum(np.float32(0), np.minimum(x2[i], x2[later]) - np.maximum(x1[i], x1[later]))
        ih = np.maximum(np.float32(0), np.minimum(y2[i], y2[later]) - np.maximum(y1[i], y1[later]))
        inter = iw * ih
        with np.errstate(divide='ignore', invalid='ignore'):
            iou = inter / (areas[i] + areas[later] - inter)
        dead[pos + 1:] |= iou.astype(np.float64) > float(iou_threshold)
    return torch.as_tensor(np.asarray(keep, dtype=np.int64), device=boxes.device)


def _mean8(cols):
    """Left-to-right fp32 sum of eight columns, divided by 8."""
    total = cols[0]
    for c in cols[1:]:
        total = total + c
    return total / 8.0


def non_max_suppression(prediction, conf_thres=0.25, iou_thres=0.45, classes=None, agnostic=False,
                        multi_label=False, max_det=300):
    assert 0 <= conf_thres <= 1, f'conf_thresh must be in 0.0 to 1.0, however {conf_thres} is provided.'
    assert 0 <= iou_thres <= 1, f'iou_thres must be in 0.0 to 1.0, however {iou_thres} is provided.'
    if prediction.is_cuda:
        from yolov6.hip import runtime
        return runtime.non_max_suppression(prediction, conf_thres, iou_thres, max_det)

    tik = time.time()
    output = [torch.zeros((0, 28), device=prediction.device)] * prediction.shape[0]
    for img_idx, x in enumerate(prediction):
        if not x.shape[0]:
            continue
        x[:, 13:] *= x[:, 4:5]                       # in place on the caller's tensor
        box = xywh2xyxy(x[:, :4])
        best = [torch.max(x[:, a:b], 1, keepdim=True) for a, b in HEAD_SLICES]
        conf = [v for v, _ in best]
        c = [v.squeeze() for v in conf]
        keep_mask = (_mean8(c[:7] + [c[6]]) >= conf_thres).squeeze()      # ad4 twice, ad5 not (reference quirk)
        rows = torch.cat([box, x[:, 5:13]] + conf + [i for _, i in best], 1)[keep_mask]
        if not rows.shape[0]:
            continue
        if rows.shape[0] > MAX_NMS:
            rows = rows[_mean8([rows[:, 12 + k] for k in range(8)]).argsort(descending=True)[:MAX_NMS]]
        scores = _mean8([rows[:, 12 + k] for k in range(8)])
        keep = greedy_nms(rows[:, :4], scores, iou_thres)[:max_det]
        output[img_idx] = rows[keep]
        if (time.time() - tik) > TIME_LIMIT:
            print(f'WARNING: NMS cost time exceed the limited {TIME_LIMIT}s.')
            break
    return output
