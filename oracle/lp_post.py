"""ORACLE (test infrastructure, not product code) -- python side of the
post-processing restatement: a ctypes wrapper around ``lp_post_ref.c`` and an
independent, loop-free numpy restatement of the same steps for small cases
(the two are cross-checked in tests).  See ``lp_post_ref.c`` for the reference
lines followed and the parity status (greedy step: "parity unpinned",
torchvision is absent)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
SEG = (13, 44, 68, 105, 142, 179, 216, 253, 290)


def build():
    subprocess.check_call(['make', '-s', '-C', _HERE])
    return os.path.join(_HERE, '_build', 'liblp_post_ref.so')


def _lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, '_build', 'liblp_post_ref.so')
        if not os.path.exists(path):
            build()
        _LIB = ctypes.CDLL(path)
        _LIB.lp_post_ref.restype = ctypes.c_int
        _LIB.lp_post_ref.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double,
                                     ctypes.c_double, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                     ctypes.c_void_p]
    return _LIB


def nms_c(pred, conf_thres, iou_thres, max_det):
    """pred: float32 ndarray [B, N, 290] (a copy is mutated, also returned).
    -> (list of [n_i, 28] arrays, list of kept anchor-index arrays, mutated pred)."""
    p = np.ascontiguousarray(pred, dtype=np.float32).copy()
    B, N, C = p.shape
    rows = np.zeros((B, max_det, 28), np.float32)
    keep = np.zeros((B, max_det), np.int32)
    cnt = np.zeros((B,), np.int32)
    rc = _lib().lp_post_ref(p.ctypes.data, B, N, C, float(conf_thres), float(iou_thres), int(max_det),
                            rows.ctypes.data, cnt.ctypes.data, keep.ctypes.data)
    if rc != 0:
        raise ValueError('lp_post_ref failed: %d' % rc)
    return [rows[b, :cnt[b]].copy() for b in range(B)], [keep[b, :cnt[b]].copy() for b in range(B)], p


def greedy_nms_np(boxes, scores, iou_thres):
    """torchvision.ops.nms CPU semantics restated with numpy fp32 ops.  -> kept indices (int64)."""
    boxes = np.asarray(boxes, np.float32)
    scores = np.asarray(scores, np.float32)
    order = np.argsort(-scores, kind='stable')
    x1, y1, x2, y2 = (boxes[:, i] for i in range(4))
    areas = (x2 - x1) * (y2 - y1)
    sup = np.zeros(len(order), bool)
    keep = []
    zero = np.float32(0)
    for a, i in enumerate(order):
        if sup[a]:
            continue
        keep.append(i)
        rest = order[a + 1:]
        w = np.maximum(zero, np.minimum(x2[i], x2[rest]) - np.maximum(x1[i], x1[rest]))
        h = np.maximum(zero, np.minimum(y2[i], y2[rest]) - np.maximum(y1[i], y1[rest]))
        inter = w * h
        with np.errstate(divide='ignore', invalid='ignore'):
            ovr = inter / (areas[i] + areas[rest] - inter)
        sup[a + 1:] |= ovr.astype(np.float64) > float(iou_thres)
    return np.asarray(keep, np.int64)


def nms_np(pred, conf_thres, iou_thres, max_det, max_nms=30000):
    """numpy restatement of nms.py:68-125 for a batch [B, N, 290] (pred is not modified)."""
    out, kept = [], []
    for x in np.asarray(pred, np.float32):
        x = x.copy()
        x[:, 13:] *= x[:, 4:5]
        box = np.stack([x[:, 0] - x[:, 2] / 2, x[:, 1] - x[:, 3] / 2, x[:, 0] + x[:, 2] / 2, x[:, 1] + x[:, 3] / 2], 1)
        conf = [x[:, a:b].max(1) for a, b in zip(SEG[:-1], SEG[1:])]
        idx = [x[:, a:b].argmax(1).astype(np.float32) for a, b in zip(SEG[:-1], SEG[1:])]
        c = conf
        mean = (c[0] + c[1] + c[2] + c[3] + c[4] + c[5] + c[6] + c[6]) / np.float32(8.0)
        mask = mean >= np.float32(conf_thres)
        det = np.concatenate([box, x[:, 5:13]] + [v[:, None] for v in conf] + [v[:, None] for v in idx], 1)
        sel = np.nonzero(mask)[0]
        det = det[sel]
        score = (det[:, 12] + det[:, 13] + det[:, 14] + det[:, 15] + det[:, 16] + det[:, 17] + det[:, 18]
                 + det[:, 19]) / np.float32(8.0)
        if len(det) > max_nms:
            top = np.argsort(-score, kind='stable')[:max_nms]
            det, score, sel = det[top], score[top], sel[top]
        k = greedy_nms_np(det[:, :4], score, iou_thres)[:max_det]
        out.append(det[k])
        kept.append(sel[k])
    return out, kept
