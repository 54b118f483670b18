"""LP accuracy metric of ``Evaler.eval`` (reference yolov6/core/evaler.py:153-283) as batched tensor work.

The reference walks every image and every label in python and pulls scalars off the device one by one.  Here the
matching produces a vector of integer counters (layout of ``lp_eval_counts``, include/lp_hip.h) -- on the GPU by one
kernel launch per batch (``yolov6.hip.runtime.eval_counts``), on the CPU by vectorised torch ops per image -- and
``finish`` turns the counters into the reference's seven results with its exact arithmetic.

One quirk is not reproduced: a matched label whose IoU is >= 1.0 fits none of the reference's bins and re-uses the bin
index left over from the previous matched label (UnboundLocalError if none); such labels are skipped here and counted in
``counts[UNBINNED]``.
"""
import torch

TRUE, PRED, PRED_BINS, COR, CLS, RIGHT, UNBINNED, NCOUNTS = 0, 1, 2, 12, 22, 32, 42, 43
IOU_LIST = [0.5 + i * 0.05 for i in range(10)]                                    # evaler.py:159


def _bins():
    lo = torch.tensor(IOU_LIST, dtype=torch.float64).float()
    hi = torch.tensor([v + 0.05 for v in IOU_LIST], dtype=torch.float64).float()  # ``iou_list[n] + 0.05`` (:203)
    return lo, hi


def _counts_cpu(pred, target, counts):
    """One image on the CPU: pred [n,28], target [m,20] float32."""
    counts[TRUE] += target.shape[0]
    if pred.shape[0] == 0 or target.shape[0] == 0:
        return
    b1, b2 = pred[:, :4], target[:, 8:12]
    area1 = (b1[:, 2] - b1[:, 0]) * (b1[:, 3] - b1[:, 1])
    area2 = (b2[:, 2] - b2[:, 0]) * (b2[:, 3] - b2[:, 1])
    d = (torch.min(b1[:, None, 2:], b2[:, 2:]) - torch.max(b1[:, None, :2], b2[:, :2])).clamp(0)
    inter = d[..., 0] * d[..., 1]
    iou, match = torch.max(inter / (area1[:, None] + area2 - inter), 0)            # first index on ties (CPU)
    lo, hi = _bins()
    live = ~(iou < 0.5)
    counts[PRED] += int((live & (iou >= 0.7)).sum())
    inbin = (iou[:, None] >= lo) & (iou[:, None] < hi)                             # [m,10], at most one per row
    has = inbin.any(1)
    counts[UNBINNED] += int((live & ~has).sum())
    sel = live & has
    if not sel.any():
        return
    bn = inbin.float().argmax(1)[sel]
    p, g = pred[match[sel]], target[sel]
    area = (g[:, 10] - g[:, 8]) * (g[:, 11] - g[:, 9])
    diff = (p[:, 4:12] - g[:, 12:20]).abs()
    s = diff[:, 0]
    for q in range(1, 8):                                                          # fixed summation order (as the HIP kernel)
        s = s + diff[:, q]
    is_cor = s / 8.0 < 0.1 * torch.sqrt(area)
    is_cls = (p[:, 20:28].to(torch.int64) == g[:, :8].to(torch.int64)).all(1)      # int(): truncation toward zero
    one = torch.ones_like(bn)
    counts[PRED_BINS:PRED_BINS + 10] += torch.zeros(10, dtype=torch.int64).index_add_(0, bn, one)
    counts[COR:COR + 10] += torch.zeros(10, dtype=torch.int64).index_add_(0, bn, is_cor.long())
    counts[CLS:CLS + 10] += torch.zeros(10, dtype=torch.int64).index_add_(0, bn, is_cls.long())
    counts[RIGHT:RIGHT + 10] += torch.zeros(10, dtype=torch.int64).index_add_(0, bn, (is_cor & is_cls).long())


def _pad(rows, width, device):
    n = max([r.shape[0] for r in rows] + [1])
    out = torch.zeros(len(rows), n, width, dtype=torch.float32, device=device)
    for i, r in enumerate(rows):
        if r.shape[0]:
            out[i, :r.shape[0]] = r.reshape(-1, width)[:, :width].to(device=device, dtype=torch.float32)
    cnt = torch.tensor([r.shape[0] for r in rows], dtype=torch.int32, device=device)
    return out, cnt


def counts(preds, targets):
    """preds / targets: per batch, per image tensors [n,28] / [m,20] (what ``Evaler.predict`` returns).  int64 [43]."""
    assert len(preds) == len(targets), 'predict imgs count is not match with targets!'
    dev = None
    for pb in preds:
        for p in pb:
            dev = p.device
            break
        if dev is not None:
            break
    if dev is not None and dev.type == 'cuda':
        from yolov6.hip import runtime
        c = None
        for pb, tb in zip(preds, targets):
            assert len(pb) == len(tb), 'predict batch size is not match with targets'
            det, dc = _pad(pb, 28, dev)
            tgt, tc = _pad(tb, 20, dev)
            c = runtime.eval_counts(det, dc, tgt, tc, c)
        return c.cpu() if c is not None else torch.zeros(NCOUNTS, dtype=torch.int64)
    c = torch.zeros(NCOUNTS, dtype=torch.int64)
    for pb, tb in zip(preds, targets):
        assert len(pb) == len(tb), 'predict batch size is not match with targets'
        for p, t in zip(pb, tb):
            _counts_cpu(p.float().reshape(-1, 28), t.float().reshape(-1, 20), c)
    return c


def finish(c):
    """evaler.py:245-283 on the counters: [mAP, mAP_50, mAP_75, mAP_50_95, recall, mAP_list, recall_list]."""
    c = [int(v) for v in c]
    true_cnt, pred_cnt = c[TRUE], c[PRED]
    pred_cnts, right_cnt = c[PRED_BINS:PRED_BINS + 10], c[RIGHT:RIGHT + 10]
    mAP_list, recall_list = [0.0] * 10, [0.0] * 10
    mAP_50_95, t_50_95_cnt = 0.0, 0
    right_50 = right_75 = pred_50 = pred_75 = t_right_cnt = 0
    for i in range(10):
        mAP_list[i] = right_cnt[i] / pred_cnts[i] if pred_cnts[i] > 0 else -int(right_cnt[i] == pred_cnts[i])
        mAP_50_95 += mAP_list[i] if mAP_list[i] != -1 else 0.0
        t_50_95_cnt += 1 if mAP_list[i] != -1 else 0
        right_50 += right_cnt[i]
        pred_50 += pred_cnts[i]
        if IOU_LIST[i] >= 0.75:
            right_75 += right_cnt[i]
            pred_75 += pred_cnts[i]
        if IOU_LIST[i] >= 0.7:
            t_right_cnt += right_cnt[i]
    mAP_50_95 = mAP_50_95 / t_50_95_cnt if t_50_95_cnt > 0 else 0.0
    mAP_50 = right_50 / pred_50 if pred_50 > 0 else 0.0
    mAP_75 = right_75 / pred_75 if pred_75 > 0 else 0.0
    mAP = t_right_cnt / pred_cnt if pred_cnt > 0 else 0.0
    recall = 0
    for i in range(10):
        for j in range(i + 1):
            recall_list[i] += right_cnt[j]
        recall_list[i] = recall_list[i] / true_cnt if true_cnt > 0 else 0.0
        recall += right_cnt[i]
    recall = recall / true_cnt
    return [mAP, mAP_50, mAP_75, mAP_50_95, recall, mAP_list, recall_list]
