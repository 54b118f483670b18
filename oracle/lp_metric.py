"""CPU oracle of the LP accuracy metric (TEST INFRASTRUCTURE ONLY: imported by tests/, never by the product path).

Restates ``Evaler.eval`` of the reference (yolov6/core/evaler.py:153-283) with explicit scalar loops over numpy
float32 values: per image the IoU matrix of predictions x targets (``box_iou``, yolov6/utils/general.py:93-115),
for every target the best prediction (max over predictions, first index on ties = torch's CPU ``max``), then the
0.05-wide IoU bins, the corner test and the 8-character class test, and finally the ratios.

Pinned by tests/golden/metric_*.npz, which tests/golden/make_golden_metric.py produced by calling the reference's
own ``Evaler.eval`` on the same seeded inputs in the build container.

Quirks of the reference kept here on purpose:
  * thresholds are python doubles (0.5 + i*0.05, and that + 0.05) compared with float32 IoUs, i.e. rounded to
    float32 by torch's scalar promotion;
  * a matched target whose IoU fits no bin (IoU >= float32(1.0), identical boxes) re-uses ``iou_idx`` of the
    previous matched target -- across images -- or raises UnboundLocalError if there was none (evaler.py:202-205);
  * ``pred_cnt`` counts targets with IoU >= 0.7 while the per-bin ``pred_cnts`` are filled by a second pass;
  * mAP_list[i] = -1 for an empty bin, which then drops out of mAP@.5:.95 (evaler.py:254-256).
"""
import numpy as np

F = np.float32
IOU_LIST = [0.5 + i * 0.05 for i in range(10)]           # evaler.py:159
BIN_LO = np.array(IOU_LIST, dtype=np.float64).astype(F)
BIN_HI = np.array([v + 0.05 for v in IOU_LIST], dtype=np.float64).astype(F)   # evaler.py:203 ``iou_list[n] + 0.05``

# counts vector shared with the HIP kernel (include/lp_hip.h, lp_eval_counts)
I_TRUE, I_PRED, I_PRED_BINS, I_COR, I_CLS, I_RIGHT, I_UNBINNED, N_COUNTS = 0, 1, 2, 12, 22, 32, 42, 43


def box_iou(b1, b2):
    """general.py:93-115 for one pair, float32 throughout, same operation order."""
    a1 = F(F(b1[2] - b1[0]) * F(b1[3] - b1[1]))
    a2 = F(F(b2[2] - b2[0]) * F(b2[3] - b2[1]))
    dx = F(min(b1[2], b2[2]) - max(b1[0], b2[0]))
    dy = F(min(b1[3], b2[3]) - max(b1[1], b2[1]))
    dx = dx if dx > 0 else F(0)
    dy = dy if dy > 0 else F(0)
    inter = F(dx * dy)
    with np.errstate(invalid='ignore', divide='ignore'):
        return F(inter / F(F(a1 + a2) - inter))


def match(pred, target):
    """Best prediction per target: (iou[m], index[m]); first index wins ties (evaler.py:193-194)."""
    n, m = pred.shape[0], target.shape[0]
    best = np.zeros(m, dtype=F)
    arg = np.zeros(m, dtype=np.int64)
    for k in range(m):
        bv, bi = None, 0
        for i in range(n):
            v = box_iou(pred[i, :4], target[k, 8:12])
            if bv is None or v > bv or (np.isnan(v) and not np.isnan(bv)):   # torch max: NaN propagates
                bv, bi = v, i
        best[k], arg[k] = bv, bi
    return best, arg


def bin_of(t):
    for n in range(10):
        if t >= BIN_LO[n] and t < BIN_HI[n]:
            return n
    return -1


def counts(preds, targets, strict=True):
    """preds / targets: lists over batches of lists over images of float32 arrays [n,28] / [m,20].

    Returns the int64 counts vector.  ``strict``: follow the reference through its stale-``iou_idx`` quirk (may raise
    UnboundLocalError like the reference); otherwise such targets are skipped and counted in c[I_UNBINNED] -- what
    the batched implementations do."""
    c = np.zeros(N_COUNTS, dtype=np.int64)
    iou_idx = None
    for pb, tb in zip(preds, targets):
        assert len(pb) == len(tb)
        for pred, target in zip(pb, tb):
            pred = np.asarray(pred, dtype=F).reshape(-1, 28)
            target = np.asarray(target, dtype=F).reshape(-1, 20)
            c[I_TRUE] += target.shape[0]
            if pred.shape[0] == 0 or target.shape[0] == 0:
                continue
            iou, arg = match(pred, target)
            for k in range(target.shape[0]):
                t = iou[k]
                if t < F(0.5):
                    continue
                if t >= F(0.7):
                    c[I_PRED] += 1
                b = bin_of(t)
                if b >= 0:
                    iou_idx = b
                elif strict:
                    if iou_idx is None:
                        raise UnboundLocalError("iou_idx referenced before assignment (reference evaler.py:218)")
                else:
                    c[I_UNBINNED] += 1
                    continue
                p, g = pred[arg[k]], target[k]
                area = F(F(g[10] - g[8]) * F(g[11] - g[9]))
                s = F(0)
                for q in range(8):
                    s = F(s + abs(F(p[4 + q] - g[12 + q])))
                with np.errstate(invalid='ignore'):
                    is_cor = bool(F(s / F(8.0)) < F(F(0.1) * np.sqrt(area, dtype=F)))
                is_cls = all(int(p[20 + q]) == int(g[q]) for q in range(8))
                c[I_COR + iou_idx] += is_cor
                c[I_CLS + iou_idx] += is_cls
                c[I_RIGHT + iou_idx] += is_cor and is_cls
            for k in range(target.shape[0]):      # second pass: predictions per bin (evaler.py:234-243)
                if iou[k] < F(0.5):
                    continue
                b = bin_of(iou[k])
                if b >= 0:
                    c[I_PRED_BINS + b] += 1
    return c


def finish(c):
    """evaler.py:245-283: [mAP, mAP_50, mAP_75, mAP_50_95, recall, mAP_list, recall_list] from the counts."""
    true_cnt, pred_cnt = int(c[I_TRUE]), int(c[I_PRED])
    pred_cnts = [int(v) for v in c[I_PRED_BINS:I_PRED_BINS + 10]]
    right_cnt = [int(v) for v in c[I_RIGHT:I_RIGHT + 10]]
    mAP_list = [0.0] * 10
    mAP_50_95, t_cnt = 0.0, 0
    right_50 = right_75 = pred_50 = pred_75 = t_right = 0
    for i in range(10):
        mAP_list[i] = right_cnt[i] / pred_cnts[i] if pred_cnts[i] > 0 else -int(right_cnt[i] == pred_cnts[i])
        mAP_50_95 += mAP_list[i] if mAP_list[i] != -1 else 0.0
        t_cnt += 1 if mAP_list[i] != -1 else 0
        right_50 += right_cnt[i]
        pred_50 += pred_cnts[i]
        if IOU_LIST[i] >= 0.75:
            right_75 += right_cnt[i]
            pred_75 += pred_cnts[i]
        if IOU_LIST[i] >= 0.7:
            t_right += right_cnt[i]
    mAP_50_95 = mAP_50_95 / t_cnt if t_cnt > 0 else 0.0
    mAP_50 = right_50 / pred_50 if pred_50 > 0 else 0.0
    mAP_75 = right_75 / pred_75 if pred_75 > 0 else 0.0
    mAP = t_right / pred_cnt if pred_cnt > 0 else 0.0
    recall_list = [0.0] * 10
    recall = 0
    for i in range(10):
        for j in range(i + 1):
            recall_list[i] += right_cnt[j]
        recall_list[i] = recall_list[i] / true_cnt if true_cnt > 0 else 0.0
        recall += right_cnt[i]
    recall = recall / true_cnt          # the reference divides unguarded here (ZeroDivisionError without targets)
    return [mAP, mAP_50, mAP_75, mAP_50_95, recall, mAP_list, recall_list]


def evaluate(preds, targets):
    return finish(counts(preds, targets, strict=True))
