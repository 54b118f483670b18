"""Golden vectors of the LP accuracy metric (run ONCE in the build container, where /root/reference exists):

    python tests/golden/make_golden_metric.py

Calls the REFERENCE's own ``Evaler.eval`` (yolov6/core/evaler.py:153-283) on seeded synthetic detections / labels and
records inputs and outputs only.  evaler.py imports cv2, torchvision and pycocotools at module level (none is used by
``eval``, none is installed here): empty module objects are registered for them first.  ``eval`` is called unbound
with a stand-in ``self`` whose ``eval_speed`` does nothing (the method prints the timing report first, :155).
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'
assert os.path.isdir(REF), 'reference not present: goldens can only be regenerated in the build container'
sys.path = [p for p in sys.path if os.path.abspath(p or '.') != REPO]
sys.path.insert(0, REF)

cv2 = types.ModuleType('cv2')
cv2.setNumThreads = lambda n: None
tv = types.ModuleType('torchvision')
tv.ops = types.ModuleType('torchvision.ops')
tv.ops.nms = None
pc, pcc, pce = types.ModuleType('pycocotools'), types.ModuleType('pycocotools.coco'), types.ModuleType('pycocotools.cocoeval')
pcc.COCO = pce.COCOeval = object
sys.modules.update({'cv2': cv2, 'torchvision': tv, 'torchvision.ops': tv.ops, 'pycocotools': pc,
                    'pycocotools.coco': pcc, 'pycocotools.cocoeval': pce})

import yolov6.core.evaler as ref_evaler   # noqa: E402
assert ref_evaler.__file__.startswith(REF)


def synth(seed, n_batches, batch, max_pred, max_tgt, hw=640.0, p_empty_pred=0.1, p_empty_tgt=0.1):
    """Detections [n,28] (xyxy, 8 corner coords, 8 confs, 8 class ids) and labels [m,20] (8 class ids, xyxy, 8 corner
    coords) with IoUs spread over [0.3, 1): labels are jittered copies of some detections."""
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.rand(*s, generator=g)
    preds, targets = [], []
    for _ in range(n_batches):
        pb, tb = [], []
        for _ in range(batch):
            n = 0 if r(1).item() < p_empty_pred else int(torch.randint(1, max_pred + 1, (1,), generator=g))
            m = 0 if r(1).item() < p_empty_tgt else int(torch.randint(1, max_tgt + 1, (1,), generator=g))
            cxy = r(n, 2) * (hw - 120) + 60
            wh = r(n, 2) * 80 + 20
            box = torch.cat([cxy - wh / 2, cxy + wh / 2], 1)
            cor = torch.cat([box[:, :2], box[:, 2:3], box[:, 1:2], box[:, 2:], box[:, 0:1], box[:, 3:4]], 1) + (r(n, 8) - 0.5) * 4
            conf = r(n, 8)
            cls = torch.randint(0, 24, (n, 8), generator=g).float()
            pred = torch.cat([box, cor, conf, cls], 1)
            if n > 0 and m > 0:
                src = torch.randint(0, n, (m,), generator=g)
                jit = (r(m, 4) - 0.5) * wh[src].repeat(1, 2) * r(m, 1) * 0.9          # up to +-45 % of the size
                tbox = box[src] + jit
                tcor = cor[src] + (r(m, 8) - 0.5) * wh[src].mean(1, keepdim=True) * r(m, 1) * 0.5
                tcls = cls[src].clone()
                flip = r(m) < 0.3
                tcls[flip, 0] = (tcls[flip, 0] + 1) % 24
                tgt = torch.cat([tcls, tbox, tcor], 1)
            else:
                tgt = torch.cat([torch.randint(0, 24, (m, 8), generator=g).float(), r(m, 2) * 300, r(m, 2) * 300 + 320, r(m, 8) * hw], 1)
            pb.append(pred.float())
            tb.append(tgt.float())
        preds.append(pb)
        targets.append(tb)
    return preds, targets


def crafted():
    """Edges: duplicate detections (argmax tie -> first), IoU on bin boundaries made of exactly representable
    coordinates (4/8 = 0.5, 7/10 = 0.7 as float32 ratios), a label nobody reaches, images without detections / labels."""
    def det(box, cls, cor=None):
        x1, y1, x2, y2 = box
        cor = cor or [x1, y1, x2, y1, x2, y2, x1, y2]
        return [x1, y1, x2, y2] + cor + [0.9] * 8 + cls
    def lab(box, cls, cor=None):
        x1, y1, x2, y2 = box
        cor = cor or [x1, y1, x2, y1, x2, y2, x1, y2]
        return cls + [x1, y1, x2, y2] + cor
    c = [1, 2, 3, 4, 5, 6, 7, 8]
    img0_p = [det((0, 0, 8, 8), c), det((0, 0, 8, 8), [9] * 8), det((100, 100, 110, 110), c), det((200, 200, 210, 210), c)]
    img0_t = [lab((0, 0, 8, 4), c),                      # IoU 32/64 = 0.5 with detections 0 and 1: tie -> 0, classes right
              lab((100, 100, 110, 107), c),              # IoU 70/100 = 0.7
              lab((200, 200, 210, 209.5), c, [200, 200, 210, 200, 210, 209.5, 200, 230]),   # IoU 0.95, corners off
              lab((400, 400, 420, 420), c)]              # unreachable
    img1_p = [det((10, 10, 50, 30), [0] * 8)]
    img1_t = []
    img2_p = []
    img2_t = [lab((10, 10, 50, 30), c)]
    img3_p = [det((10, 10, 50, 30), [3.7, 2.2, 1.9, 0.5, 4.0, 5.1, 6.99, 7.0])]
    img3_t = [lab((10, 10, 50, 29), [3.2, 2.9, 1.1, 0.0, 4.5, 5.0, 6.0, 7.9])]          # int() truncation makes the classes equal
    t = lambda rows, w: torch.tensor(rows, dtype=torch.float32).reshape(-1, w)
    return [[t(img0_p, 28), t(img1_p, 28)], [t(img2_p, 28), t(img3_p, 28)]], [[t(img0_t, 20), t(img1_t, 20)], [t(img2_t, 20), t(img3_t, 20)]]


def pack(prefix, lists, width):
    flat = [a for b in lists for a in b]
    return {prefix + '_rows': torch.cat([a.reshape(-1, width) for a in flat], 0).numpy(),
            prefix + '_len': np.array([a.shape[0] for a in flat], dtype=np.int64),
            prefix + '_batch': np.array([len(b) for b in lists], dtype=np.int64)}


def case(tag, preds, targets):
    self = types.SimpleNamespace(eval_speed=lambda task: None)
    out = ref_evaler.Evaler.eval(self, [[p.clone() for p in b] for b in preds], [[t.clone() for t in b] for t_, b in zip(preds, targets)], None, 'val')
    mAP, mAP_50, mAP_75, mAP_50_95, recall, mAP_list, recall_list = out
    arrays = dict(scalars=np.array([mAP, mAP_50, mAP_75, mAP_50_95, recall], dtype=np.float64),
                  mAP_list=np.array(mAP_list, dtype=np.float64), recall_list=np.array(recall_list, dtype=np.float64))
    arrays.update(pack('pred', preds, 28))
    arrays.update(pack('tgt', targets, 20))
    np.savez_compressed(os.path.join(HERE, tag + '.npz'), **arrays)
    print(tag, [round(float(v), 4) for v in arrays['scalars']], [round(float(v), 3) for v in mAP_list])


if __name__ == '__main__':
    torch.set_num_threads(4)
    case('metric_synth_a', *synth(21, 3, 4, 40, 6))
    case('metric_synth_b', *synth(22, 2, 8, 300, 3, p_empty_pred=0.2, p_empty_tgt=0.3))
    case('metric_crafted', *crafted())
