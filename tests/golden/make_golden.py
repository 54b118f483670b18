"""Golden-vector generator (run ONCE in the build container, where
/root/reference exists; the fixtures it writes are committed).

    PYTHONPATH=/root/reference python tests/golden/make_golden.py

It imports the REFERENCE's model code (yolov6.models.yolo, yolov6.layers.common, ...)
and the reference's own ``non_max_suppression`` and records inputs/outputs only
(tensors as .npz, no reference source).  The reference's nms.py imports cv2
(only for ``cv2.setNumThreads``) and torchvision (only for ``torchvision.ops.nms``);
neither is installed here, so two tiny module objects are registered first: cv2
with a no-op ``setNumThreads`` and torchvision whose ``ops.nms`` is the restated
greedy selection of oracle/lp_post.py.  Hence everything in nms.py EXCEPT the
greedy step is pinned by the reference's own lines; the greedy step is
"parity unpinned" (see oracle/lp_post_ref.c).
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'
assert os.path.isdir(REF), 'reference not present: goldens can only be regenerated in the build container'
sys.path = [p for p in sys.path if os.path.abspath(p or '.') != REPO]
sys.path.insert(0, REF)

spec = importlib.util.spec_from_file_location('lp_post', os.path.join(REPO, 'oracle', 'lp_post.py'))
lp_post = importlib.util.module_from_spec(spec)
spec.loader.exec_module(lp_post)

cv2 = types.ModuleType('cv2')
cv2.setNumThreads = lambda n: None
tv = types.ModuleType('torchvision')
tv.ops = types.ModuleType('torchvision.ops')
tv.ops.nms = lambda boxes, scores, thr: torch.from_numpy(
    lp_post.greedy_nms_np(boxes.numpy(), scores.numpy(), thr))
sys.modules.update({'cv2': cv2, 'torchvision': tv, 'torchvision.ops': tv.ops})

import yolov6.models.yolo as ref_yolo                       # noqa: E402
from yolov6.utils.torch_utils import fuse_model              # noqa: E402
from yolov6.layers.common import RepVGGBlock                 # noqa: E402
from yolov6.utils.nms import non_max_suppression as ref_nms  # noqa: E402
assert ref_yolo.__file__.startswith(REF)


class AttrDict(dict):
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


def wrap(d):
    return AttrDict({k: wrap(v) for k, v in d.items()}) if isinstance(d, dict) else d


def load_cfg(name, width=None):
    ns = {}
    exec(open(os.path.join(REF, 'configs', name + '.py')).read(), ns)
    cfg = wrap({'model': ns['model'], 'training_mode': 'repvgg'})   # tools/train.py:84-85 default
    if width is not None:
        cfg.model.width_multiple = width
    return cfg


def randomize(model, seed=1, sigma=0.35, bias_sigma=0.5):
    """Same recipe as yolov6/utils/synth.py::randomize of the build (kept textually separate on purpose)."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, m in model.named_modules():
            if isinstance(m, nn.BatchNorm2d):
                n = m.num_features
                m.running_mean.copy_(torch.randn(n, generator=g) * 0.1)
                m.running_var.copy_(torch.rand(n, generator=g) + 0.5)
                m.weight.copy_(torch.rand(n, generator=g) + 0.5)
                m.bias.copy_(torch.randn(n, generator=g) * 0.1)
            elif isinstance(m, nn.Conv2d) and name.startswith('detect.') and '_preds.' in name:
                m.weight.copy_(torch.randn(m.weight.shape, generator=g) * sigma)
                m.bias.add_(torch.randn(m.bias.shape, generator=g) * bias_sigma)
    return model


def build(name, width=None, sigma=0.35):
    torch.manual_seed(0)
    m = ref_yolo.build_model(load_cfg(name, width), 31, 24, 37, 'cpu')
    return randomize(m, 1, sigma).eval()


def deploy(m):
    m = fuse_model(m).eval()
    for layer in m.modules():
        if isinstance(layer, RepVGGBlock):
            layer.switch_to_deploy()
    return m


def save(name, **arrays):
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **{k: (v.numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in arrays.items()})
    print('%-28s %8.1f KB' % (name + '.npz', os.path.getsize(path) / 1024))


def model_case(tag, name, width, shape, sigma, seed, weights_tag=None):
    m = build(name, width, sigma)
    # weights are rounded to fp16-representable values so the fixture stores them in half the bytes
    with torch.no_grad():
        for t in list(m.parameters()) + list(m.buffers()):
            if t.is_floating_point():
                t.copy_(t.half().float())
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    x = torch.rand(*shape, generator=torch.Generator().manual_seed(seed)).half().float()
    with torch.no_grad():
        pred, feats = m(x.clone())
        bb = m.backbone(x.clone())
        m2 = deploy(m)
        pred2, _ = m2(x.clone())
    print(tag, 'pred', tuple(pred.shape), 'fused-vs-unfused max diff', (pred - pred2).abs().max().item())
    if weights_tag:   # weights are shared between cases of one model: stored once
        save(weights_tag, **{k: (v.half() if v.is_floating_point() else v) for k, v in sd.items()})
    arrays = dict(x=x.half(), pred=pred)
    arrays.update({'neck%d' % i: f for i, f in enumerate(feats)})
    arrays.update({'bb%d' % i: f for i, f in enumerate(bb)})
    save(tag, **arrays)
    return pred


def nms_case(tag, pred, conf, iou, max_det):
    p = pred.clone()
    out = ref_nms(p, conf, iou, max_det=max_det)
    half_ok = torch.equal(pred, pred.half().float())       # fp16-representable inputs are stored as fp16
    arrays = dict(pred=pred.half() if half_ok else pred, conf=conf, iou=iou, max_det=max_det, counts=np.array([len(o) for o in out]))
    if not torch.equal(p, pred):                            # in-place obj*cls product changed the input
        arrays['pred_after'] = p
    for b, o in enumerate(out):
        arrays['det%d' % b] = o
    print(tag, 'counts', [len(o) for o in out])
    save(tag, **arrays)


def synth_pred(B, N, seed, frac_hot=0.2, obj_one=True):
    """Random head output: clustered boxes (so IoUs straddle the threshold) and sparse high scores."""
    g = torch.Generator().manual_seed(seed)
    p = torch.rand(B, N, 290, generator=g) * 0.2
    centers = torch.rand(B, N // 8 + 1, 2, generator=g) * 560 + 40
    cxy = centers[:, torch.arange(N) % (N // 8 + 1)] + torch.randn(B, N, 2, generator=g) * 6
    wh = torch.rand(B, N, 2, generator=g) * 60 + 30
    p[..., 0:2], p[..., 2:4] = cxy, wh
    p[..., 4] = 1.0 if obj_one else torch.rand(B, N, generator=g) * 0.5 + 0.5
    p[..., 5:13] = cxy.repeat(1, 1, 4) + torch.randn(B, N, 8, generator=g) * 20
    hot = torch.rand(B, N, generator=g) < frac_hot
    for a, b in zip(lp_post.SEG[:-1], lp_post.SEG[1:]):
        cls = torch.randint(0, b - a, (B, N), generator=g)
        val = torch.rand(B, N, generator=g) * 0.7 + 0.3
        cur = p[..., a:b]
        cur.scatter_(2, cls[..., None], torch.where(hot, val, cur.gather(2, cls[..., None])[..., 0])[..., None])
    return p.half().float()


def crafted_pred():
    """Ties and threshold edges: equal scores (stable order by index), IoU exactly 1/3 and 0.5,
    a zero-area box, an argmax tie inside a segment, mean-mask quirk (ad4 counted twice, ad5 not)."""
    N = 16
    p = torch.zeros(1, N, 290)
    p[..., 4] = 1.0
    boxes = [(10, 10, 4, 2), (12, 10, 4, 2), (11, 10, 4, 2), (10, 10, 4, 2),     # IoU(0,1)=1/3, IoU(0,2)=3/5, 3==0
             (50, 50, 0, 0), (50, 50, 0, 0), (100, 100, 8, 8), (104, 100, 8, 8),  # degenerate pair, IoU(6,7)=1/3
             (200, 200, 10, 10), (200, 205, 10, 10), (300, 300, 6, 6), (300, 300, 6, 6),
             (400, 400, 5, 5), (402, 400, 5, 5), (500, 500, 7, 7), (500, 503, 7, 7)]
    p[0, :, :4] = torch.tensor(boxes, dtype=torch.float32)
    p[0, :, 5:13] = torch.arange(N * 8, dtype=torch.float32).reshape(N, 8)
    base = [0.9, 0.9, 0.9, 0.9, 0.8, 0.8, 0.7, 0.7, 0.6, 0.6, 0.5, 0.5, 0.45, 0.45, 0.41, 0.39]
    for i, s in enumerate(base):
        for a, b in zip(lp_post.SEG[:-1], lp_post.SEG[1:]):
            p[0, i, a + (i % (b - a))] = s
    p[0, 1, 13 + 5] = 0.9                      # argmax tie inside 'pro' for row 1 (cols 1 and 5): first wins
    p[0, 14, 253:290] = 0.0                    # row 14: ad5 max = 0 -> quirky mean keeps it, score drops
    p[0, 15, 216:253] = 0.0                    # row 15: ad4 max = 0 -> quirky mean drops it
    return p


if __name__ == '__main__':
    torch.set_num_threads(4)
    pred_s = model_case('lps_tiny_128x96', 'yololps', 0.0625, (2, 3, 128, 96), 1.5, 11, 'lps_tiny_weights')
    model_case('lps_tiny_64x160', 'yololps', 0.0625, (1, 3, 64, 160), 1.5, 12)
    model_case('v6m_tiny_96x128', 'yolov6m', 0.0625, (1, 3, 96, 128), 1.5, 13, 'v6m_tiny_weights')
    nms_case('nms_model_tiny', pred_s.half().float(), 0.06, 0.45, 300)
    nms_case('nms_synth_600', synth_pred(2, 600, 5), 0.4, 0.45, 1000)
    nms_case('nms_synth_maxdet5', synth_pred(1, 400, 6), 0.3, 0.65, 5)
    nms_case('nms_synth_obj', synth_pred(1, 300, 7, obj_one=False), 0.25, 0.5, 300)
    nms_case('nms_crafted', crafted_pred(), 0.4, 1.0 / 3.0, 300)
    nms_case('nms_empty', synth_pred(2, 100, 8, frac_hot=0.0), 0.4, 0.45, 300)

    # full-size digest: seeded full yololps / yololpn weights are regenerated from the seed on the other side
    for name in ('yololps', 'yololpn'):
        m = build(name, None, 0.35)
        x = torch.rand(1, 3, 640, 640, generator=torch.Generator().manual_seed(1234))
        with torch.no_grad():
            pred, _ = m(x)
        rows = torch.arange(0, 8400, 41)
        wsum = sum(v.double().sum().item() for v in m.state_dict().values() if v.is_floating_point())
        save('digest_%s_640' % name, rows=rows, pred_rows=pred[0, rows], colsum=pred[0].double().sum(0),
             weight_sum=wsum)
