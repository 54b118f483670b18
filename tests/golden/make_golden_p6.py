"""Golden vectors of the P6 and plain-PAN assemblies.  Run ONCE in the build container:

    python tests/golden/make_golden_p6.py

Same method as make_golden.py (whose helpers it re-uses): the REFERENCE's own model code builds tiny-width models from
its own configs -- yolov6s6 (EfficientRep6 + RepBiFPANNeck6), yolov6m6 (CSPBepBackbone_P6 + CSPRepBiFPANNeck_P6) and
the four plain PAN necks swapped into the stock configs with fuse_P2 off -- and only tensors are recorded.
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_golden as G   # noqa: E402  (registers the cv2 / torchvision stand-ins and imports the reference)


def load_cfg(name, width, depth, neck=None, fuse_P2=None):
    cfg = G.load_cfg(name, width)
    cfg.model.depth_multiple = depth
    if neck is not None:
        cfg.model.neck.type = neck
    if fuse_P2 is not None:
        cfg.model.backbone.fuse_P2 = fuse_P2
    return cfg


def model_case(tag, cfg, shape, seed, sigma=1.5):
    torch.manual_seed(0)
    m = G.ref_yolo.build_model(cfg, 31, 24, 37, 'cpu')
    m = G.randomize(m, 1, sigma).eval()
    with torch.no_grad():
        for t in list(m.parameters()) + list(m.buffers()):
            if t.is_floating_point():
                t.copy_(t.half().float())
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    x = torch.rand(*shape, generator=torch.Generator().manual_seed(seed)).half().float()
    with torch.no_grad():
        pred, feats = m(x.clone())
    print(tag, type(m.backbone).__name__, type(m.neck).__name__, 'pred', tuple(pred.shape))
    G.save(tag + '_weights', **{k: (v.half() if v.is_floating_point() else v) for k, v in sd.items()})
    arrays = dict(x=x.half(), pred=pred)
    arrays.update({'neck%d' % i: f for i, f in enumerate(feats)})
    G.save(tag, **arrays)


if __name__ == '__main__':
    torch.set_num_threads(4)
    W, D = 0.0625, 0.25
    model_case('s6_tiny_128x192', load_cfg('yolov6s6', W, D), (1, 3, 128, 192), 31)
    model_case('m6_tiny_128x64', load_cfg('yolov6m6', W, D), (2, 3, 128, 64), 32)
    model_case('s6pan_tiny_64x128', load_cfg('yolov6s6', W, D, 'RepPANNeck6', False), (1, 3, 64, 128), 33)
    # CSPBepBackbone_P6.forward returns P2 whatever fuse_P2 says (efficientrep.py:481-497), so the reference cannot pair it
    # with the plain P6 PAN neck; the CSP PAN neck is recorded on the Rep backbone (it then gets csp_e = 0.5, yolo.py:96-100)
    model_case('s6csppan_tiny_128x128', load_cfg('yolov6s6', W, D, 'CSPRepPANNeck_P6', False), (1, 3, 128, 128), 34)
    model_case('span_tiny_96x64', load_cfg('yololps', W, D, 'RepPANNeck', False), (1, 3, 96, 64), 35)
    model_case('mpan_tiny_64x96', load_cfg('yolov6m', W, D, 'CSPRepPANNeck', False), (1, 3, 64, 96), 36)
