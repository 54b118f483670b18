"""Fixtures captured from the REFERENCE's own ``Inferer`` statics: run ONCE in the build
container, where /root/reference exists; only tensors are written (tests/golden/inferer_ref.npz).

    python tests/golden/make_golden_inferer.py

The reference's ``yolov6/core/inferer.py`` imports cv2 and (through nms.py) torchvision at module level; neither is
installed here.  Two module objects are registered before the import:
  * ``cv2``: constants read as 0; ``copyMakeBorder`` = constant numpy padding (pure data movement -- what letterbox
    uses it for); ``resize`` = a RECORDER that returns a flat image of the requested size, so the fixtures pin the
    letterbox GEOMETRY of resized frames (new_unpad, ratio, padding) but never bilinear pixel values (those stay
    "parity unpinned": cv2 is absent).  Frames that need no resize go through the reference's code untouched, pixel
    for pixel.
  * ``torchvision``: an empty ``ops.nms`` (not called here).
What is recorded:
  rescale_*    ``Inferer.rescale(ori_shape, det[:, :12], target_shape)`` (inferer.py:203-228) and its ``.round()``
               (:100) for seeded detections that overshoot the frame on every side;
  pre_*        ``Inferer.precess_image(frame, img_size, stride, half)`` (:191-201) on seeded uint8 BGR frames whose
               size needs no resize: output tensor (RGB, CHW, /255, fp16 or fp32);
  geo_*        for frames that DO need a resize: the size handed to cv2.resize and the final letterboxed shape.
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

resize_calls = []


class _Cv2(types.ModuleType):
    def __getattr__(self, name):            # FONT_*, INTER_*, BORDER_*, ... : constants only
        if name.isupper():
            return 0
        raise AttributeError(name)


cv2 = _Cv2('cv2')
cv2.setNumThreads = lambda n: None


def _copy_make_border(im, top, bottom, left, right, border_type, value=(0, 0, 0)):
    out = np.empty((im.shape[0] + top + bottom, im.shape[1] + left + right, im.shape[2]), im.dtype)
    out[...] = np.asarray(value, im.dtype)
    out[top:top + im.shape[0], left:left + im.shape[1]] = im
    return out


def _resize_recorder(im, dsize, interpolation=0):
    resize_calls.append((im.shape[0], im.shape[1], dsize[0], dsize[1]))
    return np.full((dsize[1], dsize[0], im.shape[2]), 7, im.dtype)


cv2.copyMakeBorder = _copy_make_border
cv2.resize = _resize_recorder
tv = types.ModuleType('torchvision')
tv.ops = types.ModuleType('torchvision.ops')
tv.ops.nms = None
sys.modules.update({'cv2': cv2, 'torchvision': tv, 'torchvision.ops': tv.ops})

from yolov6.core.inferer import Inferer as RefInferer   # noqa: E402
import yolov6.core.inferer as ref_mod                   # noqa: E402
assert ref_mod.__file__.startswith(REF)

out = {}
g = torch.Generator().manual_seed(2024)

# ---- rescale: (letterboxed tensor shape H, W) , (source frame h, w, 3) -------------------------------------------
RESCALE = [((640, 416), (1160, 720, 3)), ((640, 640), (480, 640, 3)), ((384, 640), (720, 1280, 3)),
           ((640, 640), (2000, 1500, 3)), ((96, 128), (96, 128, 3)), ((640, 448), (333, 222, 3))]
for k, (ori, tgt) in enumerate(RESCALE):
    n = 40
    det = torch.rand(n, 12, generator=g) * torch.tensor([ori[1], ori[0]] * 6) * 1.3 - 0.15 * torch.tensor([ori[1], ori[0]] * 6)
    det[0, :] = 0.0
    det[1, :] = torch.tensor([ori[1], ori[0]] * 6, dtype=torch.float32)
    det[2, :4] = torch.tensor([10.5, 20.5, 30.5, 40.5])                  # ties for round-half-even after the division
    res = RefInferer.rescale(ori, det.clone(), tgt)
    out['rescale_%d_ori' % k] = np.asarray(ori, np.int64)
    out['rescale_%d_tgt' % k] = np.asarray(tgt, np.int64)
    out['rescale_%d_in' % k] = det.numpy()
    out['rescale_%d_out' % k] = res.numpy()
    out['rescale_%d_round' % k] = res.round().numpy()
out['rescale_n'] = np.asarray(len(RESCALE))

# ---- precess_image without resize: r == 1 exactly and round(shape * r) == shape ------------------------------------
PRE = [((160, 104), 160, False), ((128, 128), 128, True), ((64, 40), 64, False), ((96, 128), 128, True),
       ((61, 128), 128, False)]
for k, ((h, w), size, half) in enumerate(PRE):
    frame = torch.randint(0, 256, (h, w, 3), generator=g, dtype=torch.uint8).numpy()
    before = len(resize_calls)
    img, src = RefInferer.precess_image(frame, [size, size], 32, half)
    assert len(resize_calls) == before, 'case %d needed a resize' % k
    out['pre_%d_frame' % k] = frame
    out['pre_%d_size' % k] = np.asarray(size)
    out['pre_%d_half' % k] = np.asarray(int(half))
    out['pre_%d_out' % k] = img.float().numpy()
out['pre_n'] = np.asarray(len(PRE))

# ---- letterbox geometry of frames that need a resize ----------------------------------------------------------------
GEO = [((1160, 720), 640), ((720, 1280), 640), ((1080, 1920), 1280), ((333, 222), 640), ((100, 1000), 640),
       ((2000, 1500), 640), ((479, 641), 640), ((48, 64), 640)]
for k, ((h, w), size) in enumerate(GEO):
    frame = np.zeros((h, w, 3), np.uint8)
    before = len(resize_calls)
    img, _ = RefInferer.precess_image(frame, [size, size], 32, False)
    calls = resize_calls[before:]
    assert len(calls) == 1
    out['geo_%d_in' % k] = np.asarray([h, w, size], np.int64)
    out['geo_%d_resized_wh' % k] = np.asarray(calls[0][2:], np.int64)            # dsize handed to cv2.resize
    out['geo_%d_shape' % k] = np.asarray(img.shape[1:], np.int64)                # letterboxed H, W
    # where the resized frame sits: rows / columns that are not the 114 border (recorder fills 7)
    inner = (img[0] * 255).round() == 7
    ys, xs = torch.where(inner)
    out['geo_%d_box' % k] = np.asarray([int(ys.min()), int(xs.min()), int(ys.max()) + 1, int(xs.max()) + 1], np.int64)
out['geo_n'] = np.asarray(len(GEO))

np.savez_compressed(os.path.join(HERE, 'inferer_ref.npz'), **out)
print('wrote inferer_ref.npz:', len(out), 'arrays;', len(resize_calls), 'recorded resizes')
