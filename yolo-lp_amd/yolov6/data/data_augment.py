"""Letterbox pre-processing of the inference path (host-side mirror of reference
yolov6/data/data_augment.py:30-61).  The reference resizes with
``cv2.resize(INTER_LINEAR)``; OpenCV is not installed in this image, so when it is
missing the resize is done by torch's bilinear interpolation with half-pixel
centres (same sampling grid; OpenCV's fixed-point weights can differ by one
intensity level).  Pre-processing is a next-row item, not part
of the measured hot path."""
import numpy as np
import torch

try:
    import cv2
except ImportError:       # the rest of this module works without it
    cv2 = None


def _resize_bilinear(im, new_wh):
    if cv2 is not None:
        return cv2.resize(im, new_wh, interpolation=cv2.INTER_LINEAR)
    t = torch.from_numpy(np.ascontiguousarray(im)).permute(2, 0, 1)[None].float()
    t = torch.nn.functional.interpolate(t, size=(new_wh[1], new_wh[0]), mode='bilinear', align_corners=False)
    return t[0].permute(1, 2, 0).round().clamp(0, 255).to(torch.uint8).numpy()


def letterbox(im, new_shape=(640, 640), color=(114, 114, 114), auto=True, scaleup=True, stride=32, return_int=False):
    """Resize keeping the aspect ratio, then pad (to the next stride multiple when ``auto``)."""
    shape = im.shape[:2]
    if isinstance(new_shape, int):
        new_shape = (new_shape, new_shape)
    elif isinstance(new_shape, list) and len(new_shape) == 1:
        new_shape = (new_shape[0], new_shape[0])
    r = min(new_shape[0] / shape[0], new_shape[1] / shape[1])
    if not scaleup:
        r = min(r, 1.0)
    new_unpad = int(round(shape[1] * r)), int(round(shape[0] * r))
    dw, dh = new_shape[1] - new_unpad[0], new_shape[0] - new_unpad[1]
    if auto:
        dw, dh = np.mod(dw, stride), np.mod(dh, stride)
    dw /= 2
    dh /= 2
    if shape[::-1] != new_unpad:
        im = _resize_bilinear(im, new_unpad)
    top, bottom = int(round(dh - 0.1)), int(round(dh + 0.1))
    left, right = int(round(dw - 0.1)), int(round(dw + 0.1))
    out = np.empty((im.shape[0] + top + bottom, im.shape[1] + left + right, im.shape[2]), dtype=im.dtype)
    out[...] = np.asarray(color, dtype=im.dtype)
    out[top:top + im.shape[0], left:left + im.shape[1]] = im
    return (out, r, (left, top)) if return_int else (out, r, (dw, dh))
