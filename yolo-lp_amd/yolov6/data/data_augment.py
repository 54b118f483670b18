"""Letterbox pre-processing of the inference path (host-side mirror of reference
yolov6/data/data_augment.py:30-61).  The reference resizes with
``cv2.resize(INTER_LINEAR)``; OpenCV is not installed in this image, so when it is
missing the resize is ``resize_linear_u8`` below: OpenCV's fixed-point scheme for
8-bit images (11-bit coefficients, horizontal pass, then the ``>>4, >>16, +2, >>2``
vertical pass) restated in numpy integer arithmetic from the published algorithm --
"parity unpinned" against cv2 itself, bit-exact against the HIP kernel
``lp_preprocess_letterbox`` (tests/test_hip_model.py).  Pre-processing is a next-row
item, not part of the measured hot path."""
import numpy as np

try:
    import cv2
except ImportError:       # the rest of this module works without it
    cv2 = None


def _linear_coef(dst, src):
    """Source index pair and 11-bit weights of every destination coordinate (cv::resize, INTER_LINEAR, 8-bit)."""
    d = np.arange(dst, dtype=np.float64)
    f = ((d + 0.5) * (src / dst) - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int32)
    f = f - s.astype(np.float32)
    lo, hi = s < 0, s >= src - 1
    f[lo], s[lo] = 0, 0
    f[hi], s[hi] = 0, src - 1
    a0 = np.rint((np.float32(1) - f) * np.float32(2048)).astype(np.int32)
    a1 = np.rint(f * np.float32(2048)).astype(np.int32)
    return s, np.minimum(s + 1, src - 1), a0, a1


def resize_linear_u8(im, new_wh):
    """uint8 HWC image -> (new_h, new_w) by fixed-point bilinear interpolation."""
    h0, w0 = im.shape[:2]
    nw, nh = new_wh
    x0, x1, a0, a1 = _linear_coef(nw, w0)
    y0, y1, b0, b1 = _linear_coef(nh, h0)
    src = im.astype(np.int32)
    hor = src[:, x0] * a0[None, :, None] + src[:, x1] * a1[None, :, None]            # [h0, nw, C], scaled by 2048
    out = (((b0[:, None, None] * (hor[y0] >> 4)) >> 16) + ((b1[:, None, None] * (hor[y1] >> 4)) >> 16) + 2) >> 2
    return out.astype(np.uint8)


def _resize_bilinear(im, new_wh):
    if cv2 is not None:
        return cv2.resize(im, new_wh, interpolation=cv2.INTER_LINEAR)
    return resize_linear_u8(im, new_wh)


def letterbox_geometry(shape, new_shape=(640, 640), auto=True, scaleup=True, stride=32):
    """(ratio, (new_w, new_h) of the resized frame, (top, bottom, left, right) padding) -- the arithmetic of letterbox."""
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
    pads = int(round(dh - 0.1)), int(round(dh + 0.1)), int(round(dw - 0.1)), int(round(dw + 0.1))
    return r, new_unpad, pads, (dw, dh)


def letterbox(im, new_shape=(640, 640), color=(114, 114, 114), auto=True, scaleup=True, stride=32, return_int=False):
    """Resize keeping the aspect ratio, then pad (to the next stride multiple when ``auto``)."""
    shape = im.shape[:2]
    r, new_unpad, (top, bottom, left, right), (dw, dh) = letterbox_geometry(shape, new_shape, auto, scaleup, stride)
    if shape[::-1] != new_unpad:
        im = _resize_bilinear(im, new_unpad)
    out = np.empty((im.shape[0] + top + bottom, im.shape[1] + left + right, im.shape[2]), dtype=im.dtype)
    out[...] = np.asarray(color, dtype=im.dtype)
    out[top:top + im.shape[0], left:left + im.shape[1]] = im
    return (out, r, (left, top)) if return_int else (out, r, (dw, dh))
