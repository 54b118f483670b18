"""``LoadData``: the frame source of the inference path -- one image file, or every image / video under a directory
(host-side counterpart of reference yolov6/data/datasets.py:745-795; the training dataset classes are out of scope).

Iterating yields ``(frame, path, capture)``: ``frame`` is a BGR uint8 HWC array as ``cv2.imread`` returns it (PIL is
used when OpenCV is absent), ``capture`` is the open ``cv2.VideoCapture`` while a video is being read and ``None`` for
images; ``.type`` says which kind the last frame came from, ``len()`` is the number of source files.
"""
from pathlib import Path

import numpy as np

try:
    import cv2
except ImportError:
    cv2 = None

IMG_FORMATS = ["bmp", "jpg", "jpeg", "png", "tif", "tiff", "dng", "webp", "mpo"]
VID_FORMATS = ["mp4", "mov", "avi", "mkv"]


def imread_bgr(path):
    if cv2 is not None:
        return cv2.imread(path)
    from PIL import Image
    with Image.open(path) as im:
        return np.ascontiguousarray(np.asarray(im.convert('RGB'))[:, :, ::-1])


def _suffix(path):
    return path.rsplit('.', 1)[-1]


class LoadData:
    def __init__(self, path):
        root = Path(path).resolve()
        if root.is_dir():
            found = sorted(str(f) for f in root.rglob('*.*'))
        elif root.is_file():
            found = [str(root)]
        else:
            raise FileNotFoundError(f'Invalid path {root}')
        # images first, then videos (the reference's order); the image test is case-sensitive at listing time there too
        self.files = [f for f in found if _suffix(f) in IMG_FORMATS] + [f for f in found if _suffix(f) in VID_FORMATS]
        self.nf = len(self.files)
        self.type = 'image'
        self.cap = None

    @staticmethod
    def checkext(path):
        return 'image' if _suffix(path).lower() in IMG_FORMATS else 'video'

    def _frames_of(self, path):
        if cv2 is None:
            raise RuntimeError('video sources need OpenCV, which is not installed')
        self.cap = cv2.VideoCapture(path)
        self.frames = int(self.cap.get(cv2.CAP_PROP_FRAME_COUNT))
        try:
            while True:
                ok, frame = self.cap.read()
                if not ok:
                    return
                yield frame
        finally:
            self.cap.release()

    def __iter__(self):
        for path in self.files:
            if self.checkext(path) == 'image':
                self.type, self.cap = 'image', None
                yield imread_bgr(path), path, None
            else:
                self.type = 'video'
                for frame in self._frames_of(path):
                    yield frame, path, self.cap

    def __len__(self):
        return self.nf
