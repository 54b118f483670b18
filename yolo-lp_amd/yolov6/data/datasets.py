"""``LoadData``: image file / directory iterator of the inference path (host-side
mirror of reference yolov6/data/datasets.py:745-795).  Frames are BGR uint8 HWC
arrays like ``cv2.imread`` returns; PIL is used when OpenCV is absent.  Video
sources need OpenCV.  The training dataset classes are out of scope."""
import glob
import os
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


class LoadData:
    def __init__(self, path):
        p = str(Path(path).resolve())
        if os.path.isdir(p):
            files = sorted(glob.glob(os.path.join(p, '**/*.*'), recursive=True))
        elif os.path.isfile(p):
            files = [p]
        else:
            raise FileNotFoundError(f'Invalid path {p}')
        imgp = [i for i in files if i.split('.')[-1] in IMG_FORMATS]
        vidp = [v for v in files if v.split('.')[-1] in VID_FORMATS]
        self.files = imgp + vidp
        self.nf = len(self.files)
        self.type = 'image'
        self.cap = None
        if any(vidp):
            self.add_video(vidp[0])

    @staticmethod
    def checkext(path):
        return 'image' if path.split('.')[-1].lower() in IMG_FORMATS else 'video'

    def __iter__(self):
        self.count = 0
        return self

    def __next__(self):
        if self.count == self.nf:
            raise StopIteration
        path = self.files[self.count]
        if self.checkext(path) == 'video':
            self.type = 'video'
            ret_val, img = self.cap.read()
            while not ret_val:
                self.count += 1
                self.cap.release()
                if self.count == self.nf:
                    raise StopIteration
                path = self.files[self.count]
                self.add_video(path)
                ret_val, img = self.cap.read()
        else:
            self.count += 1
            img = imread_bgr(path)
        return img, path, self.cap

    def add_video(self, path):
        if cv2 is None:
            raise RuntimeError('video sources need OpenCV, which is not installed')
        self.frame = 0
        self.cap = cv2.VideoCapture(path)
        self.frames = int(self.cap.get(cv2.CAP_PROP_FRAME_COUNT))

    def __len__(self):
        return self.nf
