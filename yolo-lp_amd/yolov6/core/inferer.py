"""``Inferer``: the per-image inference driver (host-side mirror of reference
yolov6/core/inferer.py:25-245).

The hot core -- ``model(img)`` then ``non_max_suppression`` (reference :80-83) --
runs on the HIP engine when the device is a GPU.  Weight preparation keeps the
reference's order: checkpoint -> float -> fuse_model -> eval -> switch_to_deploy
-> half.  Detections are rescaled to the source image, rounded, and written as
label lines ``cls*8 xywh(normalised) corners(normalised)`` (reference :100-120).
Drawing / video writing are cv2 GUI plumbing outside the hot-path scope: boxes and
corner polygons are drawn with PIL when images are saved, labels are not rendered.
"""
import math
import os
import os.path as osp
import time
from collections import deque

import numpy as np
import torch

from yolov6.utils.events import LOGGER, load_yaml
from yolov6.layers.common import DetectBackend
from yolov6.data.data_augment import letterbox
from yolov6.data.datasets import LoadData
from yolov6.utils.nms import non_max_suppression


class Inferer:
    def __init__(self, source, weights, device, yaml, img_size, half):
        self.__dict__.update(locals())
        self.device = device
        self.img_size = img_size
        cuda = self.device != 'cpu' and torch.cuda.is_available()
        self.device = torch.device(f'cuda:{device}' if cuda else 'cpu')
        self.model = DetectBackend(weights, device=self.device)
        self.stride = self.model.stride
        names = load_yaml(yaml) if yaml else {}
        self.pro_names = names.get('names')
        self.alp_names = names.get('alps')
        self.ads_names = names.get('ads')
        self.img_size = self.check_img_size(self.img_size, s=self.stride)
        self.half = half

        self.model_switch(self.model.model, self.img_size)
        if self.half & (self.device.type != 'cpu'):
            self.model.model.half()
        else:
            self.model.model.float()
            self.half = False
        if self.device.type != 'cpu':   # warm-up: builds the engine and tunes it for this shape
            self.model.model.lp_graph = True      # per-image loop = launch-bound: replay the forward as one hipGraph
            self.model(torch.zeros(1, 3, *self.img_size).to(self.device).type_as(next(self.model.model.parameters())))
        self.files = LoadData(source)
        self.source = source

    def model_switch(self, model, img_size):
        """Collapse every RepVGGBlock to its single 3x3 conv."""
        from yolov6.layers.common import RepVGGBlock
        for layer in model.modules():
            if isinstance(layer, RepVGGBlock):
                layer.switch_to_deploy()
        LOGGER.info("Switch model to deploy modality.")

    def infer(self, conf_thres, iou_thres, classes, agnostic_nms, max_det, save_dir, save_txt, save_img, hide_labels,
              hide_conf, view_img=True):
        """Run every source image through model + NMS; returns the list of rescaled ``[n, 28]`` detections."""
        fps = CalcFPS()
        results = []
        for img_src, img_path, _ in self.files:
            if self.device.type != 'cpu':      # letterbox + BGR->RGB + /255 in one HIP kernel on the uploaded frame
                from yolov6.hip import runtime
                frame = torch.from_numpy(np.ascontiguousarray(img_src)).to(self.device)
                img = runtime.preprocess_letterbox(frame, self.img_size, self.stride,
                                                   torch.float16 if self.half else torch.float32)
            else:
                img, img_src = self.precess_image(img_src, self.img_size, self.stride, self.half)
                img = img.to(self.device)
            if len(img.shape) == 3:
                img = img[None]
            if img.is_cuda:      # a new frame shape runs the kernel-variant tuner once: outside the FPS window
                from yolov6.hip import runtime
                runtime.prepare_for(self.model.model, img.shape, img.dtype)
            t1 = time.time()
            if img.is_cuda:
                # model(img) -> non_max_suppression (reference :80-83) as one call: the detections-only forward, or forward +
                # lp_nms when most anchors pass the mask (runtime.Engine.detect) -- the same detections bit for bit either way
                det = runtime.detect(self.model.model, img, conf_thres, iou_thres, max_det)[0]
            else:
                pred_results = self.model(img)
                det = non_max_suppression(pred_results, conf_thres, iou_thres, classes, agnostic_nms, max_det=max_det)[0]
            t2 = time.time()
            fps.update(1.0 / max(t2 - t1, 1e-9))

            rel_path = osp.relpath(osp.dirname(img_path), osp.dirname(self.source))
            save_path = osp.join(save_dir, rel_path, osp.basename(img_path))
            txt_path = osp.join(save_dir, rel_path, osp.splitext(osp.basename(img_path))[0])
            if save_txt or save_img:
                os.makedirs(osp.join(save_dir, rel_path), exist_ok=True)
            gn = torch.tensor(img_src.shape)[[1, 0, 1, 0]]
            gn_cor = torch.tensor(img_src.shape)[[1, 0, 1, 0, 1, 0, 1, 0]]
            if len(det):
                if det.is_cuda:
                    from yolov6.hip import runtime
                    runtime.rescale_round(img.shape[2:], det, img_src.shape)
                else:
                    det[:, :12] = self.rescale(img.shape[2:], det[:, :12], img_src.shape).round()
                rows = det.detach().float().cpu()
                if save_txt:
                    with open(txt_path + '.txt', 'a') as f:
                        for output in rows:
                            xywh = (self.box_convert(output[:4].view(1, 4)) / gn).view(-1).tolist()
                            corners_gn = (output[4:12] / gn_cor).tolist()
                            line = (*output[20:].tolist(), *xywh, *corners_gn)
                            f.write(('%g ' * len(line)).rstrip() % line + '\n')
                if save_img:
                    self.save_annotated(img_src, rows, save_path)
            elif save_img:
                self.save_annotated(img_src, [], save_path)
            results.append(det)
        LOGGER.info('Average model+NMS rate: %.1f FPS' % fps.accumulate())
        return results

    @staticmethod
    def save_annotated(img_bgr, rows, save_path):
        from PIL import Image, ImageDraw
        im = Image.fromarray(np.ascontiguousarray(img_bgr[:, :, ::-1]))
        draw = ImageDraw.Draw(im)
        for r in rows:
            draw.rectangle([float(v) for v in r[:4]], outline=(255, 64, 64), width=2)
            pts = [float(v) for v in r[4:12]]
            draw.polygon(pts, outline=(64, 255, 64))
        im.save(save_path)

    @staticmethod
    def precess_image(img_src, img_size, stride, half):
        """letterbox -> CHW RGB -> fp16/fp32 in [0, 1] (reference :191-201)."""
        image = letterbox(img_src, img_size, stride=stride)[0]
        image = image.transpose((2, 0, 1))[::-1]
        image = torch.from_numpy(np.ascontiguousarray(image))
        image = image.half() if half else image.float()
        image /= 255
        return image, img_src

    @staticmethod
    def rescale(ori_shape, boxes_and_cors, target_shape):
        """Undo the letterbox on the 12 coordinates, in place: subtract the padding, divide by the ratio, clamp
        to the source image (reference :203-228)."""
        ratio = min(ori_shape[0] / target_shape[0], ori_shape[1] / target_shape[1])
        padding = (ori_shape[1] - target_shape[1] * ratio) / 2, (ori_shape[0] - target_shape[0] * ratio) / 2
        boxes_and_cors[:, [0, 2, 4, 6, 8, 10]] -= padding[0]
        boxes_and_cors[:, [1, 3, 5, 7, 9, 11]] -= padding[1]
        boxes_and_cors[:, :] /= ratio
        for k in range(12):
            boxes_and_cors[:, k].clamp_(0, target_shape[1] if k % 2 == 0 else target_shape[0])
        return boxes_and_cors

    def check_img_size(self, img_size, s=32, floor=0):
        """Round the inference size up to a multiple of the stride; always returns [h, w]."""
        if isinstance(img_size, int):
            new_size = max(self.make_divisible(img_size, int(s)), floor)
        elif isinstance(img_size, list):
            new_size = [max(self.make_divisible(x, int(s)), floor) for x in img_size]
        else:
            raise Exception(f"Unsupported type of img_size: {type(img_size)}")
        if new_size != img_size:
            print(f'WARNING: --img-size {img_size} must be multiple of max stride {s}, updating to {new_size}')
        return new_size if isinstance(img_size, list) else [new_size] * 2

    def make_divisible(self, x, divisor):
        return math.ceil(x / divisor) * divisor

    @staticmethod
    def box_convert(x):
        """xyxy -> xywh for an [n, 4] tensor / array."""
        y = x.clone() if isinstance(x, torch.Tensor) else np.copy(x)
        y[:, 0] = (x[:, 0] + x[:, 2]) / 2
        y[:, 1] = (x[:, 1] + x[:, 3]) / 2
        y[:, 2] = x[:, 2] - x[:, 0]
        y[:, 3] = x[:, 3] - x[:, 1]
        return y


class CalcFPS:
    def __init__(self, nsamples: int = 50):
        self.framerate = deque(maxlen=nsamples)

    def update(self, duration: float):
        self.framerate.append(duration)

    def accumulate(self):
        return np.average(self.framerate) if len(self.framerate) > 1 else 0.0
