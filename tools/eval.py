#!/usr/bin/env python3
"""Evaluation entry point (flags of reference tools/eval.py:25-48 that concern the hot path).

``--task speed`` runs the reference's speed protocol (pre-process / inference / NMS ms per image,
evaler.py:507-513) over the images of ``--data`` (a directory of images, or a dataset yaml whose ``val`` entry is
one).  ``--task val`` / ``test`` additionally read the label files next to the images (``.../images/x.jpg`` ->
``.../labels/x.txt`` or ``x.txt`` beside the image; rows of 8 character ids, xywh box, 8 corner coordinates, all
normalised, datasets.py:260-340) and report the LP accuracy metric of ``Evaler.eval`` (evaler.py:153-283).
"""
import argparse
import os
import os.path as osp
import sys

import numpy as np
import torch

ROOT = os.getcwd()
if str(ROOT) not in sys.path:
    sys.path.append(str(ROOT))

from yolov6.core.evaler import Evaler                   # noqa: E402
from yolov6.utils.events import LOGGER, load_yaml        # noqa: E402
from yolov6.utils.general import increment_name          # noqa: E402


def boolean_string(s):
    if s not in {'False', 'True'}:
        raise ValueError('Not a valid boolean string')
    return s == 'True'


def get_args_parser(add_help=True):
    p = argparse.ArgumentParser(description='YOLO-LP evaluation (speed protocol)', add_help=add_help)
    p.add_argument('--data', type=str, default='./data/dataset.yaml', help='dataset.yaml path or image directory')
    p.add_argument('--weights', type=str, default='./weights/yolov6s.pt', help='model.pt path(s)')
    p.add_argument('--batch-size', type=int, default=32, help='batch size')
    p.add_argument('--img-size', type=int, default=640, help='inference size (pixels)')
    p.add_argument('--conf-thres', type=float, default=0.03, help='confidence threshold')
    p.add_argument('--iou-thres', type=float, default=0.65, help='NMS IoU threshold')
    p.add_argument('--task', default='val', help='val, test, or speed')
    p.add_argument('--device', default='0', help='cuda device, i.e. 0 or 0,1,2,3 or cpu')
    p.add_argument('--half', default=False, action='store_true', help='whether to use fp16 infer')
    p.add_argument('--save_dir', type=str, default='runs/val/', help='evaluation save dir')
    p.add_argument('--name', type=str, default='exp', help='save evaluation results to save_dir/name')
    args = p.parse_args()
    LOGGER.info(args)
    return args


def read_labels(img_path):
    """Rows [m,20] of the image's label file (8 ids, xywh, 8 corner coordinates, normalised), or an empty array."""
    stem = osp.splitext(img_path)[0] + '.txt'
    sep = os.sep + 'images' + os.sep
    cands = [stem]
    if sep in stem:
        cands.insert(0, stem[::-1].replace(sep[::-1], (os.sep + 'labels' + os.sep)[::-1], 1)[::-1])
    for p in cands:
        if osp.isfile(p):
            rows = [ln.split() for ln in open(p).read().strip().splitlines() if ln.strip()]
            return np.array(rows, dtype=np.float32).reshape(-1, 20)
    return np.zeros((0, 20), dtype=np.float32)


def letterbox_labels(labels, h0, w0, ratio, pad, H, W):
    """datasets.py:136-207 for the val path: normalised labels of the original image -> normalised labels of the
    letterboxed H x W frame (box back to xywh), clipped like the reference."""
    labels = labels.copy()
    if not labels.size:
        return labels
    w, h = ratio * w0, ratio * h0
    x1 = w * (labels[:, 8] - labels[:, 10] / 2) + pad[0]
    y1 = h * (labels[:, 9] - labels[:, 11] / 2) + pad[1]
    x2 = w * (labels[:, 8] + labels[:, 10] / 2) + pad[0]
    y2 = h * (labels[:, 9] + labels[:, 11] / 2) + pad[1]
    cor = labels[:, 12:].copy()
    cor[:, 0::2] = w * cor[:, 0::2] + pad[0]
    cor[:, 1::2] = h * cor[:, 1::2] + pad[1]
    x1, x2 = x1.clip(0, W - 1e-3), x2.clip(0, W - 1e-3)
    y1, y2 = y1.clip(0, H - 1e-3), y2.clip(0, H - 1e-3)
    cor[:, 0::2] = cor[:, 0::2].clip(0, W - 1e-3)
    cor[:, 1::2] = cor[:, 1::2].clip(0, H - 1e-3)
    labels[:, 8], labels[:, 9] = ((x1 + x2) / 2) / W, ((y1 + y2) / 2) / H
    labels[:, 10], labels[:, 11] = (x2 - x1) / W, (y2 - y1) / H
    cor[:, 0::2] /= W
    cor[:, 1::2] /= H
    labels[:, 12:] = cor
    return labels


def image_batches(src, img_size, batch_size, stride=32):
    """(uint8 [B,3,S,S], labels [T,21], paths, shapes) batches of letterboxed images of a directory (the reference's
    dataloader with rect=False); labels carry the image index of the batch in column 0 (datasets.py:250-258)."""
    from yolov6.data.datasets import LoadData
    from yolov6.data.data_augment import letterbox
    frames, labels, paths, shapes = [], [], [], []

    def batch():
        lab = torch.cat(labels, 0) if labels else torch.zeros((0, 21))
        return torch.stack(frames), lab, list(paths), list(shapes)

    for img, path, _ in LoadData(src):
        lb, ratio, pad = letterbox(img, img_size, auto=False, stride=stride)[:3]
        rows = letterbox_labels(read_labels(path), img.shape[0], img.shape[1], ratio, pad, lb.shape[0], lb.shape[1])
        if len(rows):
            out = torch.zeros((len(rows), 21))
            out[:, 0] = len(frames)
            out[:, 1:] = torch.from_numpy(rows)
            labels.append(out)
        frames.append(torch.from_numpy(np.ascontiguousarray(lb.transpose(2, 0, 1)[::-1])))
        paths.append(path)
        shapes.append(img.shape[:2])
        if len(frames) == batch_size:
            yield batch()
            frames, labels, paths, shapes = [], [], [], []
    if frames:
        yield batch()


@torch.no_grad()
def run(data, weights=None, batch_size=32, img_size=640, conf_thres=0.03, iou_thres=0.65, task='val', device='',
        half=False, model=None, dataloader=None, save_dir='', name='', **unused):
    Evaler.check_task(task)
    if task == 'train':
        save_dir = save_dir
    else:
        save_dir = str(increment_name(osp.join(save_dir, name)))
        os.makedirs(save_dir, exist_ok=True)
    Evaler.check_thres(conf_thres, iou_thres, task)
    device = Evaler.reload_device(device, model, task)
    half = device.type != 'cpu' and half
    val = Evaler(data, batch_size, img_size, conf_thres, iou_thres, device, half, save_dir)
    model = val.init_model(model, weights, task)
    model.eval()
    if dataloader is None:
        src = data
        if isinstance(data, str) and data.endswith(('.yaml', '.yml')):
            d = load_yaml(data)
            src = d[task if task in ('train', 'val', 'test') else 'val']
        dataloader = image_batches(src, img_size, batch_size)
    preds, targets, _, _ = val.predict(model, dataloader, task)
    speed = val.eval_speed(task)
    metrics = None
    if task in ('val', 'test'):
        if sum(t.shape[0] for b in targets for t in b) == 0:
            LOGGER.warning('no label files found beside the images: speed report only')
        else:
            metrics = val.eval(preds, targets, model, task)
            LOGGER.info('mAP {:.4f}  mAP@.5 {:.4f}  mAP@.75 {:.4f}  mAP@.5:.95 {:.4f}  recall {:.4f}'.format(*metrics[:5]))
    return preds, speed, metrics


def main(args):
    run(**vars(args))


if __name__ == "__main__":
    main(get_args_parser())
