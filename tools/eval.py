#!/usr/bin/env python3
"""Evaluation entry point (flags of reference tools/eval.py:25-48 that concern the hot path).

``--task speed`` runs the reference's speed protocol (pre-process / inference / NMS ms per image,
evaler.py:507-513) over the images of ``--data`` (a directory of images, or a dataset yaml whose ``val`` entry is
one).  The LP accuracy metrics of ``--task val`` are outside the hot-path scope.
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


def image_batches(src, img_size, batch_size, stride=32):
    """uint8 [B,3,S,S] batches of letterboxed images of a directory (the reference's dataloader with rect=False)."""
    from yolov6.data.datasets import LoadData
    from yolov6.data.data_augment import letterbox
    frames, paths, shapes = [], [], []
    for img, path, _ in LoadData(src):
        lb = letterbox(img, img_size, auto=False, stride=stride)[0]
        frames.append(torch.from_numpy(np.ascontiguousarray(lb.transpose(2, 0, 1)[::-1])))
        paths.append(path)
        shapes.append(img.shape[:2])
        if len(frames) == batch_size:
            yield torch.stack(frames), torch.zeros((0, 21)), paths, shapes
            frames, paths, shapes = [], [], []
    if frames:
        yield torch.stack(frames), torch.zeros((0, 21)), paths, shapes


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
    preds = val.predict(model, dataloader, task)
    speed = val.eval_speed(task)
    if task in ('val', 'test'):
        LOGGER.warning('accuracy metrics are outside the hot-path scope of this build; reported: speed only')
    return preds, speed


def main(args):
    run(**vars(args))


if __name__ == "__main__":
    main(get_args_parser())
