#!/usr/bin/env python3
"""Inference entry point: the flags and the ``run`` keyword arguments of reference tools/infer.py:19-107.

    python tools/infer.py --weights weights/yololps.pt --source data/images --yaml data/dataset.yaml [--half]
"""
import argparse
import os
import os.path as osp
import sys

import torch

ROOT = os.getcwd()
if str(ROOT) not in sys.path:
    sys.path.append(str(ROOT))

from yolov6.utils.events import LOGGER      # noqa: E402
from yolov6.core.inferer import Inferer      # noqa: E402

# (flag, argparse keywords): the reference's flag set, defaults included
_FLAGS = [
    ('--weights', dict(type=str, default='weights/yolov6s.pt', help='checkpoint (.pt) to run')),
    ('--source', dict(type=str, default='data/images', help='image file or directory of images / videos')),
    ('--yaml', dict(type=str, default='data/dataset.yaml', help='dataset yaml (class names)')),
    ('--img-size', dict(nargs='+', type=int, default=[640, 640], help='network input size, h w')),
    ('--conf-thres', dict(type=float, default=0.4, help='score threshold of the NMS')),
    ('--iou-thres', dict(type=float, default=0.45, help='IoU threshold of the NMS')),
    ('--max-det', dict(type=int, default=1000, help='detections kept per image')),
    ('--device', dict(default='0', help='GPU index (0, or 0,1,2,3) or cpu')),
    ('--save-txt', dict(action='store_true', help='write one label file per image')),
    ('--not-save-img', dict(action='store_true', help='do not write the annotated images')),
    ('--save-dir', dict(type=str, help='output directory (default: project/name)')),
    ('--view-img', dict(action='store_true', help='show the annotated frames')),
    ('--classes', dict(nargs='+', type=int, help='keep these class ids only')),
    ('--agnostic-nms', dict(action='store_true', help='class-agnostic NMS')),
    ('--project', dict(default='runs/inference', help='parent of the default output directory')),
    ('--name', dict(default='exp', help='name of the default output directory')),
    ('--hide-labels', dict(default=False, action='store_true', help='draw boxes without text')),
    ('--hide-conf', dict(default=False, action='store_true', help='draw labels without scores')),
    ('--half', dict(action='store_true', help='fp16 engine')),
]


def get_args_parser(add_help=True):
    parser = argparse.ArgumentParser(description='YOLO-LP inference on MI355X (HIP engine) or CPU.', add_help=add_help)
    for flag, kw in _FLAGS:
        parser.add_argument(flag, **kw)
    args = parser.parse_args()
    LOGGER.info(args)
    return args


@torch.no_grad()
def run(weights=osp.join(ROOT, 'yolov6s.pt'), source=osp.join(ROOT, 'data/images'), yaml=None, img_size=640,
        conf_thres=0.4, iou_thres=0.45, max_det=1000, device='', save_txt=False, not_save_img=False, save_dir=None,
        view_img=True, classes=None, agnostic_nms=False, project=osp.join(ROOT, 'runs/inference'), name='exp',
        hide_labels=False, hide_conf=False, half=False):
    save_img = not not_save_img
    out_dir = save_dir if save_dir is not None else osp.join(project, name)
    if (save_img or save_txt) and not osp.exists(out_dir):
        os.makedirs(out_dir)
    else:
        LOGGER.warning('Save directory already existed')
    if save_txt:
        os.makedirs(osp.join(out_dir, 'labels'), exist_ok=True)
    results = Inferer(source, weights, device, yaml, img_size, half).infer(
        conf_thres, iou_thres, classes, agnostic_nms, max_det, out_dir, save_txt, save_img, hide_labels, hide_conf, view_img)
    if save_txt or save_img:
        LOGGER.info(f"Results saved to {out_dir}")
    return results


def main(args):
    run(**vars(args))


if __name__ == "__main__":
    main(get_args_parser())
