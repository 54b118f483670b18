#!/usr/bin/env python3
"""Inference entry point (same flags and ``run`` signature as reference tools/infer.py:19-107).

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


def get_args_parser(add_help=True):
    p = argparse.ArgumentParser(description='YOLO-LP inference on MI355X (HIP engine) or CPU.', add_help=add_help)
    p.add_argument('--weights', type=str, default='weights/yolov6s.pt', help='model path(s) for inference.')
    p.add_argument('--source', type=str, default='data/images', help='the source path, e.g. image-file/dir.')
    p.add_argument('--yaml', type=str, default='data/dataset.yaml', help='data yaml file.')
    p.add_argument('--img-size', nargs='+', type=int, default=[640, 640], help='the image-size(h,w) in inference size.')
    p.add_argument('--conf-thres', type=float, default=0.4, help='confidence threshold for inference.')
    p.add_argument('--iou-thres', type=float, default=0.45, help='NMS IoU threshold for inference.')
    p.add_argument('--max-det', type=int, default=1000, help='maximal inferences per image.')
    p.add_argument('--device', default='0', help='device to run our model i.e. 0 or 0,1,2,3 or cpu.')
    p.add_argument('--save-txt', action='store_true', help='save results to *.txt.')
    p.add_argument('--not-save-img', action='store_true', help='do not save visuallized inference results.')
    p.add_argument('--save-dir', type=str, help='directory to save predictions in. See --save-txt.')
    p.add_argument('--view-img', action='store_true', help='show inference results')
    p.add_argument('--classes', nargs='+', type=int, help='filter by classes, e.g. --classes 0, or --classes 0 2 3.')
    p.add_argument('--agnostic-nms', action='store_true', help='class-agnostic NMS.')
    p.add_argument('--project', default='runs/inference', help='save inference results to project/name.')
    p.add_argument('--name', default='exp', help='save inference results to project/name.')
    p.add_argument('--hide-labels', default=False, action='store_true', help='hide labels.')
    p.add_argument('--hide-conf', default=False, action='store_true', help='hide confidences.')
    p.add_argument('--half', action='store_true', help='whether to use FP16 half-precision inference.')
    args = p.parse_args()
    LOGGER.info(args)
    return args


@torch.no_grad()
def run(weights=osp.join(ROOT, 'yolov6s.pt'), source=osp.join(ROOT, 'data/images'), yaml=None, img_size=640,
        conf_thres=0.4, iou_thres=0.45, max_det=1000, device='', save_txt=False, not_save_img=False, save_dir=None,
        view_img=True, classes=None, agnostic_nms=False, project=osp.join(ROOT, 'runs/inference'), name='exp',
        hide_labels=False, hide_conf=False, half=False):
    if save_dir is None:
        save_dir = osp.join(project, name)
    if (not not_save_img or save_txt) and not osp.exists(save_dir):
        os.makedirs(save_dir)
    else:
        LOGGER.warning('Save directory already existed')
    if save_txt:
        os.makedirs(osp.join(save_dir, 'labels'), exist_ok=True)
    inferer = Inferer(source, weights, device, yaml, img_size, half)
    results = inferer.infer(conf_thres, iou_thres, classes, agnostic_nms, max_det, save_dir, save_txt, not not_save_img,
                            hide_labels, hide_conf, view_img)
    if save_txt or not not_save_img:
        LOGGER.info(f"Results saved to {save_dir}")
    return results


def main(args):
    run(**vars(args))


if __name__ == "__main__":
    main(get_args_parser())
