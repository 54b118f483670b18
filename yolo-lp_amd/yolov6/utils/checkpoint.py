"""Checkpoint ingestion (host-side mirror of reference
yolov6/utils/checkpoint.py:11-44).

Reference checkpoints are whole pickled ``nn.Module`` objects
(``ckpt['model']`` / ``ckpt['ema']``, fp16), so loading one needs
``weights_only=False``: only do that for files you trust / wrote yourself.
"""
import os
import os.path as osp
import shutil

import torch

from yolov6.utils.events import LOGGER
from yolov6.utils.torch_utils import fuse_model


def _load(weights, map_location):
    return torch.load(weights, map_location=map_location, weights_only=False)


def load_state_dict(weights, model, map_location=None):
    """Copy every name- and shape-matching tensor of a checkpoint into ``model``."""
    src = _load(weights, map_location)['model'].float().state_dict()
    dst = model.state_dict()
    model.load_state_dict({k: v for k, v in src.items() if k in dst and v.shape == dst[k].shape}, strict=False)
    return model


def load_checkpoint(weights, map_location=None, inplace=True, fuse=True):
    """ckpt -> fp32 model (EMA weights if present) -> BN-folded -> eval."""
    LOGGER.info("Loading checkpoint from {}".format(weights))
    ckpt = _load(weights, map_location)
    model = ckpt['ema' if ckpt.get('ema') else 'model'].float()
    if fuse:
        LOGGER.info("\nFusing model...")
        model = fuse_model(model)
    return model.eval()


def save_checkpoint(ckpt, is_best, save_dir, model_name=""):
    os.makedirs(save_dir, exist_ok=True)
    filename = osp.join(save_dir, model_name + '.pt')
    torch.save(ckpt, filename)
    if is_best:
        shutil.copyfile(filename, osp.join(save_dir, 'best_ckpt.pt'))
