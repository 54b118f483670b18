"""Model assembly (host-side mirror of reference yolov6/models/yolo.py).

``Model.forward(x[B,3,H,W]) -> [pred[B,N,290] fp32, [f_s8, f_s16, f_s32]]``.
In eval mode on a GPU tensor the forward is executed by the HIP engine
(``yolov6.hip``: NHWC fp16/bf16/fp32 graph of hand-written gfx950 kernels);
there is no eager fallback on a GPU -- a missing extension raises.  On CPU
tensors, and in training mode, the module tree runs as plain torch ops.
"""
import math

import torch
import torch.nn as nn

from yolov6.layers.common import *  # noqa: F401,F403  (type names are eval()'d from configs)
from yolov6.utils.torch_utils import initialize_weights
from yolov6.models.efficientrep import *  # noqa: F401,F403
from yolov6.models.reppan import *  # noqa: F401,F403
from yolov6.utils.events import LOGGER


class Model(nn.Module):
    """Backbone -> neck -> LP head."""

    def __init__(self, config, channels=3, npro=None, nalp=None, nads=None, fuse_ab=False, distill_ns=False):
        super().__init__()
        num_layers = config.model.head.num_layers
        self.backbone, self.neck, self.detect = build_network(config, channels, npro, nalp, nads, num_layers,
                                                              fuse_ab=fuse_ab, distill_ns=distill_ns)
        self.stride = self.detect.stride
        self.detect.initialize_biases()
        initialize_weights(self)

    def forward(self, x):
        if x.is_cuda and not self.training and not torch.onnx.is_in_onnx_export():
            from yolov6.hip import runtime
            return runtime.model_forward(self, x)
        return self._forward_torch(x)

    def _forward_torch(self, x):
        export_mode = torch.onnx.is_in_onnx_export()
        x = self.neck(self.backbone(x))
        featmaps = [] if export_mode else list(x)
        x = self.detect(x)
        return x if export_mode is True else [x, featmaps]

    def _apply(self, fn):
        self = super()._apply(fn)
        self.detect.stride = fn(self.detect.stride)
        self.detect.grid = list(map(fn, self.detect.grid))
        from yolov6.hip import runtime
        runtime.drop_engine(self)               # weights moved / cast: the packed engine is stale
        return self


def make_divisible(x, divisor):
    """Smallest multiple of ``divisor`` that is >= x."""
    return math.ceil(x / divisor) * divisor


def build_network(config, channels, npro, nalp, nads, num_layers, fuse_ab=False, distill_ns=False):
    """Scale depth/width, resolve the type names of the config and build
    (backbone, neck, head) (reference yolo.py:54-124)."""
    m = config.model
    num_repeat = [(max(round(i * m.depth_multiple), 1) if i > 1 else i)
                  for i in (m.backbone.num_repeats + m.neck.num_repeats)]
    channels_list = [make_divisible(i * m.width_multiple, 8) for i in (m.backbone.out_channels + m.neck.out_channels)]
    block = get_block(config.training_mode)
    BACKBONE = eval(m.backbone.type)
    NECK = eval(m.neck.type)

    bb_kw = dict(in_channels=channels, channels_list=channels_list, num_repeats=num_repeat, block=block,
                 fuse_P2=m.backbone.get('fuse_P2'), cspsppf=m.backbone.get('cspsppf'))
    neck_kw = dict(channels_list=channels_list, num_repeats=num_repeat, block=block)
    if 'CSP' in m.backbone.type:
        bb_kw['csp_e'] = m.backbone.csp_e
        neck_kw['csp_e'] = m.neck.csp_e
    backbone = BACKBONE(**bb_kw)
    neck = NECK(**neck_kw)

    if distill_ns or fuse_ab:
        # The reference's yolov6/models/heads/* are un-adapted COCO heads whose
        # signatures do not match these call sites: dead code there, absent here.
        LOGGER.error('ERROR: distill_ns / fuse_ab heads are not part of the LP hot path.\n')
        raise NotImplementedError('distill_ns / fuse_ab heads')
    from yolov6.models.effidehead import Detect, build_effidehead_layer
    head_layers = build_effidehead_layer(channels_list, 1, npro, nalp, nads, reg_max=m.head.reg_max,
                                         num_layers=num_layers)
    head = Detect(npro, nalp, nads, num_layers, head_layers=head_layers, use_dfl=m.head.use_dfl)
    return backbone, neck, head


def build_model(cfg, npro, nalp, nads, device, fuse_ab=False, distill_ns=False):
    return Model(cfg, channels=3, npro=npro, nalp=nalp, nads=nads, fuse_ab=fuse_ab, distill_ns=distill_ns).to(device)
