"""Decoupled license-plate head (host-side mirror).

Per pyramid level: a 1x1 stem, a 3x3 classification tower feeding eight 1x1
predictors (province, alphabet, six plate characters) and a 3x3 regression
tower feeding a box predictor (ltrb distances) and a corner predictor (four
keypoints as eight distances).  The eval branch decodes to
``[B, N, 4 xywh + 1 obj + 8 corners + npro + nalp + 6*nads]`` in fp32
(reference yolov6/models/effidehead.py:10-301; builder :304-669).
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from yolov6.layers.common import *  # noqa: F401,F403  (reference re-exports the layer names here)
from yolov6.layers.common import Conv
from yolov6.assigners.anchor_generator import generate_anchors
from yolov6.utils.general import dist2bbox, dist2cor

#: predictor families in head-layer order; the first eight are sigmoid class scores
CLS_HEADS = ('pro', 'alp', 'ad0', 'ad1', 'ad2', 'ad3', 'ad4', 'ad5')
BOX_HEADS = ('reg', 'cor')
_PER_LEVEL = 3 + len(CLS_HEADS) + len(BOX_HEADS)  # stem, cls_conv, reg_conv + 10 predictors


class Detect(nn.Module):
    """Efficient decoupled head for plate detection + recognition."""

    def __init__(self, npro=31, nalp=24, nads=37, num_layers=3, inplace=True, head_layers=None, use_dfl=True,
                 reg_max=16):
        super().__init__()
        assert head_layers is not None
        self.npro = npro
        self.nalp = nalp
        self.nads = nads
        self.no = npro + nalp + nads * 5 + 13
        self.nl = num_layers
        self.grid = [torch.zeros(1)] * num_layers
        self.prior_prob = 1e-2
        self.inplace = inplace
        self.stride = torch.tensor([8, 16, 32] if num_layers == 3 else [8, 16, 32, 64])
        self.use_dfl = use_dfl
        self.reg_max = reg_max
        self.proj_conv = nn.Conv2d(self.reg_max + 1, 1, 1, bias=False)
        self.grid_cell_offset = 0.5
        self.grid_cell_size = 5.0

        names = ['stems', 'cls_convs', 'reg_convs'] + ['%s_preds' % h for h in CLS_HEADS + BOX_HEADS]
        for name in names:
            setattr(self, name, nn.ModuleList())
        for i in range(num_layers):
            for j, name in enumerate(names):
                getattr(self, name).append(head_layers[i * _PER_LEVEL + j])

    def initialize_biases(self):
        """Zero predictor weights; class biases to the prior logit, box / corner
        biases to 1; DFL projection to linspace (reference effidehead.py:66-150)."""
        prior_logit = -math.log((1 - self.prior_prob) / self.prior_prob)
        for heads, value in ((CLS_HEADS, prior_logit), (BOX_HEADS, 1.0)):
            for h in heads:
                for conv in getattr(self, '%s_preds' % h):
                    b = conv.bias.view(-1)
                    b.data.fill_(value)
                    conv.bias = nn.Parameter(b.view(-1), requires_grad=True)
                    w = conv.weight
                    w.data.fill_(0.)
                    conv.weight = nn.Parameter(w, requires_grad=True)
        self.proj = nn.Parameter(torch.linspace(0, self.reg_max, self.reg_max + 1), requires_grad=False)
        self.proj_conv.weight = nn.Parameter(self.proj.view([1, self.reg_max + 1, 1, 1]).clone().detach(),
                                             requires_grad=False)

    def _level_outputs(self, x, i):
        """Raw predictor outputs of level ``i``; also stores the stem output back
        into ``x[i]`` like the reference does (effidehead.py:231)."""
        x[i] = self.stems[i](x[i])
        cls_feat = self.cls_convs[i](x[i])
        cls_out = [getattr(self, '%s_preds' % h)[i](cls_feat) for h in CLS_HEADS]
        reg_feat = self.reg_convs[i](x[i])
        return cls_out, self.reg_preds[i](reg_feat), self.cor_preds[i](reg_feat)

    def forward(self, x):
        if self.training:
            per_head = [[] for _ in range(len(CLS_HEADS) + 2)]
            for i in range(self.nl):
                cls_out, reg_out, cor_out = self._level_outputs(x, i)
                outs = [torch.sigmoid(o) for o in cls_out] + [reg_out, cor_out]
                for acc, o in zip(per_head, outs):
                    acc.append(o.flatten(2).permute((0, 2, 1)))
            return (x, *[torch.cat(acc, axis=1) for acc in per_head])

        anchor_points, stride_tensor = generate_anchors(
            x, self.stride, self.grid_cell_size, self.grid_cell_offset, device=x[0].device, is_eval=True, mode='af')
        widths = [self.npro, self.nalp] + [self.nads] * 6 + [4, 8]
        per_head = [[] for _ in widths]
        for i in range(self.nl):
            b, _, h, w = x[i].shape
            l = h * w
            cls_out, reg_out, cor_out = self._level_outputs(x, i)
            if self.use_dfl:
                reg_out = reg_out.reshape([-1, 4, self.reg_max + 1, l]).permute(0, 2, 1, 3)
                reg_out = self.proj_conv(F.softmax(reg_out, dim=1))
            outs = [torch.sigmoid(o) for o in cls_out] + [reg_out, cor_out]
            for acc, o, c in zip(per_head, outs, widths):
                acc.append(o.reshape([b, c, l]))
        per_head = [torch.cat(acc, axis=-1).permute(0, 2, 1) for acc in per_head]
        scores, reg_dist, cor_dist = per_head[:8], per_head[8], per_head[9]

        pred_bboxes = dist2bbox(reg_dist, anchor_points, box_format='xywh')
        pred_corners = dist2cor(cor_dist, anchor_points)
        pred_bboxes *= stride_tensor
        pred_corners *= stride_tensor
        ones = torch.ones((b, pred_bboxes.shape[1], 1), device=pred_bboxes.device, dtype=pred_bboxes.dtype)
        return torch.cat([pred_bboxes, ones, pred_corners, *scores], axis=-1)


def build_effidehead_layer(channels_list, num_anchors, npro, nalp, nads, reg_max=16, num_layers=3):
    """Flat ``nn.Sequential`` of 13 layers per level, in the order ``Detect``
    indexes them (reference effidehead.py:304-669)."""
    chx = [6, 8, 10] if num_layers == 3 else [8, 9, 10, 11]
    pred_widths = [npro, nalp] + [nads] * 6
    layers = []
    for idx in chx:
        c = channels_list[idx]
        layers.append(Conv(in_channels=c, out_channels=c, kernel_size=1, stride=1))   # stem
        layers.append(Conv(in_channels=c, out_channels=c, kernel_size=3, stride=1))   # cls tower
        layers.append(Conv(in_channels=c, out_channels=c, kernel_size=3, stride=1))   # reg tower
        for n in pred_widths:
            layers.append(nn.Conv2d(in_channels=c, out_channels=n * num_anchors, kernel_size=1))
        layers.append(nn.Conv2d(in_channels=c, out_channels=4 * (reg_max + num_anchors), kernel_size=1))
        layers.append(nn.Conv2d(in_channels=c, out_channels=8 * num_anchors, kernel_size=1))
    return nn.Sequential(*layers)
