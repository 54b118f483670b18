"""Training-mode block variants of the rep-style backbone (host-side mirror of reference
yolov6/layers/common.py:328-396): the plain conv-BN-ReLU block of the RepOpt mode and the scaled-branch block of the
CSLA / hyper-search mode, with the reference's names, constructor signatures and parameter order.  They are selected by
``get_block`` for training modes other than ``repvgg`` and never reach the HIP engine (inference runs deploy-fused
``RepVGGBlock`` / ``Conv`` layers); kept so that configs and pickled modules of those modes resolve.  Re-exported by
``yolov6.layers.common``."""
import torch
import torch.nn as nn
from torch.nn.parameter import Parameter


class RealVGGBlock(nn.Module):
    """Plain conv + BN + ReLU block of the RepOpt training mode
    (reference common.py:328-345)."""

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, padding=1, dilation=1, groups=1,
                 padding_mode='zeros', use_se=False):
        super().__init__()
        if use_se:
            raise NotImplementedError("se block not supported yet")
        self.relu = nn.ReLU()
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size=kernel_size, stride=stride,
                              padding=padding, bias=False)
        self.bn = nn.BatchNorm2d(out_channels)
        self.se = nn.Identity()

    def forward(self, inputs):
        return self.relu(self.se(self.bn(self.conv(inputs))))


class ScaleLayer(nn.Module):
    """Per-channel scale (+ optional bias) (reference common.py:348-365)."""

    def __init__(self, num_features, use_bias=True, scale_init=1.0):
        super().__init__()
        self.num_features = num_features
        self.weight = Parameter(torch.full((num_features,), float(scale_init)))
        self.bias = Parameter(torch.zeros(num_features)) if use_bias else None

    def forward(self, inputs):
        out = inputs * self.weight.view(1, self.num_features, 1, 1)
        return out if self.bias is None else out + self.bias.view(1, self.num_features, 1, 1)


class LinearAddBlock(nn.Module):
    """CSLA / hyper-search block: scaled 3x3 + scaled 1x1 (+ scaled identity),
    then BN and ReLU (reference common.py:369-396)."""

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, padding=1, dilation=1, groups=1,
                 padding_mode='zeros', use_se=False, is_csla=False, conv_scale_init=1.0):
        super().__init__()
        if use_se:
            raise NotImplementedError("se block not supported yet")
        self.in_channels = in_channels
        self.relu = nn.ReLU()
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size=kernel_size, stride=stride,
                              padding=padding, bias=False)
        self.scale_conv = ScaleLayer(out_channels, use_bias=False, scale_init=conv_scale_init)
        self.conv_1x1 = nn.Conv2d(in_channels, out_channels, kernel_size=1, stride=stride, padding=0, bias=False)
        self.scale_1x1 = ScaleLayer(out_channels, use_bias=False, scale_init=conv_scale_init)
        if in_channels == out_channels and stride == 1:
            self.scale_identity = ScaleLayer(out_channels, use_bias=False, scale_init=1.0)
        self.bn = nn.BatchNorm2d(out_channels)
        if is_csla:
            self.scale_1x1.requires_grad_(False)
            self.scale_conv.requires_grad_(False)
        self.se = nn.Identity()

    def forward(self, inputs):
        out = self.scale_conv(self.conv(inputs)) + self.scale_1x1(self.conv_1x1(inputs))
        if hasattr(self, 'scale_identity'):
            out += self.scale_identity(inputs)
        return self.relu(self.se(self.bn(out)))
