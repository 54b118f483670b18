"""Necks of the YOLO-LP hot path (host-side mirror).

``RepBiFPANNeck`` (yololps / yololpn) and ``CSPRepBiFPANNeck`` (yolov6m):
same attribute names, creation order and dataflow as the reference
(yolov6/models/reppan.py:131-236 and :657-768).  The other six necks of the
reference are unused by the BASELINE configs and out of scope.
"""
import torch
from torch import nn

from yolov6.layers.common import RepVGGBlock, RepBlock, BepC3, BottleRep, SimConv, BiFusion


class _BiFPANNeck(nn.Module):
    """Top-down BiFusion path followed by the bottom-up PAN path.

    channels_list indices: [0..4] backbone, [5..10] neck.
    """

    def _stage(self, cin, cout, n, block):
        raise NotImplementedError

    def _build(self, channels_list, num_repeats, block):
        assert channels_list is not None
        assert num_repeats is not None
        c, r = channels_list, num_repeats
        self.reduce_layer0 = SimConv(in_channels=c[4], out_channels=c[5], kernel_size=1, stride=1)
        self.Bifusion0 = BiFusion(in_channels=[c[3], c[5]], out_channels=c[5])
        self.Rep_p4 = self._stage(c[5], c[5], r[5], block)
        self.reduce_layer1 = SimConv(in_channels=c[5], out_channels=c[6], kernel_size=1, stride=1)
        self.Bifusion1 = BiFusion(in_channels=[c[5], c[6]], out_channels=c[6])
        self.Rep_p3 = self._stage(c[6], c[6], r[6], block)
        self.downsample2 = SimConv(in_channels=c[6], out_channels=c[7], kernel_size=3, stride=2)
        self.Rep_n3 = self._stage(c[6] + c[7], c[8], r[7], block)
        self.downsample1 = SimConv(in_channels=c[8], out_channels=c[9], kernel_size=3, stride=2)
        self.Rep_n4 = self._stage(c[5] + c[9], c[10], r[8], block)

    def forward(self, input):
        (x3, x2, x1, x0) = input
        fpn_out0 = self.reduce_layer0(x0)
        f_out0 = self.Rep_p4(self.Bifusion0([fpn_out0, x1, x2]))
        fpn_out1 = self.reduce_layer1(f_out0)
        pan_out2 = self.Rep_p3(self.Bifusion1([fpn_out1, x2, x3]))
        pan_out1 = self.Rep_n3(torch.cat([self.downsample2(pan_out2), fpn_out1], 1))
        pan_out0 = self.Rep_n4(torch.cat([self.downsample1(pan_out1), fpn_out0], 1))
        return [pan_out2, pan_out1, pan_out0]


class RepBiFPANNeck(_BiFPANNeck):
    def __init__(self, channels_list=None, num_repeats=None, block=RepVGGBlock):
        super().__init__()
        self._build(channels_list, num_repeats, block)

    def _stage(self, cin, cout, n, block):
        return RepBlock(in_channels=cin, out_channels=cout, n=n, block=block)


class CSPRepBiFPANNeck(_BiFPANNeck):
    def __init__(self, channels_list=None, num_repeats=None, block=BottleRep, csp_e=float(1) / 2):
        super().__init__()
        self._csp_e = csp_e
        self._build(channels_list, num_repeats, block)

    def _stage(self, cin, cout, n, block):
        return BepC3(in_channels=cin, out_channels=cout, n=n, e=self._csp_e, block=block)
