"""Necks of the YOLO-LP hot path (host-side mirror).

All eight necks of the reference (yolov6/models/reppan.py): the BiFusion ones used by the LP configs
(``RepBiFPANNeck`` :131-236, ``CSPRepBiFPANNeck`` :657-768), their P6 variants (``RepBiFPANNeck6`` :393-541,
``CSPRepBiFPANNeck_P6`` :930-1083) and the plain PAN ones whose top-down path is transpose-conv + concat
(``RepPANNeck`` :6-128, ``RepPANNeck6`` :239-390, ``CSPRepPANNeck`` :543-655, ``CSPRepPANNeck_P6`` :771-928).
Same attribute names, module creation order (it fixes the RNG stream of the initialisers) and dataflow.

channels_list indices: P5 necks [0..4] backbone, [5..10] neck; P6 necks [0..5] backbone, [6..11] neck.
"""
import torch
from torch import nn

from yolov6.layers.common import RepVGGBlock, RepBlock, BepC3, BottleRep, SimConv, BiFusion, Transpose


class _Neck(nn.Module):
    """Stage factory shared by the Rep (RepBlock) and CSP (BepC3) flavours."""
    csp = False

    def _stage(self, cin, cout, n, block):
        if self.csp:
            return BepC3(in_channels=cin, out_channels=cout, n=n, e=self._csp_e, block=block)
        return RepBlock(in_channels=cin, out_channels=cout, n=n, block=block)

    def _init(self, channels_list, num_repeats, block, csp_e=None):
        assert channels_list is not None
        assert num_repeats is not None
        self._csp_e = csp_e
        self._build(channels_list, num_repeats, block)


def _reduce(cin, cout):
    return SimConv(in_channels=cin, out_channels=cout, kernel_size=1, stride=1)


def _down(cin, cout):
    return SimConv(in_channels=cin, out_channels=cout, kernel_size=3, stride=2)


class _BiFPANNeck(_Neck):
    """Top-down BiFusion path followed by the bottom-up PAN path (three output levels)."""

    def _build(self, c, r, block):
        self.reduce_layer0 = _reduce(c[4], c[5])
        self.Bifusion0 = BiFusion(in_channels=[c[3], c[5]], out_channels=c[5])
        self.Rep_p4 = self._stage(c[5], c[5], r[5], block)
        self.reduce_layer1 = _reduce(c[5], c[6])
        self.Bifusion1 = BiFusion(in_channels=[c[5], c[6]], out_channels=c[6])
        self.Rep_p3 = self._stage(c[6], c[6], r[6], block)
        self.downsample2 = _down(c[6], c[7])
        self.Rep_n3 = self._stage(c[6] + c[7], c[8], r[7], block)
        self.downsample1 = _down(c[8], c[9])
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


class _BiFPANNeck6(_Neck):
    """BiFusion top-down path over four backbone levels + P2, bottom-up PAN path (four output levels)."""

    def _build(self, c, r, block):
        self.reduce_layer0 = _reduce(c[5], c[6])
        self.Bifusion0 = BiFusion(in_channels=[c[4], c[6]], out_channels=c[6])
        self.Rep_p5 = self._stage(c[6], c[6], r[6], block)
        self.reduce_layer1 = _reduce(c[6], c[7])
        self.Bifusion1 = BiFusion(in_channels=[c[3], c[7]], out_channels=c[7])
        self.Rep_p4 = self._stage(c[7], c[7], r[7], block)
        self.reduce_layer2 = _reduce(c[7], c[8])
        self.Bifusion2 = BiFusion(in_channels=[c[2], c[8]], out_channels=c[8])
        self.Rep_p3 = self._stage(c[8], c[8], r[8], block)
        self.downsample2 = _down(c[8], c[8])
        self.Rep_n4 = self._stage(c[8] + c[8], c[9], r[9], block)
        self.downsample1 = _down(c[9], c[9])
        self.Rep_n5 = self._stage(c[7] + c[9], c[10], r[10], block)
        self.downsample0 = _down(c[10], c[10])
        self.Rep_n6 = self._stage(c[6] + c[10], c[11], r[11], block)

    def forward(self, input):
        (x4, x3, x2, x1, x0) = input
        fpn_out0 = self.reduce_layer0(x0)
        f_out0 = self.Rep_p5(self.Bifusion0([fpn_out0, x1, x2]))
        fpn_out1 = self.reduce_layer1(f_out0)
        f_out1 = self.Rep_p4(self.Bifusion1([fpn_out1, x2, x3]))
        fpn_out2 = self.reduce_layer2(f_out1)
        pan_out3 = self.Rep_p3(self.Bifusion2([fpn_out2, x3, x4]))
        pan_out2 = self.Rep_n4(torch.cat([self.downsample2(pan_out3), fpn_out2], 1))
        pan_out1 = self.Rep_n5(torch.cat([self.downsample1(pan_out2), fpn_out1], 1))
        pan_out0 = self.Rep_n6(torch.cat([self.downsample0(pan_out1), fpn_out0], 1))
        return [pan_out3, pan_out2, pan_out1, pan_out0]


class _PANNeck(_Neck):
    """Plain PAN: transpose-conv upsampling + concat on the way down (three output levels).  The reference builds the
    four stages first and the connecting layers afterwards (reppan.py:21-118, 556-648)."""

    def _build(self, c, r, block):
        self.Rep_p4 = self._stage(c[3] + c[5], c[5], r[5], block)
        self.Rep_p3 = self._stage(c[2] + c[6], c[6], r[6], block)
        self.Rep_n3 = self._stage(c[6] + c[7], c[8], r[7], block)
        self.Rep_n4 = self._stage(c[5] + c[9], c[10], r[8], block)
        self.reduce_layer0 = _reduce(c[4], c[5])
        self.upsample0 = Transpose(in_channels=c[5], out_channels=c[5])
        self.reduce_layer1 = _reduce(c[5], c[6])
        self.upsample1 = Transpose(in_channels=c[6], out_channels=c[6])
        self.downsample2 = _down(c[6], c[7])
        self.downsample1 = _down(c[8], c[9])

    def forward(self, input):
        (x2, x1, x0) = input
        fpn_out0 = self.reduce_layer0(x0)
        f_out0 = self.Rep_p4(torch.cat([self.upsample0(fpn_out0), x1], 1))
        fpn_out1 = self.reduce_layer1(f_out0)
        pan_out2 = self.Rep_p3(torch.cat([self.upsample1(fpn_out1), x2], 1))
        pan_out1 = self.Rep_n3(torch.cat([self.downsample2(pan_out2), fpn_out1], 1))
        pan_out0 = self.Rep_n4(torch.cat([self.downsample1(pan_out1), fpn_out0], 1))
        return [pan_out2, pan_out1, pan_out0]


class _PANNeck6(_Neck):
    """Plain PAN over four backbone levels (four output levels), layers created in dataflow order."""

    def _build(self, c, r, block):
        self.reduce_layer0 = _reduce(c[5], c[6])
        self.upsample0 = Transpose(in_channels=c[6], out_channels=c[6])
        self.Rep_p5 = self._stage(c[4] + c[6], c[6], r[6], block)
        self.reduce_layer1 = _reduce(c[6], c[7])
        self.upsample1 = Transpose(in_channels=c[7], out_channels=c[7])
        self.Rep_p4 = self._stage(c[3] + c[7], c[7], r[7], block)
        self.reduce_layer2 = _reduce(c[7], c[8])
        self.upsample2 = Transpose(in_channels=c[8], out_channels=c[8])
        self.Rep_p3 = self._stage(c[2] + c[8], c[8], r[8], block)
        self.downsample2 = _down(c[8], c[8])
        self.Rep_n4 = self._stage(c[8] + c[8], c[9], r[9], block)
        self.downsample1 = _down(c[9], c[9])
        self.Rep_n5 = self._stage(c[7] + c[9], c[10], r[10], block)
        self.downsample0 = _down(c[10], c[10])
        self.Rep_n6 = self._stage(c[6] + c[10], c[11], r[11], block)

    def forward(self, input):
        (x3, x2, x1, x0) = input
        fpn_out0 = self.reduce_layer0(x0)
        f_out0 = self.Rep_p5(torch.cat([self.upsample0(fpn_out0), x1], 1))
        fpn_out1 = self.reduce_layer1(f_out0)
        f_out1 = self.Rep_p4(torch.cat([self.upsample1(fpn_out1), x2], 1))
        fpn_out2 = self.reduce_layer2(f_out1)
        pan_out3 = self.Rep_p3(torch.cat([self.upsample2(fpn_out2), x3], 1))
        pan_out2 = self.Rep_n4(torch.cat([self.downsample2(pan_out3), fpn_out2], 1))
        pan_out1 = self.Rep_n5(torch.cat([self.downsample1(pan_out2), fpn_out1], 1))
        pan_out0 = self.Rep_n6(torch.cat([self.downsample0(pan_out1), fpn_out0], 1))
        return [pan_out3, pan_out2, pan_out1, pan_out0]


class _RepInit:
    def __init__(self, channels_list=None, num_repeats=None, block=RepVGGBlock):
        super().__init__()
        self._init(channels_list, num_repeats, block)


class _CSPInit:
    csp = True

    def __init__(self, channels_list=None, num_repeats=None, block=BottleRep, csp_e=float(1) / 2):
        super().__init__()
        self._init(channels_list, num_repeats, block, csp_e)


class RepPANNeck(_RepInit, _PANNeck):
    """reference reppan.py:6-128"""


class RepBiFPANNeck(_RepInit, _BiFPANNeck):
    """reference reppan.py:131-236 (yololps / yololpn)"""


class RepPANNeck6(_RepInit, _PANNeck6):
    """reference reppan.py:239-390"""


class RepBiFPANNeck6(_RepInit, _BiFPANNeck6):
    """reference reppan.py:393-541"""


class CSPRepPANNeck(_CSPInit, _PANNeck):
    """reference reppan.py:543-655"""


class CSPRepBiFPANNeck(_CSPInit, _BiFPANNeck):
    """reference reppan.py:657-768 (yolov6m)"""


class CSPRepPANNeck_P6(_CSPInit, _PANNeck6):
    """reference reppan.py:771-928"""


class CSPRepBiFPANNeck_P6(_CSPInit, _BiFPANNeck6):
    """reference reppan.py:930-1083"""
