"""Backbones of the YOLO-LP hot path (host-side mirror).

``EfficientRep`` (yololps / yololpn) and ``CSPBepBackbone`` (yolov6m) with the
reference's constructor signatures, attribute names and module creation order
(reference yolov6/models/efficientrep.py:6-117 and :249-364).  The P6
variants are outside the hot-path scope.
"""
from torch import nn

from yolov6.layers.common import (BottleRep, RepVGGBlock, RepBlock, BepC3, SimSPPF, SPPF, SimCSPSPPF, CSPSPPF,
                                  ConvWrapper)


def _merge_layer(block, cspsppf):
    """Channel-merge layer closing stage 5: SiLU flavours only for conv_silu."""
    silu = block == ConvWrapper
    if cspsppf:
        return CSPSPPF if silu else SimCSPSPPF
    return SPPF if silu else SimSPPF


class _StagedBackbone(nn.Module):
    """stem (s2) + four stages, each a stride-2 block followed by a body."""

    def _body(self, channels, n, block):
        raise NotImplementedError

    def _build(self, in_channels, channels_list, num_repeats, block, fuse_P2, cspsppf):
        assert channels_list is not None
        assert num_repeats is not None
        self.fuse_P2 = fuse_P2
        c = channels_list
        self.stem = block(in_channels=in_channels, out_channels=c[0], kernel_size=3, stride=2)
        for i in (1, 2, 3, 4):
            layers = [block(in_channels=c[i - 1], out_channels=c[i], kernel_size=3, stride=2),
                      self._body(c[i], num_repeats[i], block)]
            if i == 4:
                layers.append(_merge_layer(block, cspsppf)(in_channels=c[4], out_channels=c[4], kernel_size=5))
            setattr(self, 'ERBlock_%d' % (i + 1), nn.Sequential(*layers))

    def forward(self, x):
        x = self.ERBlock_2(self.stem(x))
        outputs = [x] if self.fuse_P2 else []
        for stage in (self.ERBlock_3, self.ERBlock_4, self.ERBlock_5):
            x = stage(x)
            outputs.append(x)
        return tuple(outputs)


class EfficientRep(_StagedBackbone):
    """Rep-style backbone; returns (P2,) P3, P4, P5."""

    def __init__(self, in_channels=3, channels_list=None, num_repeats=None, block=RepVGGBlock, fuse_P2=False,
                 cspsppf=False):
        super().__init__()
        self._build(in_channels, channels_list, num_repeats, block, fuse_P2, cspsppf)

    def _body(self, channels, n, block):
        return RepBlock(in_channels=channels, out_channels=channels, n=n, block=block)


class CSPBepBackbone(_StagedBackbone):
    """CSP backbone whose stage bodies are ``BepC3`` blocks."""

    def __init__(self, in_channels=3, channels_list=None, num_repeats=None, block=RepVGGBlock,
                 csp_e=float(1) / 2, fuse_P2=False, cspsppf=False):
        super().__init__()
        self._csp_e = csp_e
        self._build(in_channels, channels_list, num_repeats, block, fuse_P2, cspsppf)

    def _body(self, channels, n, block):
        return BepC3(in_channels=channels, out_channels=channels, n=n, e=self._csp_e, block=block)
