"""Backbones of the YOLO-LP hot path (host-side mirror).

``EfficientRep`` (yololps / yololpn), ``CSPBepBackbone`` (yolov6m) and their P6
variants ``EfficientRep6`` / ``CSPBepBackbone_P6`` (a sixth stage at stride 64) with
the reference's constructor signatures, attribute names and module creation order
(reference yolov6/models/efficientrep.py:6-117, :120-246, :249-364, :367-497).
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
    """stem (s2) + four (P5) or five (P6) stages, each a stride-2 block followed by a body; the last stage ends with
    the channel-merge (SPPF) layer."""

    last_stage = 5          # index of the last ERBlock
    merge_checks_block = True   # EfficientRep6 picks Sim(CSP)SPPF whatever the block (efficientrep.py:213)

    def _body(self, channels, n, block):
        raise NotImplementedError

    def _build(self, in_channels, channels_list, num_repeats, block, fuse_P2, cspsppf):
        assert channels_list is not None
        assert num_repeats is not None
        self.fuse_P2 = fuse_P2
        c = channels_list
        self.stem = block(in_channels=in_channels, out_channels=c[0], kernel_size=3, stride=2)
        for i in range(1, self.last_stage):
            layers = [block(in_channels=c[i - 1], out_channels=c[i], kernel_size=3, stride=2),
                      self._body(c[i], num_repeats[i], block)]
            if i == self.last_stage - 1:
                merge = _merge_layer(block if self.merge_checks_block else None, cspsppf)
                layers.append(merge(in_channels=c[i], out_channels=c[i], kernel_size=5))
            setattr(self, 'ERBlock_%d' % (i + 1), nn.Sequential(*layers))

    def forward(self, x):
        x = self.ERBlock_2(self.stem(x))
        outputs = [x] if self.fuse_P2 else []
        for i in range(3, self.last_stage + 1):
            x = getattr(self, 'ERBlock_%d' % i)(x)
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


class EfficientRep6(EfficientRep):
    """EfficientRep + P6: returns (P2,) P3, P4, P5, P6."""
    last_stage = 6
    merge_checks_block = False


class CSPBepBackbone_P6(CSPBepBackbone):
    """CSPBepBackbone + P6.  Like the reference's forward (efficientrep.py:481-497) it returns P2 whatever ``fuse_P2``
    says, so it only pairs with the BiFusion neck."""
    last_stage = 6

    def forward(self, x):
        x = self.ERBlock_2(self.stem(x))
        outputs = [x]
        for i in range(3, 7):
            x = getattr(self, 'ERBlock_%d' % i)(x)
            outputs.append(x)
        return tuple(outputs)
