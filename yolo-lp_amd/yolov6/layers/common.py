"""Layer ops of the YOLO-LP detection hot path (host-side mirror).

Mirrors the public names, constructor signatures, attribute / state_dict key
names and parameter-creation order of the reference's
``yolov6/layers/common.py`` so that seeded construction
reproduces the same weights and reference checkpoints (whole pickled modules)
resolve against these classes.

The ``forward`` methods here are plain torch ops: they are the CPU plumbing
path (BASELINE configs[0]) and what training-mode modules run.  On a GPU the
eval-mode ``Model.forward`` does not go through them: it is routed to the HIP
engine (``yolov6.hip``), which consumes the folded weights of these modules.
"""
import warnings
from pathlib import Path

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn.parameter import Parameter


class SiLU(nn.Module):
    """x * sigmoid(x)  (reference common.py:14-18)."""

    @staticmethod
    def forward(x):
        return x * torch.sigmoid(x)


class _ConvBNAct(nn.Module):
    """conv -> bn -> act block; after ``fuse_model`` the bn is folded into the
    conv and ``forward`` is rebound to ``forward_fuse`` (reference
    common.py:21-66, torch_utils.py:85-94)."""

    _act = nn.Identity

    def __init__(self, in_channels, out_channels, kernel_size, stride, groups=1, bias=False):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size=kernel_size, stride=stride,
                              padding=kernel_size // 2, groups=groups, bias=bias)
        self.bn = nn.BatchNorm2d(out_channels)
        self.act = self._act()

    def forward(self, x):
        return self.act(self.bn(self.conv(x)))

    def forward_fuse(self, x):
        return self.act(self.conv(x))


class Conv(_ConvBNAct):
    """Conv + BN + SiLU (reference common.py:21-42)."""
    _act = nn.SiLU


class SimConv(_ConvBNAct):
    """Conv + BN + ReLU (reference common.py:45-66)."""
    _act = nn.ReLU


class ConvWrapper(nn.Module):
    """3x3 ``Conv`` with the RepVGGBlock call signature (reference common.py:68-75)."""

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, groups=1, bias=True):
        super().__init__()
        self.block = Conv(in_channels, out_channels, kernel_size, stride, groups, bias)

    def forward(self, x):
        return self.block(x)


class SimConvWrapper(nn.Module):
    """3x3 ``SimConv`` with the RepVGGBlock call signature (reference common.py:78-85)."""

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, groups=1, bias=True):
        super().__init__()
        self.block = SimConv(in_channels, out_channels, kernel_size, stride, groups, bias)

    def forward(self, x):
        return self.block(x)


def _pool_chain(pool, x):
    """x, m(x), m(m(x)), m(m(m(x))): 1/5/9/13 windows of the SPPF trick."""
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        y1 = pool(x)
        y2 = pool(y1)
        y3 = pool(y2)
    return [x, y1, y2, y3]


class _SPPFBase(nn.Module):
    """cv1 -> chained 5x5 max pools -> cv2 over the 4-way concat
    (reference common.py:88-121)."""

    _conv = SimConv

    def __init__(self, in_channels, out_channels, kernel_size=5):
        super().__init__()
        c_ = in_channels // 2
        self.cv1 = self._conv(in_channels, c_, 1, 1)
        self.cv2 = self._conv(c_ * 4, out_channels, 1, 1)
        self.m = nn.MaxPool2d(kernel_size=kernel_size, stride=1, padding=kernel_size // 2)

    def forward(self, x):
        return self.cv2(torch.cat(_pool_chain(self.m, self.cv1(x)), 1))


class SimSPPF(_SPPFBase):
    _conv = SimConv


class SPPF(_SPPFBase):
    _conv = Conv


class _CSPSPPFBase(nn.Module):
    """CSP-style SPPF (reference common.py:124-172): a 1x1/3x3/1x1 trunk, the
    pool chain, 1x1 + 3x3 on the 4-way concat, joined with a 1x1 shortcut."""

    _conv = SimConv

    def __init__(self, in_channels, out_channels, kernel_size=5, e=0.5):
        super().__init__()
        c_ = int(out_channels * e)
        mk = self._conv
        self.cv1 = mk(in_channels, c_, 1, 1)
        self.cv2 = mk(in_channels, c_, 1, 1)
        self.cv3 = mk(c_, c_, 3, 1)
        self.cv4 = mk(c_, c_, 1, 1)
        self.m = nn.MaxPool2d(kernel_size=kernel_size, stride=1, padding=kernel_size // 2)
        self.cv5 = mk(4 * c_, c_, 1, 1)
        self.cv6 = mk(c_, c_, 3, 1)
        self.cv7 = mk(2 * c_, out_channels, 1, 1)

    def forward(self, x):
        x1 = self.cv4(self.cv3(self.cv1(x)))
        y0 = self.cv2(x)
        y3 = self.cv6(self.cv5(torch.cat(_pool_chain(self.m, x1), 1)))
        return self.cv7(torch.cat((y0, y3), dim=1))


class SimCSPSPPF(_CSPSPPFBase):
    _conv = SimConv


class CSPSPPF(_CSPSPPFBase):
    _conv = Conv


class Transpose(nn.Module):
    """2x2 stride-2 transposed conv (with bias) used for upsampling
    (reference common.py:174-187)."""

    def __init__(self, in_channels, out_channels, kernel_size=2, stride=2):
        super().__init__()
        self.upsample_transpose = nn.ConvTranspose2d(in_channels, out_channels, kernel_size=kernel_size,
                                                     stride=stride, bias=True)

    def forward(self, x):
        return self.upsample_transpose(x)


class Concat(nn.Module):
    def __init__(self, dimension=1):
        super().__init__()
        self.d = dimension

    def forward(self, x):
        return torch.cat(x, self.d)


def conv_bn(in_channels, out_channels, kernel_size, stride, padding, groups=1):
    """Bias-free conv followed by BN, as one branch of a rep-style block."""
    seq = nn.Sequential()
    seq.add_module('conv', nn.Conv2d(in_channels, out_channels, kernel_size=kernel_size, stride=stride,
                                     padding=padding, groups=groups, bias=False))
    seq.add_module('bn', nn.BatchNorm2d(out_channels))
    return seq


def _bn_scale_shift(bn):
    """(gamma/sigma, beta - mu*gamma/sigma) of an eval-mode BatchNorm."""
    std = (bn.running_var + bn.eps).sqrt()
    return bn.weight / std, bn.bias - bn.running_mean * bn.weight / std


class RepVGGBlock(nn.Module):
    """Three-branch training form (3x3+BN, 1x1+BN, identity BN) that collapses
    to one 3x3 conv + bias for deployment; ReLU after the sum
    (reference common.py:208-325)."""

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, padding=1, dilation=1, groups=1,
                 padding_mode='zeros', deploy=False, use_se=False):
        super().__init__()
        assert kernel_size == 3
        assert padding == 1
        if use_se:
            raise NotImplementedError("se block not supported yet")
        self.deploy = deploy
        self.groups = groups
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.nonlinearity = nn.ReLU()
        self.se = nn.Identity()
        if deploy:
            self.rbr_reparam = nn.Conv2d(in_channels, out_channels, kernel_size=kernel_size, stride=stride,
                                         padding=padding, dilation=dilation, groups=groups, bias=True,
                                         padding_mode=padding_mode)
        else:
            has_id = out_channels == in_channels and stride == 1
            self.rbr_identity = nn.BatchNorm2d(in_channels) if has_id else None
            self.rbr_dense = conv_bn(in_channels, out_channels, kernel_size, stride, padding, groups)
            self.rbr_1x1 = conv_bn(in_channels, out_channels, 1, stride, padding - kernel_size // 2, groups)

    def forward(self, inputs):
        if hasattr(self, 'rbr_reparam'):
            return self.nonlinearity(self.se(self.rbr_reparam(inputs)))
        id_out = 0 if self.rbr_identity is None else self.rbr_identity(inputs)
        return self.nonlinearity(self.se(self.rbr_dense(inputs) + self.rbr_1x1(inputs) + id_out))

    # -- re-parameterisation (reference common.py:268-306) ------------------
    def _fuse_bn_tensor(self, branch):
        if branch is None:
            return 0, 0
        if isinstance(branch, nn.Sequential):
            kernel, bn = branch.conv.weight, branch.bn
        else:
            assert isinstance(branch, nn.BatchNorm2d)
            if not hasattr(self, 'id_tensor'):
                per_group = self.in_channels // self.groups
                eye = np.zeros((self.in_channels, per_group, 3, 3), dtype=np.float32)
                eye[np.arange(self.in_channels), np.arange(self.in_channels) % per_group, 1, 1] = 1
                self.id_tensor = torch.from_numpy(eye).to(branch.weight.device)
            kernel, bn = self.id_tensor, branch
        std = (bn.running_var + bn.eps).sqrt()
        t = (bn.weight / std).reshape(-1, 1, 1, 1)
        return kernel * t, bn.bias - bn.running_mean * bn.weight / std

    def _pad_1x1_to_3x3_tensor(self, kernel1x1):
        return 0 if kernel1x1 is None else F.pad(kernel1x1, [1, 1, 1, 1])

    def get_equivalent_kernel_bias(self):
        k3, b3 = self._fuse_bn_tensor(self.rbr_dense)
        k1, b1 = self._fuse_bn_tensor(self.rbr_1x1)
        kid, bid = self._fuse_bn_tensor(self.rbr_identity)
        return k3 + self._pad_1x1_to_3x3_tensor(k1) + kid, b3 + b1 + bid

    def switch_to_deploy(self):
        if hasattr(self, 'rbr_reparam'):
            return
        kernel, bias = self.get_equivalent_kernel_bias()
        src = self.rbr_dense.conv
        self.rbr_reparam = nn.Conv2d(src.in_channels, src.out_channels, kernel_size=src.kernel_size,
                                     stride=src.stride, padding=src.padding, dilation=src.dilation,
                                     groups=src.groups, bias=True)
        self.rbr_reparam.weight.data = kernel
        self.rbr_reparam.bias.data = bias
        for p in self.parameters():
            p.detach_()
        for name in ('rbr_dense', 'rbr_1x1', 'rbr_identity', 'id_tensor'):
            if hasattr(self, name):
                self.__delattr__(name)
        self.deploy = True


from yolov6.layers.train_blocks import RealVGGBlock, ScaleLayer, LinearAddBlock  # noqa: E402,F401  (re-exported)


class DetectBackend(nn.Module):
    """Loads a ``.pt`` checkpoint and exposes ``forward(im) -> pred``
    (reference common.py:399-413)."""

    def __init__(self, weights='yolov6s.pt', device=None, dnn=True):
        super().__init__()
        assert isinstance(weights, str) and Path(weights).suffix == '.pt', \
            f'{Path(weights).suffix} format is not supported.'
        from yolov6.utils.checkpoint import load_checkpoint
        model = load_checkpoint(weights, map_location=device)
        stride = int(model.stride.max())
        self.__dict__.update(locals())

    def forward(self, im, val=False):
        y, _ = self.model(im)
        if isinstance(y, np.ndarray):
            y = torch.tensor(y, device=self.device)
        return y


class BottleRep(nn.Module):
    """Two basic blocks with a (learnably) weighted residual
    (reference common.py:437-455)."""

    def __init__(self, in_channels, out_channels, basic_block=RepVGGBlock, weight=False):
        super().__init__()
        self.conv1 = basic_block(in_channels, out_channels)
        self.conv2 = basic_block(out_channels, out_channels)
        self.shortcut = in_channels == out_channels
        self.alpha = Parameter(torch.ones(1)) if weight else 1.0

    def forward(self, x):
        y = self.conv2(self.conv1(x))
        return y + self.alpha * x if self.shortcut else y


class RepBlock(nn.Module):
    """A stage of ``n`` rep-style blocks (reference common.py:416-434).

    When ``block`` is ``BottleRep`` the reference first builds the stage with
    un-weighted BottleReps and then rebuilds it with ``weight=True`` and
    ``n // 2`` blocks; the discarded modules still draw from the RNG, so the
    same double construction is replayed here to keep seeded weights equal.
    """

    def __init__(self, in_channels, out_channels, n=1, block=RepVGGBlock, basic_block=RepVGGBlock):
        super().__init__()
        self.conv1 = block(in_channels, out_channels)
        self.block = nn.Sequential(*[block(out_channels, out_channels) for _ in range(n - 1)]) if n > 1 else None
        if block == BottleRep:
            self.conv1 = BottleRep(in_channels, out_channels, basic_block=basic_block, weight=True)
            n = n // 2
            self.block = nn.Sequential(*[BottleRep(out_channels, out_channels, basic_block=basic_block, weight=True)
                                         for _ in range(n - 1)]) if n > 1 else None

    def forward(self, x):
        x = self.conv1(x)
        return x if self.block is None else self.block(x)


def autopad(k, p=None):
    """'same' padding for kernel ``k``."""
    if p is None:
        p = k // 2 if isinstance(k, int) else [x // 2 for x in k]
    return p


class Conv_C3(nn.Module):
    """Conv + BN + act (ReLU by default) used inside BepC3
    (reference common.py:466-476)."""

    def __init__(self, c1, c2, k=1, s=1, p=None, g=1, act=True):
        super().__init__()
        self.conv = nn.Conv2d(c1, c2, k, s, autopad(k, p), groups=g, bias=False)
        self.bn = nn.BatchNorm2d(c2)
        self.act = nn.ReLU() if act is True else (act if isinstance(act, nn.Module) else nn.Identity())

    def forward(self, x):
        return self.act(self.bn(self.conv(x)))

    def forward_fuse(self, x):
        return self.act(self.conv(x))


class BepC3(nn.Module):
    """CSP block: cv3(cat[m(cv1 x), cv2 x]) with m a BottleRep stage
    (reference common.py:479-501)."""

    def __init__(self, in_channels, out_channels, n=1, e=0.5, concat=True, block=RepVGGBlock):
        super().__init__()
        c_ = int(out_channels * e)
        self.cv1 = Conv_C3(in_channels, c_, 1, 1)
        self.cv2 = Conv_C3(in_channels, c_, 1, 1)
        self.cv3 = Conv_C3(2 * c_, out_channels, 1, 1)
        if block == ConvWrapper:
            self.cv1 = Conv_C3(in_channels, c_, 1, 1, act=nn.SiLU())
            self.cv2 = Conv_C3(in_channels, c_, 1, 1, act=nn.SiLU())
            self.cv3 = Conv_C3(2 * c_, out_channels, 1, 1, act=nn.SiLU())
        self.m = RepBlock(in_channels=c_, out_channels=c_, n=n, block=BottleRep, basic_block=block)
        self.concat = concat
        if not concat:
            self.cv3 = Conv_C3(c_, out_channels, 1, 1)

    def forward(self, x):
        if self.concat is True:
            return self.cv3(torch.cat((self.m(self.cv1(x)), self.cv2(x)), dim=1))
        return self.cv3(self.m(self.cv1(x)))


class BiFusion(nn.Module):
    """cv3(cat[upsample(x0), cv1(x1), downsample(cv2(x2))])
    (reference common.py:504-527)."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.cv1 = SimConv(in_channels[0], out_channels, 1, 1)
        self.cv2 = SimConv(in_channels[1], out_channels, 1, 1)
        self.cv3 = SimConv(out_channels * 3, out_channels, 1, 1)
        self.upsample = Transpose(in_channels=out_channels, out_channels=out_channels)
        self.downsample = SimConv(in_channels=out_channels, out_channels=out_channels, kernel_size=3, stride=2)

    def forward(self, x):
        x0 = self.upsample(x[0])
        x1 = self.cv1(x[1])
        x2 = self.downsample(self.cv2(x[2]))
        return self.cv3(torch.cat((x0, x1, x2), dim=1))


_BLOCKS = {
    'repvgg': RepVGGBlock,
    'hyper_search': LinearAddBlock,
    'repopt': RealVGGBlock,
    'conv_relu': SimConvWrapper,
    'conv_silu': ConvWrapper,
}


def get_block(mode):
    """Basic block class for a ``training_mode`` (reference common.py:530-542)."""
    try:
        return _BLOCKS[mode]
    except KeyError:
        raise NotImplementedError("Undefied Repblock choice for mode {}".format(mode))
