"""Weight-preparation helpers of the inference path (host-side mirror of
reference yolov6/utils/torch_utils.py:31-94)."""
import time

import torch
import torch.nn as nn

from yolov6.utils.events import LOGGER  # noqa: F401


def time_sync():
    """Wall clock after draining the device queue."""
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    return time.time()


def initialize_weights(model):
    """BN eps=1e-3 / momentum=0.03 and in-place activations (reference :38-47).
    The eps value feeds every BN fold, so it is part of the numerics."""
    for m in model.modules():
        t = type(m)
        if t is nn.BatchNorm2d:
            m.eps = 1e-3
            m.momentum = 0.03
        elif t in (nn.Hardswish, nn.LeakyReLU, nn.ReLU, nn.ReLU6, nn.SiLU):
            m.inplace = True


def fuse_conv_and_bn(conv, bn):
    """Eval-mode BN folded into the conv in front of it (reference :50-82): with ``s = gamma / sqrt(var + eps)`` per output
    channel, ``W' = s * W`` and ``b' = s * b + beta - s * mu``.  The reference writes the scaling as products with
    ``diag(s)``; a diagonal matrix product adds exact zeros to one term per element, so the row scaling below gives the
    same bits."""
    dev = conv.weight.device
    s = bn.weight.div(torch.sqrt(bn.eps + bn.running_var))
    fused = nn.Conv2d(conv.in_channels, conv.out_channels, kernel_size=conv.kernel_size, stride=conv.stride,
                      padding=conv.padding, groups=conv.groups, bias=True).requires_grad_(False).to(dev)
    fused.weight.copy_((conv.weight.clone().view(conv.out_channels, -1) * s[:, None]).view(fused.weight.shape))
    b = conv.bias if conv.bias is not None else torch.zeros(conv.out_channels, device=dev)
    shift = bn.bias - bn.weight.mul(bn.running_mean).div(torch.sqrt(bn.running_var + bn.eps))
    fused.bias.copy_(b * s + shift)
    return fused


def fuse_model(model):
    """Fold the BN of every Conv / SimConv / Conv_C3 and rebind their forward."""
    from yolov6.layers.common import Conv, SimConv, Conv_C3
    for m in model.modules():
        if type(m) in (Conv, SimConv, Conv_C3) and hasattr(m, 'bn'):
            m.conv = fuse_conv_and_bn(m.conv, m.bn)
            delattr(m, 'bn')
            m.forward = m.forward_fuse
    from yolov6.hip import runtime
    runtime.drop_engine(model)
    return model


def get_model_info(model, img_size=640):
    """Params / GFLOPs line.  The reference uses ``thop`` (absent here); this
    counts 2*MAC of every conv / deconv with forward hooks on a 64x64 probe
    scaled to ``img_size`` (same protocol as reference :97-111)."""
    from copy import deepcopy
    stride = 64
    probe = deepcopy(model).cpu().float()
    macs = [0]

    def hook(mod, inp, out):
        kh, kw = mod.kernel_size
        per_out = mod.in_channels // mod.groups * kh * kw
        if isinstance(mod, nn.ConvTranspose2d):
            macs[0] += inp[0].numel() // inp[0].shape[1] * mod.in_channels * mod.out_channels * kh * kw // mod.groups
        else:
            macs[0] += out.numel() * per_out

    hs = [m.register_forward_hook(hook) for m in probe.modules() if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d))]
    with torch.no_grad():
        probe(torch.zeros((1, 3, stride, stride)))
    for h in hs:
        h.remove()
    params = sum(p.numel() for p in model.parameters()) / 1e6
    img_size = img_size if isinstance(img_size, list) else [img_size, img_size]
    flops = macs[0] / 1e9 * img_size[0] * img_size[1] / stride / stride * 2
    return "Params: {:.2f}M, Gflops: {:.2f}".format(params, flops)
