"""Seeded synthetic weights for benchmarks and parity tests.

A freshly constructed model is degenerate: ``Detect.initialize_biases`` zeroes
every predictor weight, so outputs do not depend on the input.
There are no pretrained weights offline, so benches and tests use this recipe
(SURVEY.md §8(c)): seeded construction, then random BN statistics and random
predictor weights from a second generator.  ``tests/golden/make_golden.py``
applies the same recipe to the *reference's* model classes, which is what makes
the committed golden outputs comparable.
"""
import torch
import torch.nn as nn


def randomize(model, seed=1, sigma=0.35, bias_sigma=0.5):
    """BN: mean~N(0,.1^2), var~U(.5,1.5), gamma~U(.5,1.5), beta~N(0,.1^2);
    every ``detect.*_preds.*`` conv: weight~N(0,sigma^2), bias += N(0,bias_sigma^2).
    Module iteration order (``named_modules``) fixes the draw order."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, m in model.named_modules():
            if isinstance(m, nn.BatchNorm2d):
                n = m.num_features
                m.running_mean.copy_(torch.randn(n, generator=g) * 0.1)
                m.running_var.copy_(torch.rand(n, generator=g) + 0.5)
                m.weight.copy_(torch.rand(n, generator=g) + 0.5)
                m.bias.copy_(torch.randn(n, generator=g) * 0.1)
            elif isinstance(m, nn.Conv2d) and name.startswith('detect.') and '_preds.' in name:
                m.weight.copy_(torch.randn(m.weight.shape, generator=g) * sigma)
                m.bias.add_(torch.randn(m.bias.shape, generator=g) * bias_sigma)
    return model


def build_synthetic(cfg_path, seed=0, rand_seed=1, sigma=0.35, width=None, npro=31, nalp=24, nads=37, depth=None, neck=None,
                    fuse_P2=None):
    """Seeded model of a config file (optionally with another width / depth multiple, neck type or fuse_P2 flag),
    randomised, eval mode."""
    from yolov6.utils.config import Config
    from yolov6.models.yolo import build_model
    cfg = Config.fromfile(cfg_path)
    if not hasattr(cfg, 'training_mode'):
        cfg.training_mode = 'repvgg'
    if width is not None:
        cfg.model.width_multiple = width
    if depth is not None:
        cfg.model.depth_multiple = depth
    if neck is not None:
        cfg.model.neck.type = neck
    if fuse_P2 is not None:
        cfg.model.backbone.fuse_P2 = fuse_P2
    torch.manual_seed(seed)
    model = build_model(cfg, npro, nalp, nads, 'cpu')
    return randomize(model, rand_seed, sigma).eval()
