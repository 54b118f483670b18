"""Every kernel variant of a conv layer must produce the same bits (they differ in tiling only).  Sweeps the layer shapes
of yololps at 640x640 (batch 1 and 2) over all variants that lp_engine_set_op_variant accepts, several runs each, and
reports any variant whose output differs from the first one or from its own previous run."""
import os, sys, itertools, torch
sys.path.insert(0, os.getcwd())
from yolov6.hip import abi
from yolov6.hip.runtime import Engine

def one(cins, cout, k, s, hw, B, act, dtype=torch.float16):
    eng = Engine(dtype, 'cuda:0'); eng.autotune = False
    sl = 5
    srcs = [eng.tensor(c, sl) for c in cins]
    g = torch.Generator().manual_seed(0)
    cin = sum(cins)
    w = torch.randn(cout, cin, k, k, generator=g) * (2.0 / (cin * k * k)) ** 0.5
    dst = eng.conv(srcs, w, torch.randn(cout, generator=g) * 0.3, k, s, act, sl)
    eng.finish(); eng.bind(B, hw << sl, hw << sl)
    for t, c in zip(srcs, cins):
        eng.tensor_view(t).copy_(torch.randn(B, c, hw, hw, generator=g).to('cuda:0', dtype))
    x = torch.zeros(B, 3, hw << sl, hw << sl, device='cuda:0', dtype=dtype)
    op = eng.lib.lp_engine_num_ops(eng.h) - 1
    base, bad = None, []
    for cfg, nb in itertools.product(list(range(8)) + [16, 17], (1, 2)):
        try:
            eng.set_variant(op, cfg, nb)
        except RuntimeError:
            continue
        for rep in range(3):
            eng.tensor_view(dst).zero_()
            eng.forward(x)
            out = eng.tensor_view(dst).clone()
            if base is None:
                base = out
            elif not torch.equal(out, base):
                bad.append((cfg, nb, rep, float((out.float() - base.float()).abs().max())))
    return bad

SHAPES = [([64], 64, 3, 1, 160), ([64], 128, 3, 2, 160), ([128], 128, 3, 1, 80), ([128], 256, 3, 2, 80), ([256], 256, 3, 1, 40),
          ([256], 512, 3, 2, 40), ([512], 512, 3, 1, 20), ([64], 64, 1, 1, 160), ([128], 64, 1, 1, 80), ([128, 64], 64, 1, 1, 160),
          ([128], 128, 1, 1, 80), ([256], 128, 1, 1, 80), ([128, 128, 128], 128, 1, 1, 80), ([512], 256, 1, 1, 20),
          ([256, 256, 256, 256], 256, 1, 1, 20), ([64], 64, 3, 1, 80), ([128], 128, 3, 1, 40), ([256], 256, 3, 1, 20),
          ([64, 64], 64, 3, 1, 80), ([32], 64, 3, 2, 320)]
total = 0
for B in (1, 2):
    for cins, cout, k, s, hw in SHAPES:
        for act in (abi.LP_ACT_RELU, abi.LP_ACT_SILU):
            bad = one(cins, cout, k, s, hw, B, act)
            total += len(bad)
            if bad:
                print('B%d %s->%d k%d s%d @%d act%d: DIFFERENT' % (B, cins, cout, k, s, hw, act), bad[:6], flush=True)
print('done, mismatching (variant, run) pairs:', total)
