"""Repeat one pipelined-kernel case many times and report where an output differs from the generic kernel's (race hunting).
    python tools/micro/pipe_stress.py [cin cout h w B variant reps]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from yolov6.hip import abi
from yolov6.hip.runtime import Engine
cin, cout, h, w, B, variant, reps = [int(v) for v in (sys.argv[1:] + ['12', '32', '96', '160', '3', '35', '300'][len(sys.argv) - 1:])]
dt = torch.float16
g = torch.Generator().manual_seed(1)
eng = Engine(dt, 'cuda:0')
eng.autotune = False
sl = 5
src = eng.tensor(cin, sl)
wt = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (cin * 9)) ** 0.5
dst = eng.conv([src], wt, torch.randn(cout, generator=g) * 0.5, 3, 1, abi.LP_ACT_RELU, sl)
eng.finish()
H, W = h << sl, w << sl
eng.bind(B, H, W)
eng.tensor_view(src).copy_(torch.randn(B, cin, h, w, generator=g).to('cuda:0', dt))
x = torch.zeros(B, 3, H, W, device='cuda:0')
op = eng.lib.lp_engine_num_ops(eng.h) - 1
eng.forward(x)
base = eng.tensor_view(dst).clone()
eng.set_variant(op, variant, 3)
bad = 0
junk = torch.empty(64 << 20, device='cuda:0')
for rep in range(reps):
    eng.tensor_view(dst).fill_(float('nan'))
    if rep % 3 == 1:
        junk.normal_()                       # other traffic in front of the launch
    eng.forward(x)
    out = eng.tensor_view(dst)
    if not torch.equal(out, base):
        bad += 1
        d = (out.float() - base.float())
        idx = d.isnan() | (d != 0)
        nz = idx.nonzero()
        print('rep %d: %d elements differ; images %s, channels %s..%s, rows %s..%s, cols %s..%s; nan %d' % (
            rep, len(nz), nz[:, 0].unique().tolist(), int(nz[:, 1].min()), int(nz[:, 1].max()), int(nz[:, 2].min()), int(nz[:, 2].max()),
            int(nz[:, 3].min()), int(nz[:, 3].max()), int(d.isnan().sum())))
print('%d of %d runs differ' % (bad, reps))
