import os, sys, torch
sys.path.insert(0, os.getcwd())
from yolov6.hip import abi
from yolov6.hip.runtime import Engine, _f32
eng = Engine(torch.float16, 'cuda:0'); eng.autotune = False
C = int(sys.argv[1]) if len(sys.argv) > 1 else 64
f = eng.tensor(C, 3)
g = torch.Generator().manual_seed(0)
wc, bc = torch.randn(277, C, generator=g) * 0.3, torch.randn(277, generator=g)
abi.check(eng.lib.lp_engine_add_head_cls(eng.h, f, 0, 277, eng._ptr(_f32(wc)), eng._ptr(_f32(bc))))
eng.finish(); eng.bind(32, 640, 640)
eng.tensor_view(f).copy_(torch.randn(32, C, 80, 80, generator=g).to('cuda:0', torch.float16))
x = torch.zeros(32, 3, 640, 640, device='cuda:0', dtype=torch.float16)
for v in (2, 18):
    eng.set_variant(1, v, 1)
    o = eng.profile(x, reps=30)[1]
    print('C=%d variant %s: %.1f us  %.2f TB/s' % (C, o['variant'], o['ms'] * 1e3, o['bytes'] / o['ms'] / 1e9), flush=True)
