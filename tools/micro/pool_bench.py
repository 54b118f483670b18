"""Times the SPPF pool chain (256 channels, 20x20, batch 32 by default) through the engine."""
import os, sys, torch
sys.path.insert(0, os.getcwd())
from yolov6.hip import abi
from yolov6.hip.runtime import Engine
C, hw, B = 256, 20, 32
eng = Engine(torch.float16, 'cuda:0'); eng.autotune = False
src = eng.tensor(C, 5)
outs = eng.pools(src, 5, C)
eng.finish(); eng.bind(B, hw << 5, hw << 5)
eng.tensor_view(src).copy_(torch.randn(B, C, hw, hw).to('cuda:0', torch.float16))
x = torch.zeros(B, 3, hw << 5, hw << 5, device='cuda:0', dtype=torch.float16)
o = eng.profile(x, reps=50)[1]
print('pool %dch %dx%d B%d: %.1f us' % (C, hw, hw, B, o['ms'] * 1e3), flush=True)
