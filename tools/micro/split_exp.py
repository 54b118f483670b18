import os, sys, time, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from yolov6.utils.synth import build_synthetic
from yolov6.utils.torch_utils import fuse_model
from yolov6.layers.common import RepVGGBlock
from yolov6.hip import runtime
ROOT = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
m = build_synthetic(os.path.join(ROOT, 'configs/yololps.py'), sigma=0.25)
m = fuse_model(m).eval()
for l in m.modules():
    if isinstance(l, RepVGGBlock): l.switch_to_deploy()
m = m.cuda().half()
x = torch.rand(32, 3, 640, 640, device='cuda').half()
for nsplit in (1, 2, 4):
    engs = [runtime.Engine.from_model(m, torch.float16, 'cuda:0') for _ in range(nsplit)]
    streams = [torch.cuda.Stream() for _ in range(nsplit)]
    xs = list(x.chunk(nsplit))
    def step():
        for e, s, xi in zip(engs, streams, xs):
            with torch.cuda.stream(s):
                p = e.forward(xi)
                runtime.nms_padded(p, 0.4, 0.45, 1000)
    for _ in range(5): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30): step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print('split into %d concurrent sub-batches: %.0f img/s (%.3f ms/step)' % (nsplit, 32 * 30 / dt, dt / 30 * 1e3))
