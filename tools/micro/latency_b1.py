import os, sys, time, torch
ROOT = os.environ.get('GRAFT_REPO_ROOT', '/root/repo'); sys.path.insert(0, ROOT)
from yolov6.utils.synth import build_synthetic
from yolov6.utils.torch_utils import fuse_model
from yolov6.layers.common import RepVGGBlock
from yolov6.utils.nms import non_max_suppression
m = build_synthetic(os.path.join(ROOT, 'configs/yololps.py'), sigma=0.25)
m = fuse_model(m).eval()
for l in m.modules():
    if isinstance(l, RepVGGBlock): l.switch_to_deploy()
m = m.cuda().half()
import itertools
for graph, shape in itertools.product((False, True), ((1, 3, 640, 416), (1, 3, 640, 640), (8, 3, 640, 640))):
    m.lp_graph = graph
    x = torch.rand(*shape, device='cuda').half()
    with torch.no_grad():
        for _ in range(5):
            p, _ = m(x); non_max_suppression(p, 0.4, 0.45, max_det=1000)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(50):
            p, _ = m(x); d = non_max_suppression(p, 0.4, 0.45, max_det=1000)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
        t0 = time.perf_counter()
        for _ in range(50):
            p, _ = m(x)
        torch.cuda.synchronize(); df = (time.perf_counter() - t0) / 50
    print('graph=%s' % graph, '%s: model+NMS (with per-call host sync) %.3f ms, model only %.3f ms  -> %.0f img/s' % (shape, dt * 1e3, df * 1e3, shape[0] / dt))
