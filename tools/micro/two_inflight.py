"""Experiment: several batches in flight (what became yolov6/core/pipeline.py).  Optional second argument `graph`: every
engine replays its forward as one hipGraph -- +3 % in this script, nothing in bench.py, so it is not in the product.  Steps i and i+1 run their forwards on two engines (two arenas) and two HIP streams,
so the kernels of consecutive steps interleave on the GPU; NMS of every step on a third stream.  Every step still runs
the whole path on its own batch of 32 images."""
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from yolov6.utils.synth import build_synthetic
from yolov6.utils.torch_utils import fuse_model
from yolov6.layers.common import RepVGGBlock
from yolov6.hip import runtime
from yolov6.hip.runtime import Engine

m = build_synthetic('configs/yololps.py', sigma=0.25)
m = fuse_model(m).eval()
for l in m.modules():
    if isinstance(l, RepVGGBlock): l.switch_to_deploy()
m = m.cuda().half()
B = 32
xs = [torch.rand(B, 3, 640, 640, generator=torch.Generator().manual_seed(1234 + i)).cuda().half() for i in range(2)]
NE = int(sys.argv[1]) if len(sys.argv) > 1 else 2
GRAPH = len(sys.argv) > 2 and sys.argv[2] == 'graph'
engines = [Engine.from_model(m, torch.float16, torch.device('cuda:0')) for _ in range(NE)]
streams = [torch.cuda.Stream() for _ in range(NE)]
nms_done = [None] * NE
if GRAPH:
    for e in engines: e.set_graph(True)
s_post = torch.cuda.Stream()
for e, s in zip(engines, streams):
    with torch.cuda.stream(s):
        e.forward(xs[0])
torch.cuda.synchronize()

def step(i):
    e, s = engines[i % NE], streams[i % NE]
    if GRAPH and nms_done[i % NE] is not None:
        s.wait_event(nms_done[i % NE])      # the engine's persistent pred buffer is free again
    with torch.cuda.stream(s):
        pred = e.forward(xs[i % 2])
        ready = torch.cuda.Event(); ready.record(s)
    s_post.wait_event(ready)
    pred.record_stream(s_post)
    with torch.cuda.stream(s_post):
        out = runtime.nms_padded(pred, 0.4, 0.45, 1000)
        if GRAPH:
            nms_done[i % NE] = torch.cuda.Event(); nms_done[i % NE].record(s_post)
        return out

with torch.no_grad():
    for i in range(6): step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    K = 40
    for i in range(K): out = step(i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
print('%d engine(s) in flight%s: %.1f images/s, %.3f ms per step' % (NE, ' + hipGraph' if GRAPH else '', K * B / dt, dt / K * 1e3))
