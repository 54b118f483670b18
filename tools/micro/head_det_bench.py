"""Durations of the detections-only head kernels in isolation (run under rocprofv3 --kernel-trace; see tools/micro/r3_hd.sh)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from yolov6.utils.synth import build_synthetic
from yolov6.utils.torch_utils import fuse_model
from yolov6.layers.common import RepVGGBlock
from yolov6.hip import runtime
name = sys.argv[1] if len(sys.argv) > 1 else 'yololps'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
m = fuse_model(build_synthetic(os.path.join(ROOT, 'configs', name + '.py'), sigma=bench.SIGMA[name])).eval()
for layer in m.modules():
    if isinstance(layer, RepVGGBlock):
        layer.switch_to_deploy()
m = m.cuda().half()
x = torch.rand(B, 3, 640, 640, generator=torch.Generator().manual_seed(1)).cuda().half()
eng = runtime.engine_for(m)
eng.autotune = False
eng.set_single_lane(True)
with torch.no_grad():
    for _ in range(12):
        eng.detect(x, 0.4, 0.45, 1000, route='det')
torch.cuda.synchronize()
