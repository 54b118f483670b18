"""Candidate counts per image of the bench configurations (synthetic weights): what the post-processing kernels work on.
    python tools/micro/nms_counts.py [--model yololpn --batch 128 --size 640 --dtype f16]"""
import argparse, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
ap = argparse.ArgumentParser()
ap.add_argument('--model', default='yololps'); ap.add_argument('--batch', type=int, default=32)
ap.add_argument('--size', type=int, default=640); ap.add_argument('--dtype', default='f16')
ap.add_argument('--conf', type=float, default=0.4)
a = ap.parse_args()
from yolov6.utils.synth import build_synthetic
from yolov6.utils.torch_utils import fuse_model
from yolov6.layers.common import RepVGGBlock
from yolov6.hip import runtime
tdt = {'f16': torch.float16, 'bf16': torch.bfloat16}[a.dtype]
model = fuse_model(build_synthetic(os.path.join(ROOT, 'configs', a.model + '.py'), sigma=bench.SIGMA[a.model])).eval()
for layer in model.modules():
    if isinstance(layer, RepVGGBlock):
        layer.switch_to_deploy()
model = model.cuda().to(tdt)
x = torch.rand(a.batch, 3, a.size, a.size, generator=torch.Generator().manual_seed(1235)).cuda().to(tdt)
eng = runtime.engine_for(model)
ws = eng.forward_det(x, a.conf)
torch.cuda.synchronize()
off = (ws[0].data_ptr() + 255) // 256 * 256 - ws[0].data_ptr()
cnt = ws[0][off:off + 4 * a.batch].view(torch.int32).cpu()
det, count, _ = runtime.nms_candidates(ws, 0.45, 1000)
torch.cuda.synchronize()
count = count.cpu()
print(a.model, a.size, 'candidates per image: min %d median %d max %d; kept (iou 0.45, max_det 1000): min %d median %d max %d'
      % (int(cnt.min()), int(cnt.median()), int(cnt.max()), int(count.min()), int(count.median()), int(count.max())))
