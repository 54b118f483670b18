"""Where a wave of stem2_fused_kernel spends its clocks (needs the stamps build: make -C yolo-lp_amd/csrc stamps; LP_HIP_LIB=yolo-lp_amd/libyololp_hip_stamps.so).
    python tools/micro/fz_stamps.py"""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from yolov6.utils.synth import build_synthetic
from yolov6.utils.torch_utils import fuse_model
from yolov6.layers.common import RepVGGBlock
from yolov6.hip import runtime, abi
m = fuse_model(build_synthetic(os.path.join(ROOT, 'configs', 'yololps.py'), sigma=bench.SIGMA['yololps'])).eval()
for layer in m.modules():
    if isinstance(layer, RepVGGBlock):
        layer.switch_to_deploy()
m = m.cuda().half()
x = torch.rand(32, 3, 640, 640, generator=torch.Generator().manual_seed(1)).cuda().half()
eng = runtime.engine_for(m)
eng.autotune = False
stamps = torch.zeros(1 << 20, dtype=torch.int64, device='cuda')
eng.lib.lpdbg_set_stamps(ctypes.c_void_p(stamps.data_ptr()))
with torch.no_grad():
    eng.forward(x)
    eng.set_variant(2, abi.LP_VARIANT_FUSED_STEM2, 3)
    stamps.zero_()
    eng.forward(x)
torch.cuda.synchronize()
st = stamps.view(-1, 8).cpu()
st = st[(st[:, 7] & 0xff) == 3]
tiles = (st[:, 7] >> 8).float()
names = ['wait window', 'barrier 1', 'window request', 'stage A', 'barrier 2', 'stage B + stores']
print('%d waves, tiles per wave %.1f, life %.0f clocks (%.2f clocks per tile)' % (len(st), tiles.mean(), st[:, 6].float().mean(), (st[:, 6].float() / tiles).mean()))
for k, n in enumerate(names):
    print('  %-18s %8.0f clocks per tile  (wave 0: %8.0f, waves 3..7: %8.0f)' % (n, (st[:, k].float() / tiles).mean(), (st[0::8, k].float() / tiles[0::8]).mean(),
                                                                           torch.stack([(st[w::8, k].float() / tiles[w::8]).mean() for w in range(3, 8)]).mean()))
