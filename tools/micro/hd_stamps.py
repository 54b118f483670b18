"""Where a wave of head_det_kernel spends its clocks (needs the stamps build: make -C yolo-lp_amd/csrc stamps;
LP_HIP_LIB=yolo-lp_amd/libyololp_hip_stamps.so).    python tools/micro/hd_stamps.py"""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from yolov6.utils.synth import build_synthetic
from yolov6.utils.torch_utils import fuse_model
from yolov6.layers.common import RepVGGBlock
from yolov6.hip import runtime
m = fuse_model(build_synthetic(os.path.join(ROOT, 'configs', 'yololps.py'), sigma=bench.SIGMA['yololps'])).eval()
for layer in m.modules():
    if isinstance(layer, RepVGGBlock):
        layer.switch_to_deploy()
m = m.cuda().half()
x = torch.rand(32, 3, 640, 640, generator=torch.Generator().manual_seed(1)).cuda().half()
eng = runtime.engine_for(m)
eng.autotune = False
stamps = torch.zeros(1 << 21, dtype=torch.int64, device='cuda')
eng.lib.lpdbg_set_stamps(ctypes.c_void_p(stamps.data_ptr()))
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
with torch.no_grad():
    eng.detect(x, 0.4, 0.45, 1000, route='det')
    stamps.zero_()
    eng.detect(x, 0.4, 0.45, 1000, route='det')
torch.cuda.synchronize()
names = ['MFMA (+ operand request)', 'fold maxima', 'barrier A', 'step 2 (wave 0) ', 'barrier B', 'operand wait', 'step 3', 'prologue (once)']
for nch in (1, 2, 4):
    st = stamps[(1 << 18) + nch * (1 << 14):(1 << 18) + (nch + 1) * (1 << 14)].view(-1, 16).cpu()
    st = st[(st[:, 9] & 0xff) == 4]
    if not len(st):
        continue
    tiles = (st[:, 9] >> 8).float()
    life = st[:, 8].float()
    print('NCH %d: %d waves, tiles per wave %.2f, life mean %.0f clocks (min %.0f, p50 %.0f, p90 %.0f, max %.0f)' % (
        nch, len(st), tiles.mean(), life.mean(), life.min(), life.median(), life.kthvalue(int(0.9 * len(life))).values, life.max()))
    for k, n in enumerate(names):
        per = st[:, k].float() / (tiles if k < 7 else 1)
        print('  %-26s %8.0f clocks%s   (wave 0: %8.0f, waves 1-2: %8.0f)' % (n, per.mean(), ' per tile' if k < 7 else '         ', per[0::3].mean(), torch.cat([per[1::3], per[2::3]]).mean()))
