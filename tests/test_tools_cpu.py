"""CPU plumbing of the entry points (BASELINE configs[0]: yololps, one CCPD-format 720x1160 image through
tools/infer.py on the PyTorch CPU path), with a tiny-width model so it runs in seconds, and the pure-torch
helpers of Inferer pinned against hand-computed values."""
import os
import sys

import numpy as np
import pytest
import torch

from conftest import REPO


@pytest.fixture()
def workdir(tmp_path, monkeypatch):
    monkeypatch.chdir(REPO)                     # the tools resolve `yolov6` from the current directory
    if REPO not in sys.path:
        sys.path.insert(0, REPO)
    return tmp_path


def _make_checkpoint(path, width=0.0625):
    from yolov6.utils.synth import build_synthetic
    m = build_synthetic(os.path.join(REPO, 'configs', 'yololps.py'), width=width, sigma=1.5)
    torch.save({'model': m.half(), 'ema': None, 'epoch': 0}, path)      # the reference's checkpoint layout


def _make_image(path, h=1160, w=720, seed=0):
    from PIL import Image
    rng = np.random.default_rng(seed)
    Image.fromarray(rng.integers(0, 255, (h, w, 3), dtype=np.uint8)).save(path)


def test_infer_cli_cpu(workdir):
    sys.path.insert(0, os.path.join(REPO, 'tools'))
    import importlib
    infer = importlib.import_module('infer')
    ckpt, img_dir, out = workdir / 'tiny.pt', workdir / 'imgs', workdir / 'out'
    img_dir.mkdir()
    _make_checkpoint(str(ckpt))
    _make_image(str(img_dir / 'plate0.png'))
    res = infer.run(weights=str(ckpt), source=str(img_dir), yaml=None, img_size=[640, 640], conf_thres=0.06,
                    iou_thres=0.45, max_det=50, device='cpu', save_txt=True, not_save_img=False, save_dir=str(out),
                    half=False)
    assert len(res) == 1 and res[0].shape[1] == 28
    assert (out / 'imgs' / 'plate0.png').exists()     # the reference mirrors the source's directory under save_dir
    if len(res[0]):
        lines = (out / 'imgs' / 'plate0.txt').read_text().strip().splitlines()
        assert len(lines) == len(res[0]) and len(lines[0].split()) == 8 + 4 + 8
        assert float(res[0][:, :12].min()) >= 0 and float(res[0][:, 0:12:2].max()) <= 720


def test_letterbox_ccpd_shape():
    from yolov6.data.data_augment import letterbox
    img = np.zeros((1160, 720, 3), np.uint8)
    out, r, (dw, dh) = letterbox(img, [640, 640], stride=32)
    assert out.shape == (640, 416, 3)            # a CCPD frame becomes 640x416, not 640x640
    assert r == pytest.approx(640 / 1160)


def test_rescale_and_box_convert():
    from yolov6.core.inferer import Inferer
    det = torch.tensor([[100., 50., 300., 250., 10., 20., 30., 40., 500., 700., 0., 0.]])
    out = Inferer.rescale((640, 416), det.clone(), (1160, 720, 3))
    ratio = min(640 / 1160, 416 / 720)
    pad = ((416 - 720 * ratio) / 2, (640 - 1160 * ratio) / 2)
    exp = det.clone()
    exp[:, 0::2] = ((det[:, 0::2] - pad[0]) / ratio).clamp(0, 720)
    exp[:, 1::2] = ((det[:, 1::2] - pad[1]) / ratio).clamp(0, 1160)
    assert torch.allclose(out, exp)
    xywh = Inferer.box_convert(torch.tensor([[10., 20., 30., 60.]]))
    assert xywh.tolist() == [[20., 40., 20., 40.]]


def test_inferer_statics_match_the_reference_fixtures(golden):
    """Row a13: ``Inferer.rescale`` (+ ``.round()``), ``Inferer.precess_image`` and the letterbox geometry against
    fixtures captured from the reference's own inferer.py (tests/golden/make_golden_inferer.py; cv2 absent there: frames
    that need no resize are pinned pixel for pixel, resized frames by geometry only)."""
    from yolov6.core.inferer import Inferer
    from yolov6.data.data_augment import letterbox_geometry
    z = golden('inferer_ref')
    for k in range(int(z['rescale_n'])):
        ori, tgt = tuple(z['rescale_%d_ori' % k].tolist()), tuple(z['rescale_%d_tgt' % k].tolist())
        out = Inferer.rescale(ori, z['rescale_%d_in' % k].clone(), tgt)
        assert torch.equal(out, z['rescale_%d_out' % k]), k
        assert torch.equal(out.round(), z['rescale_%d_round' % k]), k
    for k in range(int(z['pre_n'])):
        frame = z['pre_%d_frame' % k].numpy()
        size, half = int(z['pre_%d_size' % k]), bool(int(z['pre_%d_half' % k]))
        img, src = Inferer.precess_image(frame, [size, size], 32, half)
        assert img.dtype == (torch.float16 if half else torch.float32) and src is frame
        assert torch.equal(img.float(), z['pre_%d_out' % k]), k
    for k in range(int(z['geo_n'])):
        h, w, size = z['geo_%d_in' % k].tolist()
        r, new_unpad, (top, bottom, left, right), _ = letterbox_geometry((h, w), [size, size], stride=32)
        assert list(new_unpad) == z['geo_%d_resized_wh' % k].tolist(), k
        assert [new_unpad[1] + top + bottom, new_unpad[0] + left + right] == z['geo_%d_shape' % k].tolist(), k
        assert [top, left, top + new_unpad[1], left + new_unpad[0]] == z['geo_%d_box' % k].tolist(), k


def test_eval_speed_cli_cpu(workdir):
    sys.path.insert(0, os.path.join(REPO, 'tools'))
    import importlib
    ev = importlib.import_module('eval')
    ckpt, img_dir = workdir / 'tiny.pt', workdir / 'imgs'
    img_dir.mkdir()
    _make_checkpoint(str(ckpt))
    for i in range(3):
        _make_image(str(img_dir / ('f%d.png' % i)), 240, 320, i)
    preds, speed, metrics = ev.run(str(img_dir), weights=str(ckpt), batch_size=2, img_size=128, conf_thres=0.4, iou_thres=0.45,
                                   task='speed', device='cpu', half=False, save_dir=str(workdir / 'val'), name='exp')
    assert sum(len(b) for b in preds) == 3 and all(v >= 0 for v in speed) and metrics is None


def test_eval_val_cli_reads_labels_and_reports_the_lp_metric(workdir):
    """--task val: label files beside the images go through the reference's letterbox transform and Evaler.eval."""
    sys.path.insert(0, os.path.join(REPO, 'tools'))
    import importlib
    import numpy as np
    ev = importlib.import_module('eval')
    ckpt = workdir / 'tiny.pt'
    img_dir, lab_dir = workdir / 'ds' / 'images' / 'val', workdir / 'ds' / 'labels' / 'val'
    img_dir.mkdir(parents=True)
    lab_dir.mkdir(parents=True)
    _make_checkpoint(str(ckpt))
    for i in range(3):
        _make_image(str(img_dir / ('f%d.png' % i)), 240, 320, i)
    row = [1, 2, 3, 4, 5, 6, 7, 8, 0.5, 0.5, 0.25, 0.2, 0.375, 0.4, 0.625, 0.4, 0.625, 0.6, 0.375, 0.6]
    (lab_dir / 'f0.txt').write_text(' '.join(str(v) for v in row) + '\n' + ' '.join(str(v) for v in row) + '\n')
    (lab_dir / 'f2.txt').write_text(' '.join(str(v) for v in row) + '\n')
    # letterbox of a 240x320 frame into 128x128: ratio 0.4, content 96x128 at top 16
    lab = ev.letterbox_labels(np.array([row], dtype=np.float32), 240, 320, 0.4, (0.0, 16.0), 128, 128)
    assert np.allclose(lab[0, 8:12], [0.5, 0.5, 0.25, 0.2 * 96 / 128], atol=1e-6)
    assert np.allclose(lab[0, 12:14], [0.375, (0.4 * 96 + 16) / 128], atol=1e-6)
    batches = list(ev.image_batches(str(img_dir), 128, 2))
    assert [b[1].shape for b in batches] == [(2, 21), (1, 21)] and batches[1][1][0, 0] == 0
    preds, speed, metrics = ev.run(str(img_dir), weights=str(ckpt), batch_size=2, img_size=128, conf_thres=0.03, iou_thres=0.65,
                                   task='val', device='cpu', half=False, save_dir=str(workdir / 'val'), name='exp')
    assert metrics is not None and len(metrics) == 7 and len(metrics[5]) == 10 and 0.0 <= metrics[4] <= 1.0


def test_roofline_from_trace_arithmetic(tmp_path):
    """tools/roofline_from_trace.py (the figure bench.py reports as roofline.frac): on a synthetic kernel trace with known
    durations it must pick exactly the 3x3 dispatches of the timed steps -- not the warm-up, not the trailing NMS timings --
    and divide the step's algorithmic FLOPs by their summed duration; tools/micro/fwd_idle.py reads the same trace."""
    import json
    import subprocess
    import sys
    steps, warm = 4, 3
    hdr = ['Kind', 'Agent_Id', 'Queue_Id', 'Stream_Id', 'Thread_Id', 'Dispatch_Id', 'Kernel_Id', 'Kernel_Name', 'Correlation_Id',
           'Start_Timestamp', 'End_Timestamp', 'LDS_Block_Size', 'Scratch_Size', 'VGPR_Count', 'Accum_VGPR_Count', 'SGPR_Count',
           'Workgroup_Size_X', 'Workgroup_Size_Y', 'Workgroup_Size_Z', 'Grid_Size_X', 'Grid_Size_Y', 'Grid_Size_Z']
    rows, t = [], [1000]

    def k(name, us, gap_us=0):
        t[0] += int(gap_us * 1000)
        rows.append(['KERNEL_DISPATCH', 'Agent 2', 1, 0, 1, len(rows) + 1, 1, name, len(rows) + 1, t[0], t[0] + int(us * 1000), 0, 0, 64, 0,
                     32, 256, 1, 1, 65536, 1, 1])
        t[0] += int(us * 1000)

    def step(scale):
        k('_ZN2lp18stem2_fused_kernelIDF16_Li2ELi2ELb1EEEvNS_8ConvArgsEi', 80 * scale)
        k('_ZN2lp19conv3x3_pipe_kernelIDF16_Li0ELb0EEEvNS_8ConvArgsEi', 60 * scale)
        k('_ZN2lp16conv_mfma_kernelIDF16_Li3ELi3ELi2ELi0ELi2EEEvNS_8ConvArgsE', 40 * scale)      # 3x3 stride 2, generic
        k('_ZN2lp16conv_mfma_kernelIDF16_Li5ELi1ELi1ELi0ELi2EEEvNS_8ConvArgsE', 15 * scale)      # a 1x1 layer: not counted
        k('_ZN2lp21conv1x1_stream_kernelIDF16_Li2ELi1ELb0ELb0EEEvNS_8ConvArgsEii', 10 * scale)
        k('lp::sort_kernel(unsigned long long*, int const*, int)', 5, gap_us=2)
        k('lp::greedy_kernel(unsigned long long const*, float const*)', 5)

    for _ in range(warm):
        step(3.0)                        # slow warm-up steps must not enter the figure
    for _ in range(steps):
        step(1.0)
    for _ in range(5):                   # bench.py's five trailing forward + NMS timings
        step(2.0)
    d = tmp_path / 'kt'
    d.mkdir()
    with open(d / 'x_kernel_trace.csv', 'w') as f:
        f.write(','.join('"%s"' % h for h in hdr) + '\n')
        for r in rows:
            f.write(','.join('"%s"' % v if isinstance(v, str) else str(v) for v in r) + '\n')
    bench = {'value_inflight1': 1.0, 'config': {'workload': 'synthetic'},
             'roofline': {'flops_per_launch': 30.0, 'launches': 3, 'frac': 0.5, 'frac_event': 0.5,
                          'backbone_dispatches': 2, 'backbone_flops': 70e9}}      # backbone = the first two forward kernels of a step
    (tmp_path / 'bench.json').write_text(json.dumps(bench) + '\n')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, 'tools', 'roofline_from_trace.py'), str(d), str(tmp_path / 'bench.json'),
                          '--steps', str(steps)], capture_output=True, text=True, check=True).stdout
    r = json.loads(out)
    assert r['dispatches_per_step'] == 3.0 and abs(r['conv3_us_per_step'] - 180.0) < 1e-6
    assert abs(r['achieved_tflops'] - 90e9 / 180e-6 / 1e12) < 0.06 and abs(r['frac'] - 0.2) < 1e-3
    # the backbone fraction: the first two of the five forward kernels of each timed step (80 + 60 us), NMS kernels not counted
    assert r['forward_dispatches_per_step'] == 5 and abs(r['backbone_us_per_step'] - 140.0) < 1e-6
    assert abs(r['backbone_frac'] - 70e9 / 140e-6 / 1e12 / 2500.0) < 1e-3
    # the whole forward of a step (the denominator of a bandwidth-bound configuration's roofline): all five kernels, not the NMS's
    assert abs(r['forward_us_per_step'] - 205.0) < 1e-6
    from yolov6.hip.srchash import source_hash
    assert r['kernel_source_hash'] == source_hash()
    idle = subprocess.run([sys.executable, os.path.join(root, 'tools', 'micro', 'fwd_idle.py'), str(d), str(steps)],
                          capture_output=True, text=True, check=True).stdout
    assert 'idle 2.0 us' in idle and 'idle before lp::sort_kernel' in idle, idle


def test_pmc_traffic_arithmetic(tmp_path):
    """tools/pmc_traffic.py (roofline.traffic): two synthetic rocprofv3 --pmc passes with known counter values.  The per-layer figure
    takes the 3x3 dispatches of the last `steps` steps only (mangled names of fp16 AND bf16 instantiations: rocprofv3's demangler garbles
    the latter, the profile runs use -M), FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950; the whole-forward figure
    takes every kernel of those steps but the NMS's."""
    import json
    import subprocess
    import sys
    steps, warm = 3, 2

    def write(d, counter, scale):
        d.mkdir()
        rows = []

        def k(name, val):
            rows.append((len(rows) + 1, name, counter, val))

        def step(f):
            k('_ZN2lp22conv3x3_pipe16v_kernelIDF16bLi10ELb0EEEvNS_8ConvArgsEi.kd', 1000 * f)         # bf16 instantiation, mangled
            k('_ZN2lp16conv_mfma_kernelIDF16_Li3ELi3ELi2ELi0ELi2EEEvNS_8ConvArgsE.kd', 500 * f)       # 3x3 stride 2
            k('_ZN2lp16conv_mfma_kernelIDF16bLi5ELi1ELi1ELi0ELi2EEEvNS_8ConvArgsE.kd', 200 * f)       # a 1x1 layer: forward, not 3x3
            k('_ZN2lp12score_kernelEPKfiiPyPi.kd', 7000)                                               # NMS: in neither figure
            k('_ZN2lp11sort_kernelEPyPKii.kd', 9000)
        for _ in range(warm):
            step(10 * scale)
        for _ in range(steps):
            step(scale)
        with open(d / 'p_counter_collection.csv', 'w') as f:
            f.write('"Dispatch_Id","Kernel_Name","Counter_Name","Counter_Value"\n')
            for r in rows:
                f.write('%d,"%s","%s",%s\n' % r)
    write(tmp_path / 'fetch', 'FETCH_SIZE', 1.0)
    write(tmp_path / 'write', 'WRITE_SIZE', 0.5)
    (tmp_path / 'bench.json').write_text(json.dumps({'roofline': {'launches': 1, 'conv3_layers': 2}}) + '\n')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, 'tools', 'pmc_traffic.py'), str(tmp_path / 'fetch'), str(tmp_path / 'write'),
                          '--steps', str(steps), '--bench', str(tmp_path / 'bench.json')], capture_output=True, text=True, check=True).stdout
    r = json.loads(out)
    assert r['dispatches_per_step'] == 2 and r['layers_per_step'] == 2 and r['dispatches_averaged'] == 2 * steps
    assert r['fetch_bytes_per_launch'] == round(1500 / 2 * 1024 * 2) and r['write_bytes_per_launch'] == round(750 / 2 * 1024)
    assert r['hbm_bytes_per_launch'] == r['fetch_bytes_per_launch'] + r['write_bytes_per_launch']
    assert r['forward_hbm_bytes_per_step'] == round(1700 * 1024 * 2 + 850 * 1024)
