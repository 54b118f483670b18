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
