"""Pins the ORACLE (oracle/) and the host-side mirror's CPU path against the
golden vectors recorded from the reference itself (tests/golden/make_golden.py).
CPU only."""
import os
import sys

import numpy as np
import pytest
import torch

from conftest import load_golden, REPO

sys.path.insert(0, os.path.join(REPO))
from oracle import lp_oracle, lp_post  # noqa: E402

MODEL_CASES = [('lps_tiny_128x96', 'lps_tiny_weights', 'yololps'),
               ('lps_tiny_64x160', 'lps_tiny_weights', 'yololps'),
               ('v6m_tiny_96x128', 'v6m_tiny_weights', 'yolov6m')]
NMS_CASES = ['nms_model_tiny', 'nms_synth_600', 'nms_synth_maxdet5', 'nms_synth_obj', 'nms_crafted', 'nms_empty']


@pytest.mark.parametrize('case,weights,name', MODEL_CASES)
def test_forward_oracle_matches_reference(case, weights, name):
    g, sd = load_golden(case), load_golden(weights)
    a = lp_oracle.arch(name, width=0.0625)
    pred, neck, bb = lp_oracle.forward(sd, a, g['x'], return_stages=True)
    # The oracle folds BN before the conv (deploy form) while the fixture is the reference's un-fused
    # forward: the reference's own fused-vs-unfused difference on these cases is 5e-5 .. 4e-4 absolute
    # (printed by make_golden.py), so that is the floor of any cross-implementation comparison.
    for i in range(4):
        torch.testing.assert_close(bb[i], g['bb%d' % i], rtol=1e-4, atol=1e-4)
    for i in range(3):
        torch.testing.assert_close(neck[i], g['neck%d' % i], rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(pred, g['pred'], rtol=1e-4, atol=1e-3)
    assert torch.equal(pred[..., 4], torch.ones_like(pred[..., 4]))


@pytest.mark.parametrize('case,weights,name', MODEL_CASES)
def test_forward_mirror_cpu_is_bit_exact(case, weights, name):
    """Same torch ops in the same order as the reference => identical bits on CPU."""
    from yolov6.utils.synth import build_synthetic
    g, sd = load_golden(case), load_golden(weights)
    m = build_synthetic(os.path.join(REPO, 'configs', name + '.py'), width=0.0625, sigma=1.5)
    msd = m.state_dict()
    assert list(msd.keys()) == list(sd.keys())
    # the seeded recipe reproduces the reference's weights (up to the fp16 rounding applied to the fixture)
    for k in sd:
        if sd[k].is_floating_point():
            assert torch.equal(msd[k].half().float(), sd[k]), k
    m.load_state_dict(sd)
    nthreads = torch.get_num_threads()
    torch.set_num_threads(4)      # the fixtures were recorded with 4 threads; ATen reductions depend on it
    try:
        with torch.no_grad():
            pred, feats = m(g['x'].clone())
    finally:
        torch.set_num_threads(nthreads)
    assert torch.equal(pred, g['pred'])
    for i in range(3):
        assert torch.equal(feats[i], g['neck%d' % i])


@pytest.mark.parametrize('case', NMS_CASES)
def test_post_oracles_match_reference(case):
    g = load_golden(case)
    conf, iou, max_det = float(g['conf']), float(g['iou']), int(g['max_det'])
    pred = g['pred'].numpy()
    rows_c, keep_c, after = lp_post.nms_c(pred, conf, iou, max_det)
    rows_n, keep_n = lp_post.nms_np(pred, conf, iou, max_det)
    assert [len(r) for r in rows_c] == g['counts'].tolist()
    for b in range(pred.shape[0]):
        ref = g['det%d' % b].numpy()
        assert np.array_equal(rows_c[b], ref), case
        assert np.array_equal(rows_n[b], ref), case
        assert np.array_equal(keep_c[b], keep_n[b])
    if 'pred_after' in g:
        assert np.array_equal(after, g['pred_after'].numpy())
    else:
        assert np.array_equal(after, pred)


@pytest.mark.parametrize('case', NMS_CASES)
def test_nms_mirror_cpu_matches_reference(case):
    from yolov6.utils.nms import non_max_suppression
    g = load_golden(case)
    p = g['pred'].clone()
    out = non_max_suppression(p, float(g['conf']), float(g['iou']), max_det=int(g['max_det']))
    assert len(out) == p.shape[0]
    for b, o in enumerate(out):
        assert o.shape[1] == 28 and o.dtype == torch.float32
        assert torch.equal(o, g['det%d' % b]), case
    if 'pred_after' in g:
        assert torch.equal(p, g['pred_after'])


def test_nms_argument_checks():
    from yolov6.utils.nms import non_max_suppression
    with pytest.raises(AssertionError):
        non_max_suppression(torch.zeros(1, 4, 290), conf_thres=1.5)
    with pytest.raises(AssertionError):
        non_max_suppression(torch.zeros(1, 4, 290), iou_thres=-0.1)
    out = non_max_suppression(torch.zeros(2, 0, 290))
    assert [tuple(o.shape) for o in out] == [(0, 28), (0, 28)]


@pytest.mark.parametrize('name', ['yololps', 'yololpn'])
def test_full_size_digest(name):
    """Full-size seeded model of the mirror == the reference's (weights by seed, sampled output rows)."""
    from yolov6.utils.synth import build_synthetic
    g = load_golden('digest_%s_640' % name)
    m = build_synthetic(os.path.join(REPO, 'configs', name + '.py'), sigma=0.35)
    wsum = sum(v.double().sum().item() for v in m.state_dict().values() if v.is_floating_point())
    assert wsum == pytest.approx(float(g['weight_sum']), rel=1e-12)
    x = torch.rand(1, 3, 640, 640, generator=torch.Generator().manual_seed(1234))
    with torch.no_grad():
        pred, _ = m(x)
    torch.testing.assert_close(pred[0, g['rows']], g['pred_rows'], rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(pred[0].double().sum(0), g['colsum'], rtol=1e-6, atol=1e-4)
