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
from oracle import lp_oracle, lp_post, lp_metric as metric_oracle  # noqa: E402
from lp_testing import unpack_lists  # noqa: E402

MODEL_CASES = [('lps_tiny_128x96', 'lps_tiny_weights', 'yololps'),
               ('lps_tiny_64x160', 'lps_tiny_weights', 'yololps'),
               ('v6m_tiny_96x128', 'v6m_tiny_weights', 'yolov6m')]
# P6 and plain-PAN assemblies (tests/golden/make_golden_p6.py): (case, config, oracle arch overrides, build overrides)
P6_CASES = [
    ('s6_tiny_128x192', 'yolov6s6', {}, {}),
    ('m6_tiny_128x64', 'yolov6m6', {}, {}),
    ('s6pan_tiny_64x128', 'yolov6s6', dict(bifusion=False), dict(neck='RepPANNeck6', fuse_P2=False)),
    ('s6csppan_tiny_128x128', 'yolov6s6', dict(bifusion=False, csp_neck=True), dict(neck='CSPRepPANNeck_P6', fuse_P2=False)),
    ('span_tiny_96x64', 'yololps', dict(bifusion=False), dict(neck='RepPANNeck', fuse_P2=False)),
    ('mpan_tiny_64x96', 'yolov6m', dict(bifusion=False), dict(neck='CSPRepPANNeck', fuse_P2=False)),
]
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


@pytest.mark.parametrize('case,name,arch_kw,build_kw', P6_CASES, ids=[c[0] for c in P6_CASES])
def test_p6_and_pan_oracle_and_mirror_match_reference(case, name, arch_kw, build_kw):
    """The six other assemblies of the reference (P6 backbones / necks, plain PAN necks): oracle within the fused-vs-
    unfused floor, mirror bit-exact (same seeded weights, same ops in the same order)."""
    from yolov6.utils.synth import build_synthetic
    g, sd = load_golden(case), load_golden(case + '_weights')
    a = lp_oracle.arch(name, width=0.0625, depth=0.25, **arch_kw)
    pred, neck = lp_oracle.forward(sd, a, g['x'])
    nlev = 4 if a.p6 else 3
    assert len(neck) == nlev
    for i in range(nlev):
        torch.testing.assert_close(neck[i], g['neck%d' % i], rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(pred, g['pred'], rtol=1e-4, atol=1e-3)
    m = build_synthetic(os.path.join(REPO, 'configs', name + '.py'), width=0.0625, depth=0.25, sigma=1.5, **build_kw)
    msd = m.state_dict()
    assert list(msd.keys()) == list(sd.keys())
    for k in sd:
        if sd[k].is_floating_point():
            assert torch.equal(msd[k].half().float(), sd[k]), k
    m.load_state_dict(sd)
    nthreads = torch.get_num_threads()
    torch.set_num_threads(4)
    try:
        with torch.no_grad():
            mp, feats = m(g['x'].clone())
    finally:
        torch.set_num_threads(nthreads)
    assert torch.equal(mp, g['pred'])
    for i in range(nlev):
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


METRIC_CASES = ['metric_synth_a', 'metric_synth_b', 'metric_crafted']


@pytest.mark.parametrize('case', METRIC_CASES)
def test_lp_metric_oracle_matches_reference(case):
    """oracle/lp_metric.py against the outputs of the reference's own Evaler.eval (tests/golden/make_golden_metric.py)."""
    z = load_golden(case)
    preds, targets = unpack_lists(z, 'pred', 28), unpack_lists(z, 'tgt', 20)
    out = metric_oracle.evaluate([[p.numpy() for p in b] for b in preds], [[t.numpy() for t in b] for b in targets])
    assert out[:5] == z['scalars'].tolist()               # python floats from integer ratios: exact
    assert out[5] == z['mAP_list'].tolist() and out[6] == z['recall_list'].tolist()


@pytest.mark.parametrize('case', METRIC_CASES)
def test_lp_metric_mirror_cpu_matches_reference(case):
    """yolov6.utils.lp_metric (vectorised CPU path) and Evaler.eval of the mirror: same counters as the oracle, same
    seven results as the reference."""
    from yolov6.utils import lp_metric
    from yolov6.core.evaler import Evaler
    z = load_golden(case)
    preds, targets = unpack_lists(z, 'pred', 28), unpack_lists(z, 'tgt', 20)
    c = lp_metric.counts(preds, targets)
    ref_c = metric_oracle.counts([[p.numpy() for p in b] for b in preds], [[t.numpy() for t in b] for b in targets], strict=False)
    assert c.tolist() == ref_c.tolist()
    ev = Evaler(None, device=torch.device('cpu'), half=False)
    out = ev.eval(preds, targets, None, 'val')
    assert out[:5] == z['scalars'].tolist() and out[5] == z['mAP_list'].tolist() and out[6] == z['recall_list'].tolist()


def test_lp_metric_unbinned_labels_are_reported():
    """IoU == 1 fits no bin: the reference re-uses a stale bin index (or raises); oracle strict mode does the same, the
    batched implementations skip the label and count it."""
    from yolov6.utils import lp_metric
    pred = torch.zeros(1, 28)
    pred[0, :4] = torch.tensor([10., 10., 50., 30.])
    tgt = torch.zeros(1, 20)
    tgt[0, 8:12] = pred[0, :4]
    c = lp_metric.counts([[pred]], [[tgt]])
    assert int(c[lp_metric.UNBINNED]) == 1 and int(c[lp_metric.PRED]) == 1 and int(c[lp_metric.TRUE]) == 1
    assert c[lp_metric.PRED_BINS:lp_metric.UNBINNED].sum() == 0
    with pytest.raises(UnboundLocalError):
        metric_oracle.counts([[pred.numpy()]], [[tgt.numpy()]], strict=True)
    assert metric_oracle.counts([[pred.numpy()]], [[tgt.numpy()]], strict=False).tolist() == c.tolist()


def test_evaler_split_targets_matches_reference_loop():
    """Evaler.split_targets against the reference's per-row python loop (evaler.py:120-128) restated here."""
    from yolov6.core.evaler import Evaler
    from yolov6.utils.nms import xywh2xyxy
    g = torch.Generator().manual_seed(3)
    T, B, h, w = 9, 4, 96, 160
    targets = torch.rand(T, 21, generator=g)
    targets[:, 0] = torch.tensor([0, 0, 2, 3, 3, 3, 0, 2, 3]).float()
    exp_t = targets.clone()
    exp_t[:, 9:13] = xywh2xyxy(exp_t[:, 9:13])
    exp = [torch.zeros((0, 20))] * B
    for row in exp_t:
        for j in range(9, 21, 2):
            row[j] = row[j] * w
            row[j + 1] = row[j + 1] * h
        exp[int(row[0])] = torch.cat((exp[int(row[0])], row[None, 1:]), dim=0)
    got = Evaler(None, device=torch.device('cpu'), half=False).split_targets(targets, B, h, w)
    assert len(got) == B and all(torch.equal(a, b) for a, b in zip(got, exp))
