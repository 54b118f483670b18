"""Randomised parity sweep of the conv kernels: random channel counts, kernel / stride, ragged map sizes, batch, sources,
residual and activation; EVERY kernel variant the op accepts is run and compared (a) bit for bit with the first variant and
(b) with torch's conv2d on the CPU within the dtype's tolerance.  Not part of the test-suite (minutes); run on a GPU box:
    python tools/micro/random_conv_sweep.py [cases] [seed]"""
import os, sys, random, torch
import torch.nn.functional as F
sys.path.insert(0, os.getcwd())
from yolov6.hip import abi
from yolov6.hip.runtime import Engine

TOL = {torch.float32: 2e-5, torch.float16: 4e-3, torch.bfloat16: 3e-2}
ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for case in range(ncases):
    dtype = rng.choice([torch.float32, torch.float16, torch.float16, torch.bfloat16])
    k = rng.choice([1, 3, 3]); s = rng.choice([1, 1, 2]) if k == 3 else 1
    nsrc = rng.choice([1, 1, 1, 2, 3])
    cins = [rng.choice([3, 8, 16, 24, 40, 64, 96, 128, 160]) for _ in range(nsrc)]
    cout = rng.choice([8, 12, 16, 32, 40, 64, 96, 128, 192, 256])
    h, w = rng.randint(1, 23) * s, rng.randint(1, 23) * s
    B = rng.randint(1, 3)
    act = rng.choice([abi.LP_ACT_NONE, abi.LP_ACT_RELU, abi.LP_ACT_SILU])
    use_res = rng.random() < 0.3 and s == 1
    sl = 5
    eng = Engine(dtype, 'cuda:0'); eng.autotune = False
    srcs = [eng.tensor(c, sl) for c in cins]
    g = torch.Generator().manual_seed(case)
    cin = sum(cins)
    wt = torch.randn(cout, cin, k, k, generator=g) * (2.0 / (cin * k * k)) ** 0.5
    bias = torch.randn(cout, generator=g) * 0.5
    res_id = eng.tensor(cout, sl + (1 if s == 2 else 0)) if use_res else None
    dst = eng.conv(srcs, wt, bias, k, s, act, sl, res=res_id, alpha=0.6)
    eng.finish(); eng.bind(B, h << sl, w << sl)
    q = lambda t: t.to(dtype).float()
    xs = [torch.randn(B, c, h, w, generator=g) for c in cins]
    for t, x in zip(srcs, xs):
        eng.tensor_view(t).copy_(x.to('cuda:0', dtype))
    res = torch.randn(B, cout, h // s, w // s, generator=g) if use_res else None
    if use_res:
        eng.tensor_view(res_id).copy_(res.to('cuda:0', dtype))
    ref = F.conv2d(torch.cat([q(x) for x in xs], 1), q(wt), bias, stride=s, padding=k // 2)
    ref = {abi.LP_ACT_NONE: lambda t: t, abi.LP_ACT_RELU: F.relu, abi.LP_ACT_SILU: F.silu}[act](ref)
    if use_res:
        ref = q(ref) + 0.6 * q(res) if dtype != torch.float32 else ref + 0.6 * res
    x0 = torch.zeros(B, 3, h << sl, w << sl, device='cuda:0', dtype=dtype)
    op = eng.lib.lp_engine_num_ops(eng.h) - 1
    base = None
    for cfg in list(range(8)) + [16, 17]:
        for nb in (1, 2):
            try:
                eng.set_variant(op, cfg, nb)
            except RuntimeError:
                continue
            eng.tensor_view(dst).zero_()
            eng.forward(x0)
            out = eng.tensor_view(dst).clone()
            err = float((out.float().cpu() - ref).abs().max() / max(float(ref.abs().max()), 1e-6))
            same = base is None or torch.equal(out, base)
            if base is None:
                base = out
            if err > TOL[dtype] or not same or not torch.isfinite(out).all():
                bad += 1
                print('CASE %d FAIL dtype=%s cins=%s cout=%d k%d s%d %dx%d B%d act%d res%d variant (%d,%d): rel err %.2e same=%s' %
                      (case, dtype, cins, cout, k, s, h, w, B, act, use_res, cfg, nb, err, same), flush=True)
    if case % 25 == 24:
        print('... %d cases done, %d failures' % (case + 1, bad), flush=True)
print('done: %d cases, %d failing (case, variant) pairs' % (ncases, bad))
