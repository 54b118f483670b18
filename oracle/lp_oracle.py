"""ORACLE (test infrastructure, not product code) -- CPU restatement of the
YOLO-LP detection forward pass: BN fold / RepVGG re-parameterisation, backbone,
BiFPAN neck, decoupled LP head and anchor-free decode, as plain fp32 torch
functional ops over a *state_dict* (no nn.Module classes of the product are
used).  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this file; the product path never does.

Parity status: PINNED.  ``tests/golden/make_golden.py`` imports the reference
model code from ``/root/reference`` in the build container and stores its
outputs (weights, input, pred, feature maps) as fixtures;
``tests/test_oracle_golden.py`` checks this restatement against them.

The arithmetic of conv / max-pool itself lives in torch (ATen / oneDNN), which
is the same third-party library the reference calls (requirements.txt:4).

Every function cites the reference lines it restates (paths relative to the
reference repo root).
"""
import math

import torch
import torch.nn.functional as F

BN_EPS = 1e-3   # yolov6/utils/torch_utils.py:44 (initialize_weights sets every BatchNorm2d.eps)

# Rounding-aware mode (``forward(..., round_to=torch.float16 / torch.bfloat16)``): what the reference's ``--half`` path
# (inferer.py:46-50: model.half(), frames .half()) computes when every convolution accumulates in fp32 -- weights (after
# the fp32 fold), biases, BottleRep's alpha and every layer output are rounded to the 16-bit type (model.half() rounds every
# parameter); accumulation, the class sigmoids and the box decode are fp32 (anchors / strides are fp32 and promote
# the decode).  That is the arithmetic contract of the HIP
# engine's fp16 / bf16 mode; what still differs from it is the fp32 summation order inside a convolution and the last bit
# of exp / reciprocal in SiLU, each of which can move an activation by one 16-bit ulp when its fp32 value sits at a
# rounding boundary.  None = the plain fp32 restatement.
_RND = None


def _r(t):
    """Round to the 16-bit activation type of the rounding-aware mode (identity in fp32 mode)."""
    return t if _RND is None else t.to(_RND).float()


class Arch:
    """Static description of one model (what configs/*.py + yolo.py:54-67 resolve to)."""

    def __init__(self, name, depth, width, csp=False, csp_e=0.5, cspsppf=True, use_dfl=False, reg_max=0,
                 npro=31, nalp=24, nads=37, p6=False, bifusion=True, csp_neck=None):
        self.name, self.csp, self.csp_e, self.cspsppf = name, csp, csp_e, cspsppf
        self.use_dfl, self.reg_max = use_dfl, reg_max
        self.npro, self.nalp, self.nads = npro, nalp, nads
        self.csp_neck = csp if csp_neck is None else csp_neck      # a CSP neck on a Rep backbone gets csp_e = 0.5 (yolo.py:96-100)
        self.csp_e_neck = csp_e if csp else 0.5
        self.p6, self.bifusion = p6, bifusion      # P6: sixth backbone stage + four head levels; bifusion False: plain PAN neck
        if p6:      # configs/yolov6{n,s,m,l}6.py
            reps = [1, 6, 12, 18, 6, 6, 12, 12, 12, 12, 12, 12]
            chans = [64, 128, 256, 512, 768, 1024, 512, 256, 128, 256, 512, 1024]
        else:
            reps = [1, 6, 12, 18, 6, 12, 12, 12, 12]
            chans = [64, 128, 256, 512, 1024, 256, 128, 128, 256, 256, 512]
        self.repeats = [(max(round(i * depth), 1) if i > 1 else i) for i in reps]      # yolo.py:65
        self.channels = [math.ceil(i * width / 8) * 8 for i in chans]                   # yolo.py:66
        self.no = npro + nalp + 5 * nads + 13                                           # effidehead.py:21


ARCHS = {
    'yololps': dict(depth=0.33, width=0.50),
    'yololpn': dict(depth=0.33, width=0.25),
    'yolov6m': dict(depth=0.60, width=0.75, csp=True, csp_e=2.0 / 3, cspsppf=False, use_dfl=True, reg_max=16),
    # P6 / plain-PAN variants: stock YOLOv6 assemblies with the LP head
    'yolov6s6': dict(depth=0.33, width=0.50, p6=True),
    'yolov6m6': dict(depth=0.60, width=0.75, csp=True, csp_e=2.0 / 3, cspsppf=False, use_dfl=True, reg_max=16, p6=True),
}


def arch(name, width=None, **overrides):
    kw = dict(ARCHS[name])
    if width is not None:
        kw['width'] = width
    kw.update(overrides)
    return Arch(name, **kw)


# --------------------------------------------------------------------------
# weight preparation
# --------------------------------------------------------------------------
def _bn(sd, p):
    return sd[p + '.weight'], sd[p + '.bias'], sd[p + '.running_mean'], sd[p + '.running_var']


def fold_conv_bn(sd, p):
    """Conv/SimConv/Conv_C3 -> (W', b').  torch_utils.py:50-82:
    W' = diag(g/sqrt(eps+var)) W ; b' = beta - g*mu/sqrt(var+eps) (conv has no bias).
    Already-fused state_dicts carry ``conv.bias`` and no bn: passed through."""
    w = sd[p + '.conv.weight']
    if p + '.bn.weight' not in sd:
        return w, sd[p + '.conv.bias']
    g, beta, mu, var = _bn(sd, p + '.bn')
    scale = g / torch.sqrt(BN_EPS + var)
    w2 = torch.mm(torch.diag(scale), w.reshape(w.shape[0], -1)).reshape(w.shape)
    b_conv = sd.get(p + '.conv.bias', torch.zeros(w.shape[0]))
    b2 = torch.mm(torch.diag(scale), b_conv.reshape(-1, 1)).reshape(-1) + (beta - g * mu / torch.sqrt(var + BN_EPS))
    return w2, b2


def fold_repvgg(sd, p):
    """RepVGGBlock -> one 3x3 (W_eq, b_eq).  common.py:268-306:
    W_eq = W3*t3 + pad(W1*t1) + I*tid ; b_eq = sum(beta - mu*g/std), std = sqrt(var+eps)."""
    if p + '.rbr_reparam.weight' in sd:
        return sd[p + '.rbr_reparam.weight'], sd[p + '.rbr_reparam.bias']

    def branch(kernel, bnp):
        g, beta, mu, var = _bn(sd, bnp)
        std = (var + BN_EPS).sqrt()
        return kernel * (g / std).reshape(-1, 1, 1, 1), beta - mu * g / std

    w3, b3 = branch(sd[p + '.rbr_dense.conv.weight'], p + '.rbr_dense.bn')
    w1, b1 = branch(sd[p + '.rbr_1x1.conv.weight'], p + '.rbr_1x1.bn')
    w, b = w3 + F.pad(w1, [1, 1, 1, 1]), b3 + b1
    if p + '.rbr_identity.weight' in sd:
        c = w3.shape[1]
        eye = torch.zeros(c, c, 3, 3)
        eye[torch.arange(c), torch.arange(c), 1, 1] = 1
        wid, bid = branch(eye, p + '.rbr_identity')
        w, b = w + wid, b + bid
    return w, b


# --------------------------------------------------------------------------
# layer ops
# --------------------------------------------------------------------------
def rep(sd, p, x, stride=1):
    """RepVGGBlock deploy forward: ReLU(conv3x3(x)) (common.py:258-259)."""
    w, b = fold_repvgg(sd, p)
    return _r(F.relu(F.conv2d(x, _r(w), _r(b), stride=stride, padding=1)))


def cba(sd, p, x, stride=1, act='relu'):
    """Conv (SiLU) / SimConv, Conv_C3 (ReLU) fused forward (common.py:41-42,65-66,475-476)."""
    w, b = fold_conv_bn(sd, p)
    y = F.conv2d(x, _r(w), _r(b), stride=stride, padding=w.shape[-1] // 2)
    return _r(F.silu(y) if act == 'silu' else F.relu(y))


def rep_stage(sd, p, x, n):
    """RepBlock with RepVGG blocks (common.py:430-434)."""
    x = rep(sd, p + '.conv1', x)
    for i in range(n - 1):
        x = rep(sd, '%s.block.%d' % (p, i), x)
    return x


def bottle_rep(sd, p, x):
    """BottleRep: conv2(conv1(x)) + alpha*x (in==out always on this path; common.py:452-455)."""
    y = rep(sd, p + '.conv2', rep(sd, p + '.conv1', x))     # (16-bit mode: conv2's output is rounded, then the sum is)
    return _r(y + _r(sd[p + '.alpha']) * x)


def bepc3(sd, p, x, n):
    """BepC3: cv3(cat[m(cv1 x), cv2 x]); m = RepBlock of n//2 weighted BottleReps
    (common.py:423-428, 497-501)."""
    y = cba(sd, p + '.cv1', x)
    y = bottle_rep(sd, p + '.m.conv1', y)
    for i in range(n // 2 - 1):
        y = bottle_rep(sd, '%s.m.block.%d' % (p, i), y)
    return cba(sd, p + '.cv3', torch.cat((y, cba(sd, p + '.cv2', x)), 1))


def pool_chain(x):
    """Three chained 5x5 s1 p2 max pools = 5/9/13 windows (common.py:144-146)."""
    y1 = F.max_pool2d(x, 5, 1, 2)
    y2 = F.max_pool2d(y1, 5, 1, 2)
    return [x, y1, y2, F.max_pool2d(y2, 5, 1, 2)]


def sim_cspsppf(sd, p, x):
    """SimCSPSPPF.forward (common.py:139-147)."""
    x1 = cba(sd, p + '.cv4', cba(sd, p + '.cv3', cba(sd, p + '.cv1', x)))
    y0 = cba(sd, p + '.cv2', x)
    y3 = cba(sd, p + '.cv6', cba(sd, p + '.cv5', torch.cat(pool_chain(x1), 1)))
    return cba(sd, p + '.cv7', torch.cat((y0, y3), 1))


def sim_sppf(sd, p, x):
    """SimSPPF.forward (common.py:97-103)."""
    return cba(sd, p + '.cv2', torch.cat(pool_chain(cba(sd, p + '.cv1', x)), 1))


def bifusion(sd, p, x0, x1, x2):
    """BiFusion.forward (common.py:523-527); Transpose = 2x2 s2 deconv with bias (:186-187)."""
    up = _r(F.conv_transpose2d(x0, _r(sd[p + '.upsample.upsample_transpose.weight']),
                               _r(sd[p + '.upsample.upsample_transpose.bias']), stride=2))
    a = cba(sd, p + '.cv1', x1)
    d = cba(sd, p + '.downsample', cba(sd, p + '.cv2', x2), stride=2)
    return cba(sd, p + '.cv3', torch.cat((up, a, d), 1))


# --------------------------------------------------------------------------
# network
# --------------------------------------------------------------------------
def backbone(sd, a, x):
    """EfficientRep(6).forward / CSPBepBackbone(_P6).forward (efficientrep.py:103-117, :229-246, :350-364, :481-497)."""
    def body(p, t, n):
        return bepc3(sd, p, t, n) if a.csp else rep_stage(sd, p, t, n)

    x = rep(sd, 'backbone.stem', x, 2)
    outs = []
    last = 6 if a.p6 else 5
    for i in range(2, last + 1):
        p = 'backbone.ERBlock_%d' % i
        x = body(p + '.1', rep(sd, p + '.0', x, 2), a.repeats[i - 1])
        if i == last:
            x = sim_cspsppf(sd, p + '.2', x) if a.cspsppf else sim_sppf(sd, p + '.2', x)
        outs.append(x)
    return outs if a.bifusion else outs[1:]          # (P2,) P3, P4, P5 (, P6): the plain PAN necks run with fuse_P2=False


def neck(sd, a, feats):
    """The eight necks of reppan.py: BiFusion or transpose-conv + concat on the way down, P5 (three levels out) or P6
    (four); forward passes at :108-128, :214-236, :364-390, :514-541, :635-655, :746-768, :900-928, :1053-1083."""
    def stage(p, t, n):
        return bepc3(sd, 'neck.' + p, t, n) if a.csp_neck else rep_stage(sd, 'neck.' + p, t, n)

    def up(i, t):      # Transpose.forward (common.py:186-187)
        return _r(F.conv_transpose2d(t, _r(sd['neck.upsample%d.upsample_transpose.weight' % i]),
                                     _r(sd['neck.upsample%d.upsample_transpose.bias' % i]), stride=2))

    r = a.repeats
    feats = list(feats)
    nlev = 4 if a.p6 else 3
    base = 6 if a.p6 else 5                      # index of the first neck entry in the repeats list
    names_p = ['Rep_p5', 'Rep_p4', 'Rep_p3'] if a.p6 else ['Rep_p4', 'Rep_p3']
    names_n = ['Rep_n4', 'Rep_n5', 'Rep_n6'] if a.p6 else ['Rep_n3', 'Rep_n4']
    down = ['downsample2', 'downsample1', 'downsample0'] if a.p6 else ['downsample2', 'downsample1']
    x = feats[-1]
    fpn = []
    for k in range(nlev - 1):                    # top-down
        f = cba(sd, 'neck.reduce_layer%d' % k, x)
        fpn.append(f)
        if a.bifusion:       # inputs: reduced map, the backbone level below, and the one below that
            t = bifusion(sd, 'neck.Bifusion%d' % k, f, feats[-2 - k], feats[-3 - k])
        else:
            t = torch.cat([up(k, f), feats[-2 - k]], 1)
        x = stage(names_p[k], t, r[base + k])
    outs = [x]
    for k in range(nlev - 1):                    # bottom-up
        x = stage(names_n[k], torch.cat([cba(sd, 'neck.' + down[k], x, 2), fpn[-1 - k]], 1), r[base + nlev - 1 + k])
        outs.append(x)
    return outs


CLS_HEADS = ('pro', 'alp', 'ad0', 'ad1', 'ad2', 'ad3', 'ad4', 'ad5')


def anchors(shapes, strides=(8, 16, 32, 64)):
    """generate_anchors(is_eval=True, mode='af') (anchor_generator.py:11-31)."""
    pts, st = [], []
    for (h, w), s in zip(shapes, strides):
        gy, gx = torch.meshgrid(torch.arange(h) + 0.5, torch.arange(w) + 0.5, indexing='ij')
        pts.append(torch.stack([gx, gy], -1).float().reshape(-1, 2))
        st.append(torch.full((h * w, 1), float(s)))
    return torch.cat(pts), torch.cat(st)


def decode(reg, cor, pts, st):
    """dist2bbox(...,'xywh'), dist2cor, then *= stride (general.py:29-40, :51-66; effidehead.py:283-286).
    reg [B,N,4] ltrb, cor [B,N,8]."""
    x1y1 = pts - reg[..., :2]
    x2y2 = pts + reg[..., 2:]
    box = torch.cat([(x1y1 + x2y2) / 2, x2y2 - x1y1], -1)
    ax, ay = pts[:, 0:1], pts[:, 1:2]
    d = [cor[..., i:i + 1] for i in range(8)]
    corners = torch.cat([ax - d[0], ay - d[1], ax - d[2], ay + d[3], ax + d[4], ay + d[5], ax + d[6], ay - d[7]], -1)
    return box * st, corners * st


def head(sd, a, feats):
    """Detect.forward eval branch (effidehead.py:214-301) -> [B, N, no] fp32."""
    B = feats[0].shape[0]
    cls_all = [[] for _ in CLS_HEADS]
    reg_all, cor_all = [], []
    for i, f in enumerate(feats):
        l = f.shape[2] * f.shape[3]
        s = cba(sd, 'detect.stems.%d' % i, f, act='silu')
        c = cba(sd, 'detect.cls_convs.%d' % i, s, act='silu')
        for acc, h in zip(cls_all, CLS_HEADS):
            o = F.conv2d(c, _r(sd['detect.%s_preds.%d.weight' % (h, i)]), _r(sd['detect.%s_preds.%d.bias' % (h, i)]))
            acc.append(torch.sigmoid(o).reshape(B, -1, l))
        r = cba(sd, 'detect.reg_convs.%d' % i, s, act='silu')
        reg = F.conv2d(r, _r(sd['detect.reg_preds.%d.weight' % i]), _r(sd['detect.reg_preds.%d.bias' % i]))
        if a.use_dfl:   # effidehead.py:247-249: softmax over the reg_max+1 bins, projected by proj_conv
            reg = reg.reshape(-1, 4, a.reg_max + 1, l).permute(0, 2, 1, 3)
            reg = F.conv2d(F.softmax(reg, dim=1), sd['detect.proj_conv.weight'])
        reg_all.append(reg.reshape(B, 4, l))
        cor = F.conv2d(r, _r(sd['detect.cor_preds.%d.weight' % i]), _r(sd['detect.cor_preds.%d.bias' % i]))
        cor_all.append(cor.reshape(B, 8, l))
    cat = lambda parts: torch.cat(parts, -1).permute(0, 2, 1)   # noqa: E731
    pts, st = anchors([f.shape[2:] for f in feats])
    box, corners = decode(cat(reg_all), cat(cor_all), pts, st)
    ones = torch.ones(B, box.shape[1], 1)
    return torch.cat([box, ones, corners] + [cat(p) for p in cls_all], -1)


def forward(sd, a, x, return_stages=False, round_to=None):
    """Model.forward (yolo.py:32-40): pred [B,N,no] fp32 and the three neck maps.  ``round_to``: the rounding-aware mode
    (see ``_RND``) for the fp16 / bf16 engines; ``sd`` stays the fp32 state_dict (it is folded in fp32, then rounded)."""
    global _RND
    if round_to not in (None, torch.float16, torch.bfloat16):
        raise ValueError('round_to must be None, torch.float16 or torch.bfloat16')
    sd = {k: v.float() for k, v in sd.items()}
    x = x.float()
    prev, _RND = _RND, round_to
    try:
        with torch.no_grad():
            bb = backbone(sd, a, _r(x))
            nk = neck(sd, a, bb)
            pred = head(sd, a, nk)
    finally:
        _RND = prev
    if return_stages:
        return pred, nk, bb
    return pred, nk
