"""Host side of the HIP engine: turns a ``yolov6.models.yolo.Model`` into the op
graph of libyololp_hip.so (folded weights, NHWC tensors, concat-free sources) and
runs forward / NMS through the C ABI on torch-owned device memory and torch's
current stream.

Replaces, on a GPU, ``Model.forward`` (reference yolov6/models/yolo.py:32-40)
and ``non_max_suppression`` (yolov6/utils/nms.py:31-130).  Weight preparation
follows the reference's order ``float -> fuse_model -> switch_to_deploy``
(inferer.py:25-68): modules that are still un-fused are folded on the fly with
the same formulas, without touching the caller's model.
"""
import copy
import ctypes
import os
import weakref

import numpy as np
import torch
import torch.nn as nn

from yolov6.hip import abi
from yolov6.layers import common as L

DET_CROSSOVER = 0.5   # candidate density (candidates / anchors) above which forward + lp_nms beats the detections-only forward
_DT = {torch.float16: abi.LP_F16, torch.bfloat16: abi.LP_BF16, torch.float32: abi.LP_F32}
_TORCH_DT = {v: k for k, v in _DT.items()}
CLS_HEADS = ('pro', 'alp', 'ad0', 'ad1', 'ad2', 'ad3', 'ad4', 'ad5')


def _act_of(module):
    if isinstance(module, nn.ReLU):
        return abi.LP_ACT_RELU
    if isinstance(module, (nn.SiLU, L.SiLU)):
        return abi.LP_ACT_SILU
    if isinstance(module, nn.Identity):
        return abi.LP_ACT_NONE
    raise NotImplementedError('activation %s has no HIP epilogue' % type(module).__name__)


def _f32(t):
    if isinstance(t, np.ndarray):
        return np.ascontiguousarray(t, dtype=np.float32)
    return np.ascontiguousarray(t.detach().float().cpu().numpy())


def _fold_conv_bn(conv, bn):
    """fp32 (weight OIHW, bias) of conv followed by an eval-mode BN, the fuse_conv_and_bn formula
    (torch_utils.py:50-82) evaluated on fp32 copies: the caller's modules are not modified."""
    w = conv.weight.detach().float()
    b = conv.bias.detach().float() if conv.bias is not None else torch.zeros(w.shape[0], device=w.device)
    if bn is not None:
        g, beta = bn.weight.detach().float(), bn.bias.detach().float()
        mu, var = bn.running_mean.float(), bn.running_var.float()
        scale = torch.diag(g.div(torch.sqrt(bn.eps + var)))
        w = torch.mm(scale, w.reshape(w.shape[0], -1)).view(w.shape)
        b = torch.mm(scale, b.reshape(-1, 1)).reshape(-1) + (beta - g.mul(mu).div(torch.sqrt(var + bn.eps)))
    return w, b


def _folded(m):
    """Conv / SimConv / Conv_C3, fused or not."""
    return _fold_conv_bn(m.conv, getattr(m, 'bn', None))


class Engine:
    """One frozen graph + its device buffers for one (model, activation dtype, device)."""

    def __init__(self, dtype, device, mfma16=None):
        """mfma16: None = the library's default (on, unless LP_NO_MFMA16 is set); False = every layer on the 32x32x16 MFMA family;
        True = eligible 3x3 stride-1 layers on v_mfma_f32_16x16x32 (lp_engine_set_mfma16: another fp32 summation order)."""
        self.lib = abi.load()
        self.device = torch.device(device)
        self.dtype = dtype
        self.lp_dtype = _DT[dtype]
        h = ctypes.c_void_p()
        abi.check(self.lib.lp_engine_create(ctypes.byref(h), self.lp_dtype), 'lp_engine_create')
        self.h = h
        if mfma16 is not None:
            abi.check(self.lib.lp_engine_set_mfma16(self.h, 1 if mfma16 else 0), 'lp_engine_set_mfma16')
        self._keep = []            # numpy arrays must outlive the add_* calls
        self.bound = None          # (B, H, W)
        self.arena = None
        self.weights = None
        self.neck_ids = []
        self.tuned = set()         # (B, H, W) shapes whose per-layer kernel variants were autotuned
        self.autotune = os.environ.get('LP_AUTOTUNE', '1') != '0'
        self.max_tuned_shapes = 32   # a directory of oddly sized frames must not pay the tuner for every new shape
        self.graph = False         # hipGraph replay of the forward (set_graph); pred is then a persistent buffer
        self.single_lane = True    # set_single_lane (the library's default; LP_LANES=1: execution lanes on)
        self.fuse_siblings = os.environ.get('LP_NO_SIBLINGS') is None   # sibling layers on one input as one launch (conv_pair)
        self._last_stream = None   # stream of the last forward: a forward on ANOTHER stream waits for it (one arena)
        self._graph_pred = None
        self._graph_x = None       # graph mode: persistent staging copy of the input (fixed address)
        self.det_crossover = DET_CROSSOVER   # `detect`: candidate density above which forward + lp_nms is the faster form
        self.pass_rate = None      # candidates / anchors of the last batch that went through `detect` (None: not known yet)
        self._pass_probe = None    # (pinned int32 [B], event, N): asynchronous read-back of the candidate counts
        self.det_routes = {'det': 0, 'pred': 0}   # how often `detect` took each form (introspection / tests)
        self.input_id = self.tensor(3, 0)
        abi.check(self.lib.lp_engine_add_input(self.h, self.input_id), 'lp_engine_add_input')
        if os.environ.get('LP_LANES'):
            self.set_single_lane(False)

    @classmethod
    def from_model(cls, model, dtype, device):
        eng = cls(dtype, device)
        with torch.no_grad():
            eng._build(model)
        return eng.finish(model.detect.nl)

    def finish(self, n_levels=3):
        """Freeze the graph, pack the weights and (on a GPU) upload them."""
        abi.check(self.lib.lp_engine_finalize(self.h, n_levels), 'lp_engine_finalize')
        self._keep = []
        self.weight_bytes = self.lib.lp_engine_weight_bytes(self.h)
        if self.device.type == 'cuda':
            with torch.cuda.device(self.device):
                self.weights = torch.empty(self.weight_bytes + 256, dtype=torch.uint8, device=self.device)
                abi.check(self.lib.lp_engine_upload(self.h, self._aligned(self.weights), self._stream()),
                          'lp_engine_upload')
        return self

    def __reduce__(self):
        raise TypeError('an lp Engine owns device memory and a native handle and cannot be pickled or deep-copied; '
                        'it is rebuilt on demand (runtime.engine_for)')

    def __del__(self):
        try:
            if getattr(self, 'h', None):
                self.lib.lp_engine_destroy(self.h)
                self.h = None
        except Exception:
            pass

    # -- helpers -------------------------------------------------------------
    @staticmethod
    def _aligned(buf):
        return ctypes.c_void_p((buf.data_ptr() + 255) // 256 * 256)

    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _ptr(self, arr):
        self._keep.append(arr)
        return arr.ctypes.data_as(ctypes.c_void_p)

    def lane(self, k):
        """Ops added from now on run on execution lane ``k`` (0 = the caller's stream, 1 / 2 = side streams)."""
        abi.check(self.lib.lp_engine_set_lane(self.h, int(k)), 'lp_engine_set_lane')

    def tensor(self, channels, sl):
        return abi.check(self.lib.lp_engine_tensor(self.h, int(channels), int(sl)), 'lp_engine_tensor')

    def conv(self, srcs, weight, bias, k, s, act, sl, res=None, alpha=0.0):
        """act(conv(cat(srcs))+b) [+ alpha*res] -> new tensor id; ``sl`` is the sources' log2 stride."""
        w, b = _f32(weight), _f32(bias)
        dst = self.tensor(w.shape[0], sl + (1 if s == 2 else 0))
        d = abi.ConvDesc()
        d.n_src = len(srcs)
        for i in range(abi.LP_MAX_SRC):
            d.src[i] = srcs[i] if i < len(srcs) else -1
        d.dst, d.ksize, d.stride, d.act = dst, k, s, act
        d.res = -1 if res is None else res
        d.res_alpha = float(alpha)
        d.weight, d.bias = self._ptr(w), self._ptr(b)
        d.dst2 = -1
        abi.check(self.lib.lp_engine_add_conv(self.h, ctypes.byref(d)), 'lp_engine_add_conv')
        return dst

    def conv_pair(self, srcs, wb1, wb2, k, s, act, sl):
        """Two sibling layers on the same input (same kernel size, stride, activation) as ONE launch with two destination
        tensors (lp_conv_desc.dst2): their weight rows stacked.  The sums of every output channel are those of the two
        separate layers (a cout tile never mixes rows), so the results are the same bits.  Returns (dst1, dst2)."""
        (w1, b1), (w2, b2) = [(_f32(w), _f32(b)) for w, b in (wb1, wb2)]
        if w1.shape[0] % 8 != 0 or not self.fuse_siblings:
            return (self.conv(srcs, w1, b1, k, s, act, sl), self.conv(srcs, w2, b2, k, s, act, sl))
        w, b = np.ascontiguousarray(np.concatenate([w1, w2], 0)), np.ascontiguousarray(np.concatenate([b1, b2], 0))
        sl_out = sl + (1 if s == 2 else 0)
        dst1, dst2 = self.tensor(w1.shape[0], sl_out), self.tensor(w2.shape[0], sl_out)
        d = abi.ConvDesc()
        d.n_src = len(srcs)
        for i in range(abi.LP_MAX_SRC):
            d.src[i] = srcs[i] if i < len(srcs) else -1
        d.dst, d.ksize, d.stride, d.act = dst1, k, s, act
        d.res, d.res_alpha = -1, 0.0
        d.weight, d.bias = self._ptr(w), self._ptr(b)
        d.dst2 = dst2
        abi.check(self.lib.lp_engine_add_conv(self.h, ctypes.byref(d)), 'lp_engine_add_conv')
        return dst1, dst2

    def cba_pair(self, m1, m2, srcs, sl):
        """Two Conv / SimConv / Conv_C3 modules reading the same input: one launch when their shapes allow it."""
        c1, c2 = m1.conv, m2.conv
        same = (c1.kernel_size == c2.kernel_size and c1.stride == c2.stride and _act_of(m1.act) == _act_of(m2.act)
                and c1.in_channels == c2.in_channels)
        if not same:
            return self.cba(m1, srcs, sl), self.cba(m2, srcs, sl)
        return self.conv_pair(srcs, _folded(m1), _folded(m2), c1.kernel_size[0], c1.stride[0], _act_of(m1.act), sl)

    # -- module -> ops ---------------------------------------------------------
    def cba(self, m, srcs, sl):
        """Conv / SimConv / Conv_C3 (common.py:21-66, 466-476)."""
        w, b = _folded(m)
        k, s = m.conv.kernel_size[0], m.conv.stride[0]
        return self.conv(srcs, w, b, k, s, _act_of(m.act), sl)

    def basic(self, m, srcs, sl, res=None, alpha=0.0):
        """A 'basic block' of the rep-style stages -> one conv op."""
        if isinstance(m, L.RepVGGBlock):
            if hasattr(m, 'rbr_reparam'):
                w, b, s = m.rbr_reparam.weight, m.rbr_reparam.bias, m.rbr_reparam.stride[0]
            else:
                mf = m if next(m.parameters()).dtype == torch.float32 else copy.deepcopy(m).float()
                w, b = mf.get_equivalent_kernel_bias()
                s = m.rbr_dense.conv.stride[0]
            return self.conv(srcs, w, b, 3, s, abi.LP_ACT_RELU, sl, res, alpha)
        if isinstance(m, (L.ConvWrapper, L.SimConvWrapper)):
            w, b = _folded(m.block)
            c = m.block.conv
            return self.conv(srcs, w, b, c.kernel_size[0], c.stride[0], _act_of(m.block.act), sl, res, alpha)
        if isinstance(m, L.RealVGGBlock):
            w, b = _fold_conv_bn(m.conv, m.bn)
            return self.conv(srcs, w, b, 3, m.conv.stride[0], abi.LP_ACT_RELU, sl, res, alpha)
        raise NotImplementedError('%s has no HIP lowering' % type(m).__name__)

    def stage_block(self, m, srcs, sl):
        """One element of a RepBlock: a basic block or a BottleRep (common.py:437-455)."""
        if isinstance(m, L.BottleRep):
            assert len(srcs) == 1 or not m.shortcut
            y = self.basic(m.conv1, srcs, sl)
            if m.shortcut:
                alpha = float(m.alpha) if not torch.is_tensor(m.alpha) else float(m.alpha.detach().float().item())
                return self.basic(m.conv2, [y], sl, res=srcs[0], alpha=alpha)
            return self.basic(m.conv2, [y], sl)
        return self.basic(m, srcs, sl)

    def rep_block(self, m, srcs, sl):
        """RepBlock (common.py:416-434)."""
        x = self.stage_block(m.conv1, srcs, sl)
        if m.block is not None:
            for blk in m.block:
                x = self.stage_block(blk, [x], sl)
        return x

    def bepc3(self, m, srcs, sl):
        """BepC3 (common.py:479-501): cv3 reads [m(cv1 x), cv2 x] as two sources."""
        if m.concat is True:                              # cv1 and the shortcut cv2 read the same input: one launch
            c1, c2 = self.cba_pair(m.cv1, m.cv2, srcs, sl)
            return self.cba(m.cv3, [self.rep_block(m.m, [c1], sl), c2], sl)
        a = self.rep_block(m.m, [self.cba(m.cv1, srcs, sl)], sl)
        return self.cba(m.cv3, [a], sl)

    def stage(self, m, srcs, sl):
        if isinstance(m, L.RepBlock):
            return self.rep_block(m, srcs, sl)
        if isinstance(m, L.BepC3):
            return self.bepc3(m, srcs, sl)
        raise NotImplementedError('%s has no HIP lowering' % type(m).__name__)

    def pools(self, x, sl, c):
        ids = [self.tensor(c, sl) for _ in range(3)]
        abi.check(self.lib.lp_engine_add_pool5_chain(self.h, x, *ids), 'lp_engine_add_pool5_chain')
        return ids

    def merge_layer(self, m, x, sl):
        """SimCSPSPPF / CSPSPPF (common.py:124-172) or SimSPPF / SPPF (:88-121); concats are multi-source reads."""
        if isinstance(m, L._CSPSPPFBase):
            c1, y0 = self.cba_pair(m.cv1, m.cv2, [x], sl)         # the trunk's first layer and the CSP shortcut read the same input
            x1 = self.cba(m.cv4, [self.cba(m.cv3, [c1], sl)], sl)
            y3 = self.cba(m.cv6, [self.cba(m.cv5, [x1] + self.pools(x1, sl, m.cv4.conv.out_channels), sl)], sl)
            return self.cba(m.cv7, [y0, y3], sl)
        if isinstance(m, L._SPPFBase):
            x1 = self.cba(m.cv1, [x], sl)
            return self.cba(m.cv2, [x1] + self.pools(x1, sl, m.cv1.conv.out_channels), sl)
        raise NotImplementedError('%s has no HIP lowering' % type(m).__name__)

    def bifusion(self, m, x0, sl0, x1, x2):
        """BiFusion (common.py:504-527): x0 at stride sl0 is upsampled 2x; x1 is at sl0-1; x2 at sl0-2."""
        t = m.upsample.upsample_transpose
        up = self.tensor(t.out_channels, sl0 - 1)
        abi.check(self.lib.lp_engine_add_deconv2x2(self.h, x0, up, self._ptr(_f32(t.weight)), self._ptr(_f32(t.bias))),
                  'lp_engine_add_deconv2x2')
        self.lane(1)                                      # the three inputs of cv3 are independent branches
        a = self.cba(m.cv1, [x1], sl0 - 1)
        self.lane(2)
        d = self.cba(m.downsample, [self.cba(m.cv2, [x2], sl0 - 2)], sl0 - 2)
        self.lane(0)
        return self.cba(m.cv3, [up, a, d], sl0 - 1)

    def upsample(self, m, x, sl):
        """Transpose (common.py:174-187): 2x2 stride-2 transposed conv of a map at stride level ``sl``."""
        t = m.upsample_transpose
        up = self.tensor(t.out_channels, sl - 1)
        abi.check(self.lib.lp_engine_add_deconv2x2(self.h, x, up, self._ptr(_f32(t.weight)), self._ptr(_f32(t.bias))),
                  'lp_engine_add_deconv2x2')
        return up

    def _build(self, model):
        bb, nk, det = model.backbone, model.neck, model.detect
        p6 = hasattr(bb, 'ERBlock_6')
        last = 6 if p6 else 5
        x = self.basic(bb.stem, [self.input_id], 0)
        feats = []
        for i in range(2, last + 1):                              # ERBlock_i takes stride 2^(i-1) to 2^i
            st = getattr(bb, 'ERBlock_%d' % i)
            x = self.basic(st[0], [x], i - 1)
            x = self.stage(st[1], [x], i)
            if len(st) > 2:
                x = self.merge_layer(st[2], x, i)
            feats.append(x)
        self.backbone_ops = self.lib.lp_engine_num_ops(self.h)     # ops [0, backbone_ops) are the backbone (bench.py: roofline.backbone_frac)
        # feats[k] is at stride level k + 2; the backbones return P2 only with fuse_P2 (CSPBepBackbone_P6: always)
        bifusion = hasattr(nk, 'Bifusion0')
        has_p2 = bool(getattr(bb, 'fuse_P2', False)) or type(bb).__name__ == 'CSPBepBackbone_P6'
        if bifusion and not has_p2:
            raise NotImplementedError('the BiFusion necks need the P2 output of the backbone (fuse_P2=True)')
        if not bifusion and has_p2:
            raise ValueError('the plain PAN necks take (P3, P4, P5[, P6]): build the backbone with fuse_P2=False')
        nlev = 4 if p6 else 3
        if det.nl != nlev:
            raise ValueError('head with %d levels on a neck with %d outputs' % (det.nl, nlev))
        names_p = ['Rep_p5', 'Rep_p4', 'Rep_p3'] if p6 else ['Rep_p4', 'Rep_p3']
        names_n = ['Rep_n4', 'Rep_n5', 'Rep_n6'] if p6 else ['Rep_n3', 'Rep_n4']
        downs = ['downsample2', 'downsample1', 'downsample0'] if p6 else ['downsample2', 'downsample1']
        x, sl = feats[-1], last                                    # top-down (reppan.py forward passes)
        fpn = []
        for k in range(nlev - 1):
            f = self.cba(getattr(nk, 'reduce_layer%d' % k), [x], sl)
            fpn.append(f)
            if bifusion:
                t = [self.bifusion(getattr(nk, 'Bifusion%d' % k), f, sl, feats[-2 - k], feats[-3 - k])]
            else:                                                  # torch.cat([upsample(f), x_below]) read as two sources
                t = [self.upsample(getattr(nk, 'upsample%d' % k), f, sl), feats[-2 - k]]
            sl -= 1
            x = self.stage(getattr(nk, names_p[k]), t, sl)
        # The heads run behind the neck, level i's two towers on lanes i % 3 and (i + 1) % 3.  (Experiment, LP_HEADS_EARLY=1: each
        # level's head issued right behind the neck layer that feeds it, on lanes 3 / 4 of its own, to run UNDER the rest of the
        # bottom-up path.  Measured same-box: one batch in flight 2.46 ms either way, six in flight 13.8 -> 13.2 k images/s -- the
        # neck's persistent 3x3 kernels hold every CU's LDS, so the head kernels interleave with them instead of filling gaps.)
        heads_last = not os.environ.get('LP_HEADS_EARLY')
        self.neck_ids = [x]
        if not heads_last:
            self._head(det, 0, x, (0, 1) if nlev == 1 else (3, 4))
        for k in range(nlev - 1):                                  # bottom-up
            d = self.cba(getattr(nk, downs[k]), [x], sl)
            sl += 1
            x = self.stage(getattr(nk, names_n[k]), [d, fpn[-1 - k]], sl)
            self.neck_ids.append(x)
            if not heads_last:
                self._head(det, k + 1, x, (0, 1) if k + 2 == nlev else (3, 4))
        if heads_last:
            for i, f in enumerate(self.neck_ids):
                self._head(det, i, f, (i % 3, (i + 1) % 3))
        self.lane(0)

    def _head(self, det, i, f, lanes):
        """Detect's level ``i`` on the neck output ``f`` (effidehead.py:228-245): stem + class tower + the eight class predictors
        on lane ``lanes[0]``, box tower + box / corner predictors on ``lanes[1]``."""
        sl = 3 + i
        self.lane(lanes[0])
        s = self.cba(det.stems[i], [f], sl)
        c, r = self.cba_pair(det.cls_convs[i], det.reg_convs[i], [s], sl)      # the two towers read the stem's output: one launch
        preds = [getattr(det, '%s_preds' % h)[i] for h in CLS_HEADS]
        wc = np.concatenate([_f32(p.weight).reshape(p.out_channels, -1) for p in preds], 0)
        bc = np.concatenate([_f32(p.bias) for p in preds], 0)
        abi.check(self.lib.lp_engine_add_head_cls(self.h, c, i, wc.shape[0], self._ptr(wc), self._ptr(bc)),
                  'lp_engine_add_head_cls')
        self.lane(lanes[1])
        rp, cp = det.reg_preds[i], det.cor_preds[i]
        bins = det.reg_max + 1 if det.use_dfl else 1
        if rp.out_channels != 4 * bins:
            raise NotImplementedError('reg_preds width %d does not match use_dfl/reg_max' % rp.out_channels)
        wb = np.concatenate([_f32(rp.weight).reshape(rp.out_channels, -1), _f32(cp.weight).reshape(8, -1)], 0)
        bbias = np.concatenate([_f32(rp.bias), _f32(cp.bias)], 0)
        proj = self._ptr(_f32(det.proj_conv.weight).reshape(-1)) if bins > 1 else None
        abi.check(self.lib.lp_engine_add_head_box(self.h, r, i, bins, self._ptr(wb), self._ptr(bbias), proj),
                  'lp_engine_add_head_box')
        self.lane(0)
        self.lane(0)

    # -- execution ---------------------------------------------------------------
    def bind(self, B, H, W):
        if self.bound == (B, H, W):
            return
        if H % 32 or W % 32:
            raise ValueError('input height/width must be multiples of 32, got %dx%d' % (H, W))
        need = self.lib.lp_engine_arena_bytes(self.h, B, H, W)
        if need == 0:
            raise ValueError('bad input shape %s: H and W must be positive multiples of the coarsest stride of the model '
                             '(32; 64 with a P6 level), and small enough for the SPPF pool chain, which keeps a whole stride-32 '
                             'map in LDS (up to ~2048x2048 px in f16 / bf16, ~1440x1440 in f32)' % ((B, H, W),))
        if self.arena is None or self.arena.numel() < need + 256:
            self.arena = None
            self.arena = torch.zeros(need + 256, dtype=torch.uint8, device=self.device)
        abi.check(self.lib.lp_engine_bind(self.h, self._aligned(self.arena), need, B, H, W), 'lp_engine_bind')
        self.bound = (B, H, W)
        self.n_anchors = self.lib.lp_engine_num_anchors(self.h)

    def set_graph(self, enable=True):
        """Replay the forward as one hipGraph (launch-bound shapes, e.g. the per-image loop of Inferer).  The prediction
        tensor returned by ``forward`` is then a persistent buffer that the next forward overwrites."""
        self.graph = bool(enable)
        abi.check(self.lib.lp_engine_set_graph(self.h, 1 if enable else 0), 'lp_engine_set_graph')

    def _stream_enter(self):
        """The engine has ONE activation arena: a forward issued on another stream than the previous one (the model's
        one-at-a-time callers and an InflightForward slot share engine 0) first waits for what that stream holds.  (No event
        per forward: an event record is a barrier packet with a system-scope fence, ~10 us at the start of the next forward.)"""
        cur = torch.cuda.current_stream(self.device)
        if self._last_stream is not None and self._last_stream.cuda_stream != cur.cuda_stream:
            cur.wait_stream(self._last_stream)
        return cur

    def _stream_leave(self, cur):
        self._last_stream = cur

    def set_single_lane(self, enable=True):
        """Issue every kernel of a forward on the caller's stream (no side lanes): what several forwards in flight on several
        streams want (lp_engine_set_single_lane), and the default; ``False``: independent branches on side streams (worth
        +1.8 % for one yolov6m 1280x1280 batch at a time, -1 ... -4 % elsewhere: profiles/r03_round_ab.txt)."""
        if bool(enable) != self.single_lane:
            self.single_lane = bool(enable)
            abi.check(self.lib.lp_engine_set_single_lane(self.h, 1 if enable else 0), 'lp_engine_set_single_lane')

    def copy_tuning(self, other):
        """Take over the tuned kernel variants (all shapes) of ``other``, an engine of the same model and dtype."""
        abi.check(self.lib.lp_engine_copy_tuning(self.h, other.h), 'lp_engine_copy_tuning')
        self.tuned = set(other.tuned)

    def set_variant(self, op, cfg, nbuf):
        """Force the kernel variant of conv op ``op`` (see lp_engine_set_op_variant); switches the autotuner off."""
        self.autotune = False
        abi.check(self.lib.lp_engine_set_op_variant(self.h, op, cfg, nbuf), 'lp_engine_set_op_variant')

    def tensor_view(self, tid):
        """Zero-copy [B,C,h,w] view (channels_last strides) of an arena tensor."""
        off, c, cs, h, w = ctypes.c_size_t(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        abi.check(self.lib.lp_engine_tensor_info(self.h, tid, ctypes.byref(off), ctypes.byref(c), ctypes.byref(cs),
                                                 ctypes.byref(h), ctypes.byref(w)), 'lp_engine_tensor_info')
        B = self.bound[0]
        esz = torch.empty(0, dtype=self.dtype).element_size()
        base = (self.arena.data_ptr() + 255) // 256 * 256 - self.arena.data_ptr() + off.value
        n = B * h.value * w.value * cs.value
        flat = self.arena[base:base + n * esz].view(self.dtype)
        return flat.view(B, h.value, w.value, cs.value)[..., :c.value].permute(0, 3, 1, 2)

    def prepare(self, B, H, W, x_dtype=None):
        """Run the one-off kernel-variant tuner for a new shape on a dummy batch, so that callers who time their forwards
        (Evaler / Inferer speed protocol) do not fold the tuner (hundreds of timed launches) into the first timed batch.
        Does nothing for a shape that is already tuned, or that will never be (tuner off, or ``max_tuned_shapes`` reached):
        ``forward`` re-binds such a shape by itself, which is cheap.  No hipGraph is captured here: graphs are keyed on the
        input pointer (lp_engine_forward), so one captured for the dummy tensor could never be replayed; the caller's first
        forward captures its own."""
        shape = (B, H, W)
        if not self.autotune or shape in self.tuned or len(self.tuned) >= self.max_tuned_shapes:
            return
        x = torch.zeros(B, 3, H, W, dtype=x_dtype or self.dtype, device=self.device)
        graph = self.graph
        if graph:
            self.set_graph(False)
        try:
            self.forward(x)
            torch.cuda.current_stream(self.device).synchronize()
        finally:
            if graph:
                self.set_graph(True)

    def forward(self, x):
        if x.dim() != 4 or x.shape[1] != 3:
            raise ValueError('expected [B,3,H,W], got %s' % (tuple(x.shape),))
        if x.dtype not in _DT:
            raise TypeError('unsupported input dtype %s' % x.dtype)
        x = x.contiguous()
        B, _, H, W = x.shape
        with torch.cuda.device(self.device):
            cur = self._stream_enter()
            self.bind(B, H, W)
            if self.graph:      # fixed input and output addresses: one captured graph per shape, never re-captured
                if self._graph_pred is None or self._graph_pred.shape != (B, self.n_anchors, abi.LP_PRED_COLS):
                    self._graph_pred = torch.empty(B, self.n_anchors, abi.LP_PRED_COLS, dtype=torch.float32, device=self.device)
                pred = self._graph_pred
                x = self._stage_for_graph(x)
            else:
                pred = torch.empty(B, self.n_anchors, abi.LP_PRED_COLS, dtype=torch.float32, device=self.device)
            if self.autotune and self.bound not in self.tuned and len(self.tuned) < self.max_tuned_shapes:
                # first batch of this shape: time the kernel variants of every conv layer in place, keep the best
                abi.check(self.lib.lp_engine_autotune(self.h, ctypes.c_void_p(x.data_ptr()), _DT[x.dtype],
                                                      ctypes.c_void_p(pred.data_ptr()), self._stream(), 5),
                          'lp_engine_autotune')
                self.tuned.add(self.bound)
            abi.check(self.lib.lp_engine_forward(self.h, ctypes.c_void_p(x.data_ptr()), _DT[x.dtype],
                                                 ctypes.c_void_p(pred.data_ptr()), self._stream()), 'lp_engine_forward')
            self._stream_leave(cur)
        return pred

    def _stage_for_graph(self, x):
        """Graph mode: the captured graph reads a persistent staging buffer (a frame-sized copy on the caller's stream)."""
        gx = self._graph_x
        if gx is None or gx.shape != x.shape or gx.dtype != x.dtype:
            gx = self._graph_x = torch.empty_like(x)
        if gx.data_ptr() != x.data_ptr():
            gx.copy_(x)
        return gx

    def forward_det(self, x, conf_thres, ws=None):
        """Detections-only forward (lp_engine_forward_det) on the current stream: the head writes NMS candidates into the
        workspace instead of the [B,N,290] prediction tensor.  Returns the handle ``nms_candidates`` takes: (workspace tensor,
        B, N).  ``ws``: a workspace to reuse (the caller orders its previous use before this call); by default the
        workspace of the current (device, stream)."""
        if x.dim() != 4 or x.shape[1] != 3:
            raise ValueError('expected [B,3,H,W], got %s' % (tuple(x.shape),))
        if x.dtype not in _DT:
            raise TypeError('unsupported input dtype %s' % x.dtype)
        if not 0.0 <= conf_thres <= 1.0:
            raise ValueError('conf_thres must be in [0, 1]')
        x = x.contiguous()
        B, _, H, W = x.shape
        with torch.cuda.device(self.device):
            if self.autotune and (B, H, W) not in self.tuned and len(self.tuned) < self.max_tuned_shapes:
                self.forward(x)                                 # first batch of this shape: bind + tune through the plain forward
            cur = self._stream_enter()
            self.bind(B, H, W)
            N = self.n_anchors
            need = self.lib.lp_nms_workspace_bytes(B, N)
            if ws is None:
                key = (self.device, torch.cuda.current_stream(self.device).cuda_stream)
                ws = _nms_ws.get(key)
                if ws is None or ws.numel() < need + 256:
                    ws = _nms_ws[key] = torch.empty(need + 256, dtype=torch.uint8, device=self.device)
            elif ws.numel() < need + 256:
                raise ValueError('workspace too small: %d bytes needed' % (need + 256))
            wsp = ctypes.c_void_p((ws.data_ptr() + 255) // 256 * 256)
            if self.graph:
                x = self._stage_for_graph(x)
            abi.check(self.lib.lp_engine_forward_det(self.h, ctypes.c_void_p(x.data_ptr()), _DT[x.dtype], float(conf_thres), wsp, need,
                                                     self._stream()), 'lp_engine_forward_det')
            self._stream_leave(cur)
        return ws, B, N

    def det_workspace(self, B, H, W):
        """A private workspace for ``forward_det`` on this engine (several batches in flight: one per engine)."""
        with torch.cuda.device(self.device):
            self.bind(B, H, W)
            return torch.empty(self.lib.lp_nms_workspace_bytes(B, self.n_anchors) + 256, dtype=torch.uint8, device=self.device)

    def detect(self, x, conf_thres, iou_thres, max_det, want_keep=False, route=None):
        """``non_max_suppression(Model.forward(x))`` as one call: (det[B,max_det,28], count[B] int32, keep or None).

        Two forms give the same bits (tested): the detections-only forward (lp_engine_forward_det + lp_nms_candidates: the head
        writes NMS candidates, the [B,N,290] prediction tensor is never written) and forward + lp_nms through that tensor.  The
        first wins while few anchors pass the confidence mask (yololps bs=32 at 3.5 %: +5 %; yololpn bs=128 at 17.6 %: +11 %),
        the second when nearly all do (100 %: the 112-byte candidate rows cost more than the tensor they replace, -9 %;
        DESIGN 6.3).  ``route`` None picks by the candidate density of the previous batch (``pass_rate``, read back
        asynchronously: no host sync here) against ``det_crossover``; 'det' / 'pred' force a form."""
        if route is None:
            self._poll_pass_rate()
            route = 'pred' if (self.pass_rate is not None and self.pass_rate > self.det_crossover) else 'det'
        self.det_routes[route] += 1
        if route == 'pred':
            pred = self.forward(x)
            out = nms_padded(pred, conf_thres, iou_thres, max_det, want_keep)
            ws = _nms_ws[(self.device, torch.cuda.current_stream(self.device).cuda_stream)]
            self._probe_pass_rate(ws, pred.shape[0], pred.shape[1])
            return out
        handle = self.forward_det(x, conf_thres)
        out = nms_candidates(handle, iou_thres, max_det, want_keep)
        self._probe_pass_rate(*handle)
        return out

    def _probe_pass_rate(self, ws, B, N):
        """Queue a copy of the workspace's per-image candidate counts to pinned host memory behind the work just enqueued."""
        with torch.cuda.device(self.device):
            base = (ws.data_ptr() + 255) // 256 * 256
            ptr = self.lib.lp_nms_candidate_counts(ctypes.c_void_p(base), B, N)
            off = ptr - ws.data_ptr()
            pr = self._pass_probe
            if pr is None or pr[0].numel() != B:
                pr = (torch.empty(B, dtype=torch.int32).pin_memory(), torch.cuda.Event(), N)
            pr[0].copy_(ws[off:off + 4 * B].view(torch.int32), non_blocking=True)
            pr[1].record(torch.cuda.current_stream(self.device))
            self._pass_probe = (pr[0], pr[1], N)

    def _poll_pass_rate(self):
        pr = self._pass_probe
        if pr is not None and pr[1].query():
            self.pass_rate = float(pr[0].float().mean()) / max(1, pr[2])

    def op_kinds(self):
        """Kind name of every op of the frozen graph, in op order ('input', 'conv', 'deconv', 'pool', 'head_cls', 'head_box'); needs a
        bound arena (``bind`` / a forward)."""
        out = []
        for i in range(self.lib.lp_engine_num_ops(self.h)):
            kind, ks, cin, cout = ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
            fl, by = ctypes.c_double(), ctypes.c_double()
            abi.check(self.lib.lp_engine_op_info(self.h, i, ctypes.byref(kind), ctypes.byref(ks), ctypes.byref(cin),
                                                 ctypes.byref(cout), ctypes.byref(fl), ctypes.byref(by)), 'lp_engine_op_info')
            out.append(('input', 'conv', 'deconv', 'pool', 'head_cls', 'head_box', 'stem')[kind.value])
        return out

    def profile(self, x, reps=3, inner=1):
        """Per-op device milliseconds (hipEvent pairs around ``inner`` back-to-back launches of each op) + op descriptions,
        for bench.py."""
        x = x.contiguous()
        B, _, H, W = x.shape
        with torch.cuda.device(self.device):
            self.bind(B, H, W)
            pred = torch.empty(B, self.n_anchors, abi.LP_PRED_COLS, dtype=torch.float32, device=self.device)
            n = self.lib.lp_engine_num_ops(self.h)
            ms = (ctypes.c_float * n)()
            abi.check(self.lib.lp_engine_profile_ops(self.h, ctypes.c_void_p(x.data_ptr()), _DT[x.dtype],
                                                     ctypes.c_void_p(pred.data_ptr()), self._stream(), ms, reps, inner),
                      'lp_engine_profile_ops')
        ops = []
        for i in range(n):
            kind, ks, cin, cout = ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
            fl, by = ctypes.c_double(), ctypes.c_double()
            abi.check(self.lib.lp_engine_op_info(self.h, i, ctypes.byref(kind), ctypes.byref(ks), ctypes.byref(cin),
                                                 ctypes.byref(cout), ctypes.byref(fl), ctypes.byref(by)), 'lp_engine_op_info')
            cfg, nb = ctypes.c_int(), ctypes.c_int()
            self.lib.lp_engine_op_variant(self.h, i, ctypes.byref(cfg), ctypes.byref(nb))
            ops.append(dict(kind=('input', 'conv', 'deconv', 'pool', 'head_cls', 'head_box', 'stem')[kind.value], ksize=ks.value,
                            cin=cin.value, cout=cout.value, flops=fl.value, bytes=by.value, ms=float(ms[i]),
                            variant='%s%d' % ({16: 'S', 17: 'W', 18: 'R', 32: 'Pd', 33: 'Pb', 34: 'Pf', 35: 'Pc', 36: 'Pp', 37: 'Fz', 38: 'Fp', 45: 'Fb', 39: 'Md', 41: 'Mf', 42: 'V0', 43: 'V1', 48: 'Xa', 49: 'Xb'}.get(cfg.value) or 'ABCDEFGH'[cfg.value], nb.value)))
        # Ops that launch nothing because a fused kernel carries them (the input op and the stem inside stem2_fused_kernel or behind
        # stem_planar_kernel, the 1x1 layer inside pw_s2_fused_kernel) are folded into their carrier's row: their FLOPs and bytes
        # are work of that kernel, and their own row keeps only the note (an empty event pair -- 1.4 us -- is not a 4 000 TFLOP/s launch).
        direct = x.dtype == self.dtype
        for o in ops:
            o['flops_own'], o['bytes_own'] = o['flops'], o['bytes']     # (the layer's own algorithmic figures: what bench.py's roofline sums)
        for j, c in enumerate(ops):
            i = self.lib.lp_engine_op_carrier(self.h, j, 1 if direct else 0)
            if i < 0 or i == j:
                continue
            ops[i]['flops'] += c['flops']
            # the tensor between the layers never reaches memory: the carrier's bytes are its own input-side and output-side ones
            c.update(flops=0.0, bytes=0.0, ms=0.0, carried_by=i, variant=c['variant'] + '>%d' % i)
        return ops


def _weights_version(model):
    return sum(t._version for t in list(model.parameters()) + list(model.buffers()))


# model -> (key, Engine).  Kept OUT of the module's state: an Engine holds a ctypes handle and a CDLL, and nn.Module pickles /
# deep-copies its __dict__ wholesale (the reference's checkpoint format stores whole pickled modules, checkpoint.py:22-32;
# EMA and get_model_info deep-copy the model), which would fail once the model had run on the GPU.
_engines = weakref.WeakKeyDictionary()


def drop_engine(model):
    """Forget the cached engine of ``model`` (its weights were moved, cast or re-fused)."""
    _engines.pop(model, None)


def engine_for(model, dtype=None):
    """Cached engine of a model; rebuilt when the weights were modified in place or the dtype changed."""
    p = next(model.parameters())
    if not p.is_cuda:
        raise RuntimeError('model is not on a GPU')
    dtype = dtype or getattr(model, 'lp_dtype', None) or p.dtype
    if dtype not in _DT:
        raise TypeError('unsupported activation dtype %s' % dtype)
    key = (dtype, p.device, _weights_version(model))
    cached = _engines.get(model)
    if cached is None or cached[0] != key:
        cached = (key, Engine.from_model(model, dtype, p.device))
        _engines[model] = cached
    return cached[1]


def prepare_for(model, shape, x_dtype=None):
    """Untimed set-up (the one-off kernel-variant tuner) of ``model``'s engine for input shape [B,3,H,W]."""
    eng = engine_for(model)
    if bool(getattr(model, 'lp_graph', False)) != eng.graph:
        eng.set_graph(getattr(model, 'lp_graph', False))
    eng.prepare(int(shape[0]), int(shape[2]), int(shape[3]), x_dtype)


def model_forward(model, x):
    # model.lp_graph = True (set by Inferer) switches the engine to hipGraph replay
    """``Model.forward`` on a GPU: [pred[B,N,290] fp32, [f_s8, f_s16, f_s32]].  The feature maps are
    zero-copy channels_last views of the engine's arena (valid until the next forward of this model)."""
    eng = engine_for(model)
    if bool(getattr(model, 'lp_graph', False)) != eng.graph:
        eng.set_graph(getattr(model, 'lp_graph', False))
    pred = eng.forward(x)
    return [pred, [eng.tensor_view(t) for t in eng.neck_ids]]


def nms_candidates(handle, iou_thres, max_det, want_keep=False):
    """Second half of the NMS (lp_nms_candidates: sort, > 30000 cut, greedy suppression, max_det) on the current stream for the
    candidate lists ``Engine.forward_det`` left in its workspace: (det[B,max_det,28], count[B] int32, keep or None)."""
    ws, B, N = handle
    if not 0.0 <= iou_thres <= 1.0:
        raise ValueError('iou_thres must be in [0, 1]')
    lib = abi.load()
    dev = ws.device
    with torch.cuda.device(dev):
        need = lib.lp_nms_workspace_bytes(B, N)
        det = torch.empty(B, max_det, abi.LP_DET_COLS, dtype=torch.float32, device=dev)
        count = torch.empty(B, dtype=torch.int32, device=dev)
        keep = torch.empty(B, max_det, dtype=torch.int32, device=dev) if want_keep else None
        abi.check(lib.lp_nms_candidates(B, N, float(iou_thres), int(max_det), ctypes.c_void_p(det.data_ptr()),
                                        ctypes.c_void_p(count.data_ptr()), ctypes.c_void_p(keep.data_ptr()) if want_keep else None,
                                        ctypes.c_void_p((ws.data_ptr() + 255) // 256 * 256), need,
                                        ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), 'lp_nms_candidates')
    return det, count, keep


def detect_padded(model, x, conf_thres, iou_thres, max_det, want_keep=False, route=None):
    """``Model.forward`` + ``non_max_suppression`` of a GPU model as one call: (det[B,max_det,28], count[B], keep or None);
    see ``Engine.detect`` for the two forms it chooses between.  For callers that only want detections (Inferer, serving)."""
    eng = engine_for(model)
    if bool(getattr(model, 'lp_graph', False)) != eng.graph:
        eng.set_graph(getattr(model, 'lp_graph', False))
    return eng.detect(x, conf_thres, iou_thres, max_det, want_keep, route)


def detect(model, x, conf_thres, iou_thres, max_det, route=None):
    """Reference-shaped result of ``non_max_suppression(model(x)[0], ...)``: list (len B) of [n_i, 28] tensors."""
    det, count, _ = detect_padded(model, x, conf_thres, iou_thres, max_det, route=route)
    return [det[b, :n] for b, n in enumerate(count.cpu().tolist())]


# ---------------------------------------------------------------------------------------------------
_nms_ws = {}


def nms_padded(prediction, conf_thres, iou_thres, max_det, want_keep=False):
    """Batch NMS through lp_nms without a host sync: (det[B,max_det,28], count[B] int32, keep or None)."""
    if not (prediction.is_cuda and prediction.dtype == torch.float32 and prediction.is_contiguous()):
        raise ValueError('prediction must be a contiguous fp32 CUDA tensor')
    B, N, C = prediction.shape
    if C != abi.LP_PRED_COLS:
        raise ValueError('prediction must have %d columns, got %d' % (abi.LP_PRED_COLS, C))
    lib = abi.load()
    dev = prediction.device
    with torch.cuda.device(dev):
        need = lib.lp_nms_workspace_bytes(B, N)
        # one workspace per (device, stream): lp_nms memsets and fills it on the current stream, so two streams must never
        # share one; a regrown buffer is released through the caching allocator, which orders its reuse on this stream
        key = (dev, torch.cuda.current_stream(dev).cuda_stream)
        ws = _nms_ws.get(key)
        if ws is None or ws.numel() < need + 256:
            ws = _nms_ws[key] = torch.empty(need + 256, dtype=torch.uint8, device=dev)
        det = torch.empty(B, max_det, abi.LP_DET_COLS, dtype=torch.float32, device=dev)
        count = torch.empty(B, dtype=torch.int32, device=dev)
        keep = torch.empty(B, max_det, dtype=torch.int32, device=dev) if want_keep else None
        abi.check(lib.lp_nms(ctypes.c_void_p(prediction.data_ptr()), B, N, float(conf_thres), float(iou_thres),
                             int(max_det), ctypes.c_void_p(det.data_ptr()), ctypes.c_void_p(count.data_ptr()),
                             ctypes.c_void_p(keep.data_ptr()) if want_keep else None,
                             ctypes.c_void_p((ws.data_ptr() + 255) // 256 * 256), need,
                             ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), 'lp_nms')
    return det, count, keep


def non_max_suppression(prediction, conf_thres, iou_thres, max_det):
    """Reference-shaped result: list (len B) of [n_i, 28] tensors; one host sync for the whole batch."""
    B, N = prediction.shape[0], prediction.shape[1]
    if B == 0 or N == 0:
        return [torch.zeros((0, 28), device=prediction.device)] * B
    src = prediction
    work = prediction
    if work.dtype != torch.float32 or not work.is_contiguous():
        work = work.float().contiguous()
    det, count, _ = nms_padded(work, conf_thres, iou_thres, max_det)
    if work is not src:
        src.copy_(work)          # keep the reference's in-place obj*cls side effect on the caller's tensor
    counts = count.cpu().tolist()
    return [det[b, :n] for b, n in enumerate(counts)]


# ---------------------------------------------------------------------------------------------------
def preprocess_letterbox(frame_bgr_u8, img_size, stride, dtype):
    """GPU form of Inferer.precess_image: device uint8 [h,w,3] BGR frame -> [3,H,W] RGB /255 tensor of ``dtype``.
    Geometry (ratio, resized size, padding) is the reference's letterbox arithmetic, done on the host."""
    from yolov6.data.data_augment import letterbox_geometry
    if not (frame_bgr_u8.is_cuda and frame_bgr_u8.dtype == torch.uint8 and frame_bgr_u8.dim() == 3 and
            frame_bgr_u8.shape[2] == 3 and frame_bgr_u8.is_contiguous()):
        raise ValueError('frame must be a contiguous uint8 CUDA tensor [h, w, 3]')
    h0, w0 = frame_bgr_u8.shape[:2]
    _, (rw, rh), (top, bottom, left, right), _ = letterbox_geometry((h0, w0), img_size, stride=stride)
    H, W = rh + top + bottom, rw + left + right
    out = torch.empty(3, H, W, dtype=dtype, device=frame_bgr_u8.device)
    with torch.cuda.device(out.device):
        abi.check(abi.load().lp_preprocess_letterbox(ctypes.c_void_p(frame_bgr_u8.data_ptr()), h0, w0,
                                                     ctypes.c_void_p(out.data_ptr()), _DT[dtype], H, W, rh, rw, top, left,
                                                     ctypes.c_void_p(torch.cuda.current_stream(out.device).cuda_stream)),
                  'lp_preprocess_letterbox')
    return out


def rescale_round(ori_shape, det, target_shape):
    """GPU form of ``Inferer.rescale(ori_shape, det[:, :12], target_shape).round()``, in place on det [n, 28]."""
    if not (det.is_cuda and det.dtype == torch.float32 and det.dim() == 2 and det.shape[1] == abi.LP_DET_COLS and
            det.stride(1) == 1 and det.stride(0) == abi.LP_DET_COLS):
        raise ValueError('det must be a CUDA fp32 [n, 28] tensor with contiguous rows')
    ratio = min(ori_shape[0] / target_shape[0], ori_shape[1] / target_shape[1])
    padx, pady = (ori_shape[1] - target_shape[1] * ratio) / 2, (ori_shape[0] - target_shape[0] * ratio) / 2
    with torch.cuda.device(det.device):
        abi.check(abi.load().lp_rescale_round(ctypes.c_void_p(det.data_ptr()), det.shape[0], float(ratio), float(padx),
                                              float(pady), int(target_shape[1]), int(target_shape[0]),
                                              ctypes.c_void_p(torch.cuda.current_stream(det.device).cuda_stream)),
                  'lp_rescale_round')
    return det


def eval_counts(det, det_count, tgt, tgt_count, counts=None):
    """Counters of the LP accuracy metric for one batch (``lp_eval_counts``): det [B,max_det,28] fp32 + det_count [B]
    int32 as ``nms_padded`` returns them, tgt [B,max_t,20] fp32 + tgt_count [B] int32; ``counts`` (int64 [43], CUDA) is
    accumulated into and returned (allocated zeroed when None)."""
    dev = det.device
    ok = (det.is_cuda and det.dtype == torch.float32 and det.dim() == 3 and det.shape[2] == abi.LP_DET_COLS and det.is_contiguous()
          and tgt.dtype == torch.float32 and tgt.dim() == 3 and tgt.shape[2] == 20 and tgt.is_contiguous() and tgt.device == dev
          and det_count.dtype == torch.int32 and tgt_count.dtype == torch.int32 and det_count.is_contiguous() and tgt_count.is_contiguous()
          and det_count.device == dev and tgt_count.device == dev
          and det.shape[0] == tgt.shape[0] == det_count.numel() == tgt_count.numel())
    if not ok:
        raise ValueError('eval_counts: expected contiguous CUDA det [B,D,28] / tgt [B,T,20] fp32 and int32 counts [B] on one device')
    if counts is None:
        counts = torch.zeros(abi.LP_EVAL_NCOUNTS, dtype=torch.int64, device=dev)
    elif not (counts.dtype == torch.int64 and counts.numel() == abi.LP_EVAL_NCOUNTS and counts.device == dev and counts.is_contiguous()):
        raise ValueError('eval_counts: counts must be a contiguous CUDA int64 [%d] tensor' % abi.LP_EVAL_NCOUNTS)
    with torch.cuda.device(dev):
        abi.check(abi.load().lp_eval_counts(ctypes.c_void_p(det.data_ptr()), ctypes.c_void_p(det_count.data_ptr()), det.shape[1],
                                            ctypes.c_void_p(tgt.data_ptr()), ctypes.c_void_p(tgt_count.data_ptr()), tgt.shape[1],
                                            det.shape[0], ctypes.c_void_p(counts.data_ptr()),
                                            ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), 'lp_eval_counts')
    return counts
