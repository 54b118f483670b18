"""CPU-only checks of the host side: the C-ABI library loads and exports every symbol the header declares,
the engine graph of every BASELINE config builds and packs (host code only, no kernel is launched), argument
validation of the ABI, and the sharding / all-gather logic over gloo with world_size 2."""
import ctypes
import os
import re
import subprocess
import sys

import pytest
import torch

from conftest import REPO


def test_abi_exports_every_declared_symbol():
    from yolov6.hip import abi
    lib = abi.load()
    header = open(os.path.join(REPO, 'include', 'lp_hip.h')).read()
    declared = set(re.findall(r'\b(lp_[a-z0-9_]+)\s*\(', header))
    assert declared == set(abi.SYMBOLS), declared ^ set(abi.SYMBOLS)
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.lp_version().startswith(b'yololp-hip')


def test_abi_argument_validation():
    from yolov6.hip import abi
    lib = abi.load()
    h = ctypes.c_void_p()
    assert lib.lp_engine_create(ctypes.byref(h), 7) < 0
    assert lib.lp_engine_create(ctypes.byref(h), abi.LP_F16) == 0
    assert lib.lp_engine_tensor(h, 0, 0) < 0 and b'bad shape' in lib.lp_last_error()
    t0 = lib.lp_engine_tensor(h, 3, 0)
    t1 = lib.lp_engine_tensor(h, 16, 1)
    assert lib.lp_engine_finalize(h, 3) < 0                          # no input op yet
    assert lib.lp_engine_add_input(h, t1) < 0                        # must be 3 channels at full resolution
    assert lib.lp_engine_add_input(h, t0) == 0
    assert lib.lp_engine_add_input(h, t0) < 0                        # only once, first
    d = abi.ConvDesc()
    d.n_src, d.dst, d.ksize, d.stride, d.act, d.res = 1, t1, 5, 1, 0, -1
    d.src[0] = t0
    w = (ctypes.c_float * (16 * 3 * 25))()
    b = (ctypes.c_float * 16)()
    d.weight, d.bias = ctypes.cast(w, ctypes.c_void_p), ctypes.cast(b, ctypes.c_void_p)
    assert lib.lp_engine_add_conv(h, ctypes.byref(d)) == -4         # 5x5: unsupported
    d.ksize, d.stride = 3, 1
    assert lib.lp_engine_add_conv(h, ctypes.byref(d)) < 0            # stride 1 cannot halve the resolution
    d.stride = 2
    assert lib.lp_engine_add_conv(h, ctypes.byref(d)) == 0
    assert lib.lp_engine_finalize(h, 3) == 0
    assert lib.lp_engine_tensor(h, 8, 1) < 0                         # frozen
    assert lib.lp_engine_weight_bytes(h) > 0
    assert lib.lp_engine_arena_bytes(h, 1, 100, 64) == 0             # not a multiple of 32
    assert lib.lp_engine_arena_bytes(h, 2, 64, 96) > 0
    assert lib.lp_engine_forward(h, None, 0, None, None) < 0         # nothing uploaded / bound
    lib.lp_engine_destroy(h)
    assert lib.lp_nms_workspace_bytes(32, 8400) > 32 * 8400 * 28 * 4
    assert lib.lp_nms(None, 1, 1, 0.4, 0.45, 10, None, None, None, None, 0, None) < 0


@pytest.mark.parametrize('name,n_ops,weight_mb', [('yololps', 69, 37.3), ('yololpn', 69, 9.4), ('yolov6m', 99, 70.0)])
def test_engine_graph_builds_for_baseline_configs(name, n_ops, weight_mb):
    from yolov6.hip.runtime import Engine
    from yolov6.utils.synth import build_synthetic
    m = build_synthetic(os.path.join(REPO, 'configs', name + '.py'))
    eng = Engine.from_model(m, torch.float16, 'cpu')                 # host part only: graph + weight packing
    assert eng.lib.lp_engine_num_ops(eng.h) == n_ops
    # packed fp16 weights ~ the reference's fused parameter bytes (37.26 / 9.39 / 69.99 MB) + padding
    assert weight_mb <= eng.weight_bytes / 1e6 <= weight_mb * 1.05


def test_shard_bounds_cover_the_batch():
    from yolov6.core.sharded import shard_bounds
    for B in (1, 7, 32, 256):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(B, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == B
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


_WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
from yolov6.core.sharded import gather_detections, shard_bounds, unpad
rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
dist.init_process_group('gloo', init_method='tcp://127.0.0.1:%%s' %% os.environ['MASTER_PORT'], rank=rank, world_size=world)
B, max_det = int(os.environ.get('LP_TEST_B', '6')), 5
g = torch.Generator().manual_seed(0)
det_full = torch.rand(B, max_det, 28, generator=g)
cnt_full = torch.randint(0, max_det + 1, (B,), generator=g, dtype=torch.int32)
lo, hi = shard_bounds(B, rank, world)
det_all, cnt_all = gather_detections(det_full[lo:hi].clone(), cnt_full[lo:hi].clone(), global_batch=B)
assert torch.equal(det_all, det_full) and torch.equal(cnt_all, cnt_full), rank
outs = unpad(det_all, cnt_all)
assert [len(o) for o in outs] == cnt_full.tolist()
dist.barrier()
dist.destroy_process_group()
print('rank', rank, 'ok')
'''


@pytest.mark.parametrize('world,B', [(2, 6), (3, 7), (2, 1)], ids=['world2-even', 'world3-uneven', 'world2-empty-shard'])
def test_gather_detections_gloo(tmp_path, world, B):
    """The one exchange step of the sharded path over gloo: equal shards, shards that differ by one image (B % world != 0)
    and a rank with an empty shard."""
    script = tmp_path / 'worker.py'
    script.write_text(_WORKER % REPO)
    port = str(29500 + (os.getpid() * 7 + world * 13 + B) % 2000)
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1', MASTER_PORT=port, LP_TEST_B=str(B))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT))
    for p in procs:
        out, _ = p.communicate(timeout=180)
        assert p.returncode == 0, out.decode()


_WORKER_BAD = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
from yolov6.core.sharded import gather_detections, ShardSizeError
rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
dist.init_process_group('gloo', init_method='tcp://127.0.0.1:%%s' %% os.environ['MASTER_PORT'], rank=rank, world_size=world)
mode = os.environ['LP_TEST_MODE']
max_det = 4
b = 3 + (1 if rank == 1 else 0)            # rank 1 holds one image too many
det = torch.full((b, max_det, 28), float(rank))
cnt = torch.full((b,), rank + 1, dtype=torch.int32)
raised = False
try:
    if mode == 'global':
        gather_detections(det, cnt, global_batch=3 * world)
    elif mode == 'equal':
        gather_detections(det, cnt)
    else:                                    # deferred check: gathers without reading the sizes, unpad() raises
        from yolov6.core.sharded import unpad
        d, c = gather_detections(det, cnt, global_batch=3 * world, check=False) if rank != 1 else (None, None)
        if rank == 1:
            gather_detections(det, cnt, global_batch=3 * world, check=False)
        unpad(d, c)
except ShardSizeError as e:
    raised = True
    print('rank', rank, 'raised:', e)
assert raised, 'rank %%d did not see the size mismatch' %% rank
dist.barrier()                               # every rank is still in step: nobody is stranded in a collective
dist.destroy_process_group()
print('rank', rank, 'ok')
'''


@pytest.mark.parametrize('mode', ['global', 'equal', 'deferred'])
def test_gather_detections_shard_mismatch_raises_on_every_rank(tmp_path, mode):
    """a rank whose shard has the wrong size must not strand the others in the all-gather.  World 2
    over gloo, rank 1 holds one image too many: with ``global_batch`` (sizes ride in the count collective), without it (sizes
    exchanged first) and with the deferred check (``unpad`` reads the sizes); every rank raises ShardSizeError and both reach
    the barrier behind it."""
    script = tmp_path / 'worker_bad.py'
    script.write_text(_WORKER_BAD % REPO)
    port = str(29500 + (os.getpid() * 11 + len(mode) * 17) % 2000)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE='2', MASTER_ADDR='127.0.0.1', MASTER_PORT=port, LP_TEST_MODE=mode)
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for p in procs:
        try:
            out, _ = p.communicate(timeout=120)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise AssertionError('a rank hung in the collective (mode %s)' % mode)
        assert p.returncode == 0, out.decode()


_LISTING = """
\t.text
_Z6kernelv: ; @_Z6kernelv
\ts_load_dword s4, s[0:1], 0x0
\ts_waitcnt lgkmcnt(0)
.LBB0_1:
\t;;#ASMSTART
\tds_read_b128 v[4:7], v0 offset:0
\t;;#ASMEND
\t;;#ASMSTART
\tds_read_b128 v[8:11], v0 offset:16
\t;;#ASMEND
%s
\t;;#ASMSTART
\ts_waitcnt lgkmcnt(1)
\t;;#ASMEND
\tv_mfma_f32_32x32x16_f16 v[16:31], v[4:7], v[4:7], v[16:31]
%s
\ts_cbranch_scc1 .LBB0_1
\ts_endpgm
.Lfunc_end0:
"""


def test_asm_audit_finds_the_hazards_it_is_meant_for(tmp_path):
    """tools/asm_audit.py on synthetic listings: a clean hand-counted read pipeline passes; a compiler copy of a register whose
    LDS read is in flight, an MFMA on a fragment the counted wait does not cover, and a counted wait with a scalar load pending
    are each reported (the hazard classes behind VERDICT r3 'What's weak' 1)."""
    sys.path.insert(0, os.path.join(REPO, 'tools'))
    import asm_audit

    def run(mid, tail):
        f = tmp_path / 'k.s'
        f.write_text(_LISTING % (mid, tail))
        (name, (n, v)), = asm_audit.audit_file(str(f)).items()
        assert name == '_Z6kernelv' and n >= 6
        return v
    assert run('', '\ts_waitcnt lgkmcnt(0)') == []
    v = run('\tv_mov_b32_e32 v12, v9', '\ts_waitcnt lgkmcnt(0)')                 # copy of an in-flight fragment register
    assert len(v) == 1 and 'v9' in v[0]
    v = run('', '\tv_mfma_f32_32x32x16_f16 v[16:31], v[8:11], v[8:11], v[16:31]\n\ts_waitcnt lgkmcnt(0)')   # second read never waited for
    assert len(v) == 1 and 'v8' in v[0]
    v = run('\ts_load_dword s5, s[0:1], 0x4', '\ts_waitcnt lgkmcnt(0)')            # SMEM returns out of order: lgkmcnt(1) proves nothing
    assert any('scalar load' in x for x in v)
    v = run('', '')                                                                # the loop edge carries the second read into the next pass
    assert any('v8' in x or 'v9' in x for x in v)


def test_hand_scheduled_kernels_of_the_product_build_pass_the_asm_audit():
    """Every kernel of libyololp_hip.so that waits for inline-asm LDS reads with hand-counted lgkmcnt is disassembled from the
    shipped binary and audited along all control-flow paths: no instruction may touch a register whose LDS read can still be in
    flight, no counted wait may be issued with a scalar load pending."""
    sys.path.insert(0, os.path.join(REPO, 'tools'))
    import asm_audit
    from yolov6.hip import abi
    pats = ('conv3x3_pipe', 'conv3x3_s2p16', 'stem2_fused_kernel', 'pw_s2_fused_kernel', 'stem_planar_kernel', 'head_det_kernel', 'head_cls_rows_kernel',
            'head_box_det_kernel')
    res = asm_audit.audit_file(abi.LIB_PATH, pats)
    assert sum('conv3x3_pipe' in k for k in res) >= 16 and sum('conv3x3_s2p16' in k for k in res) == 4 and len(res) >= 54, sorted(res)[:5]
    bad = {k: v[:3] for k, (n, v) in res.items() if v}
    assert not bad, bad


def test_block_tile_planning_respects_the_kernels_limits():
    """lp_plan_block_tile: the tiles the engine hands to conv3x3_pipe16v_kernel (stride 1) and conv3x3_s2p16_kernel (stride 2) stay inside
    the variant's pixel blocks and halo slot, and for the layers of the 640x640 network (32 images) they fill the 256 workgroups in whole
    rounds -- the reason these kernels exist."""
    from yolov6.hip import abi
    lib = abi.load()
    th, tw, hp = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    limits = {abi.LP_VARIANT_PIPE16_V0: (448, 512, 1), abi.LP_VARIANT_PIPE16_V1: (224, 512, 1),
              abi.LP_VARIANT_PIPE16_S2A: (256, 864, 2), abi.LP_VARIANT_PIPE16_S2B: (224, 864, 2)}
    for variant, (pb, hpmax, s) in limits.items():
        for ho, wo, B, nct in [(40, 40, 32, 2), (80, 80, 32, 1), (20, 20, 32, 4), (160, 160, 32, 1), (17, 11, 3, 2), (1, 1, 1, 1), (5, 252, 7, 1), (160, 160, 8, 3)]:
            seen = set()
            for choice in range(3):
                assert lib.lp_plan_block_tile(variant, ho, wo, B, nct, choice, ctypes.byref(th), ctypes.byref(tw), ctypes.byref(hp)) == 0
                t = (th.value, tw.value)
                assert 1 <= t[0] <= ho and 1 <= t[1] <= wo and t[0] * t[1] <= pb, (variant, ho, wo, t)
                assert hp.value == (t[1] - 1) * s + 3 and ((t[0] - 1) * s + 3) * hp.value <= hpmax, (variant, ho, wo, t)
                seen.add(t)
            assert len(seen) >= 1
    # whole rounds of 256 workgroups on the layers these variants were built for (first choice)
    for variant, ho, wo, nct, rounds in [(abi.LP_VARIANT_PIPE16_V0, 40, 40, 2, 1), (abi.LP_VARIANT_PIPE16_V0, 80, 80, 1, 2), (abi.LP_VARIANT_PIPE16_V1, 20, 20, 4, 1),
                                         (abi.LP_VARIANT_PIPE16_S2A, 40, 40, 2, 2), (abi.LP_VARIANT_PIPE16_S2A, 20, 20, 4, 1), (abi.LP_VARIANT_PIPE16_S2A, 80, 80, 1, 5)]:
        assert lib.lp_plan_block_tile(variant, ho, wo, 32, nct, 0, ctypes.byref(th), ctypes.byref(tw), None) == 0
        tiles = 32 * -(-ho // th.value) * -(-wo // tw.value) * nct
        assert tiles == rounds * 256, (variant, ho, wo, th.value, tw.value, tiles)
    assert lib.lp_plan_block_tile(abi.LP_VARIANT_PIPE_D, 40, 40, 32, 2, 0, ctypes.byref(th), ctypes.byref(tw), None) < 0
    assert lib.lp_plan_block_tile(abi.LP_VARIANT_PIPE16_V0, 0, 40, 32, 2, 0, ctypes.byref(th), ctypes.byref(tw), None) < 0


def test_engine_cache_stays_out_of_the_module_state():
    """the cached engine must not ride along in ``Model.__dict__`` -- the reference's checkpoint format pickles
    whole modules and EMA / get_model_info deep-copy them (checkpoint.py:22-32)."""
    import copy
    import io
    import pickle
    from yolov6.hip import runtime
    from yolov6.utils.synth import build_synthetic
    m = build_synthetic(os.path.join(REPO, 'configs', 'yololps.py'), width=0.0625)

    class FakeEngine:                       # stands in for a built engine (no GPU here); real engines refuse pickling
        def __reduce__(self):
            raise TypeError('not picklable')
    runtime._engines[m] = (('key',), FakeEngine())
    assert '_lp_engine' not in m.__dict__
    m2 = copy.deepcopy(m)
    buf = io.BytesIO()
    torch.save({'model': m, 'ema': None}, buf)
    pickle.dumps(m)
    assert m2 not in runtime._engines and m in runtime._engines
    m.float()                                # Module._apply drops the stale engine
    assert m not in runtime._engines
    with pytest.raises(TypeError):
        pickle.dumps(runtime.Engine.__new__(runtime.Engine))


def test_stem_tile_planning_respects_the_kernels_limits():
    """lp_plan_stem_tile: the tiles the engine hands to stem_planar_kernel / stem2_fused_kernel cover the map and stay inside
    the kernels' LDS buffers and alignment rules, for every output size a multiple-of-32 frame can give."""
    from yolov6.hip import abi
    lib = abi.load()
    th, tw, hp = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    for ho, wo in [(320, 320), (160, 160), (16, 16), (8, 8), (48, 64), (640, 640), (24, 200), (320, 8), (80, 112)]:
        for fused in (0, 1):
            seen = set()
            for choice in range(3):
                rc = lib.lp_plan_stem_tile(fused, ho, wo, choice, ctypes.byref(th), ctypes.byref(tw), ctypes.byref(hp))
                if rc != 0:
                    assert choice > 0, (fused, ho, wo)            # there is always a first choice
                    break
                t = (th.value, tw.value)
                assert t not in seen and 1 <= t[0] <= ho
                seen.add(t)
                if fused:
                    assert t[1] % 2 == 0 and t[0] * t[1] <= 128 and hp.value >= 2 * t[1] + 1 and (2 * t[0] + 1) * hp.value <= 640
                    assert ((2 * t[0] + 3) * 6 + 2) * (t[1] // 2 + 2) * 16 <= 21 * 1024
                else:
                    assert t[1] % 4 == 0 and t[0] * t[1] <= 512 and ((t[0] + 2) * 6 + 2) * (t[1] // 4 + 2) * 16 <= 20 * 1024
    assert lib.lp_plan_stem_tile(0, 0, 8, 0, ctypes.byref(th), ctypes.byref(tw), None) < 0
