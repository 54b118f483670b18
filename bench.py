#!/usr/bin/env python3
"""Benchmark of the YOLO-LP hot path on MI355X: images/s of (forward + decode + NMS [+ all-gather]).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One step = one pass of the hot path over one per-GPU batch of synthetic 640x640 frames already resident in
HBM: Model.forward through the HIP engine, lp_nms, and (N > 1) the RCCL all-gather of the padded detections.
Rank 0 prints ONE JSON line.  Workload at N=1: BASELINE.json configs[1] (yololps 640x640 bs=32 fp16).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

METRIC = 'images/sec (640×640) end-to-end detect+NMS, yololps, 1/2/4/8 MI355X'
PEAK_TFLOPS = {'f16': 2500.0, 'bf16': 2500.0, 'f32': 157.3}     # dense MFMA peaks, MI355X_MICROARCH.md
HBM_BOUND = ('yololpn',)   # configurations whose roofline is HBM bandwidth; the others: dense MFMA
ALG_MB = {'yololpn': (74.66, 9.39)}   # algorithmic MB per 640x640 image, MB of weights per batch
SIGMA = {'yololps': 0.25, 'yololpn': 0.6, 'yolov6m': 0.25, 'yolov6s6': 0.25, 'yolov6m6': 0.25}   # predictor-weight scale of the synthetic recipe (the P6 assemblies are extra configs)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--model', default='yololps', choices=list(SIGMA))
    ap.add_argument('--batch', type=int, default=32, help='images per GPU per step')
    ap.add_argument('--size', type=int, default=640)
    ap.add_argument('--dtype', default='f16', choices=['f16', 'bf16', 'f32'])
    ap.add_argument('--conf', type=float, default=0.4)
    ap.add_argument('--iou', type=float, default=0.45)
    ap.add_argument('--max-det', type=int, default=1000)
    ap.add_argument('--no-overlap', action='store_true', help='run forward and NMS on one stream (default: NMS of step i on a '
                    'second stream under the forward of step i+1)')
    ap.add_argument('--inflight', type=int, default=6, help='batches in flight on the GPU: consecutive steps run on this many '
                    'engines (arenas + streams), see yolov6/core/pipeline.py; 1 = one forward at a time')
    ap.add_argument('--via-pred', action='store_true', help='Model.forward writes the [B,N,290] prediction tensor and lp_nms reads it back '
                    '(the reference-shaped two-call path); default: the detections-only forward, whose head writes NMS candidates')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-seconds', type=float, default=15.0)
    ap.add_argument('--detail', default='', help='write the per-op device-time table to this file')
    ap.add_argument('--single-lane', type=int, default=-1, help='1: every engine issues its kernels on ONE stream (no side lanes for '
                    'independent branches; the default); 0: up to three lanes per forward '
                    '(profiles/r03_inflight_lanes.txt, profiles/r03_round_ab.txt)')
    ap.add_argument('--roofline-file', default='',
                    help='roofline of the 3x3 layers from a rocprofv3 --kernel-trace of this command (tools/roofline_from_trace.py); '
                         'reported as roofline.frac when its kernel-source hash equals that of this build '
                         '(default: profiles/r04_roofline[_<model>_<size>_bs<batch>_<dtype>].json, the suffix for every workload but the default one)')
    ap.add_argument('--traffic-file', default='',
                    help='per-launch HBM bytes of the dominant kernel from a rocprofv3 --pmc run (tools/pmc_traffic.py); used only '
                         'when the kernel-source hash recorded in it equals that of the sources this build was made from '
                         '(default: profiles/r04_pmc_traffic[_<workload>].json)')
    ap.add_argument('--profile-inner', type=int, default=4, help='back-to-back launches of each op per event pair in the per-op '
                    'timing (amortises the cost of the event pair itself)')
    return ap.parse_args()


def host_cores():
    """Threads this process may really use: affinity mask and cgroup quota, capped at the GPU box's 16-core share
    per GPU (os.cpu_count() reports all 256 host threads there and oversubscribing them is ~100x slower)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 16))


def cpu_baseline(model_name, sigma, size, conf, iou, max_det, budget_s):
    """The oracle (CPU restatement of the reference path: torch fp32 forward + C NMS) timed on this host."""
    import torch
    from oracle import lp_oracle, lp_post
    from yolov6.utils.synth import build_synthetic
    m = build_synthetic(os.path.join(ROOT, 'configs', model_name + '.py'), sigma=sigma)
    sd = {k: v.float() for k, v in m.state_dict().items()}
    a = lp_oracle.arch(model_name)
    B = 2
    x = torch.rand(B, 3, size, size, generator=torch.Generator().manual_seed(1234))
    cores = host_cores()
    torch.set_num_threads(cores)

    def step():
        pred, _ = lp_oracle.forward(sd, a, x)
        lp_post.nms_c(pred.numpy(), conf, iou, max_det)

    step()                                             # warm-up (oneDNN primitive creation)
    n, t0 = 0, time.time()
    while True:
        step()
        n += 1
        el = time.time() - t0
        if el >= budget_s or n >= 50:
            break
    return {'value': round(n * B / el, 3), 'unit': 'images/s', 'cores': cores, 'kind': 'port',
            'sample': '%d steps of %d images %dx%d, %s fp32, oracle/lp_oracle.py forward + oracle/lp_post_ref.c NMS, '
                      '%d torch threads' % (n, B, size, size, model_name, cores)}


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    from yolov6.utils.synth import build_synthetic
    from yolov6.hip import runtime
    from yolov6.core.sharded import gather_detections

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d'
                         % (args.gpus, world, args.gpus))
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', device_id=dev)

    tdt = {'f16': torch.float16, 'bf16': torch.bfloat16, 'f32': torch.float32}[args.dtype]
    sigma = SIGMA[args.model]
    model = build_synthetic(os.path.join(ROOT, 'configs', args.model + '.py'), sigma=sigma)
    # the reference's inference preparation order: float -> fuse_model -> switch_to_deploy -> half (inferer.py:25-68)
    from yolov6.utils.torch_utils import fuse_model
    from yolov6.layers.common import RepVGGBlock
    model = fuse_model(model).eval()
    for layer in model.modules():
        if isinstance(layer, RepVGGBlock):
            layer.switch_to_deploy()
    model = model.to(dev).to(tdt)
    B = args.batch
    overlap = not args.no_overlap
    depth = max(1, args.inflight) if overlap else 1
    # one synthetic batch per slot in flight, all resident in HBM before the timed region
    xs = [torch.rand(B, 3, args.size, args.size, generator=torch.Generator().manual_seed(1234 + rank + 1000 * k)).to(dev).to(tdt)
          for k in range(depth)]
    x = xs[0]
    torch.cuda.synchronize(dev)        # the batches are complete in HBM before any side stream reads them
    eng = runtime.engine_for(model)

    gathered = None
    # Streams.  (1) The NMS (+ all-gather) of a step -- bandwidth / latency-bound, one workgroup per image in its last two
    # kernels -- runs on a post stream under the forwards of the following steps.  (2) `depth` batches are in flight:
    # consecutive steps run their forwards on `depth` engines (own arena + streams, yolov6/core/pipeline.py), so that the
    # workgroups of one batch's kernels fill the CU slots another batch's kernels leave empty (at 32 images the 40x40 and
    # 20x20 layers cannot fill 256 CUs evenly).  Every step still runs the whole path on its own batch; pred is a fresh
    # buffer per step, the stages only meet through recorded events.
    s_fwd = torch.cuda.current_stream(dev)
    s_post = torch.cuda.Stream(dev) if overlap else s_fwd
    pipe = None
    if depth > 1:
        from yolov6.core.pipeline import InflightForward
        pipe = InflightForward(model, depth, single_lane=None if args.single_lane < 0 else bool(args.single_lane))
    nstep = [0]

    ws1 = [[None, None], [None, None]]   # one-batch-in-flight path: two [candidate workspace, event of its last NMS], used alternately
    done_events = []        # when not None: one event per step, recorded on the post stream when the step's detections are complete

    def step(pipe=None, depth=1):
        nonlocal gathered
        k = nstep[0] % depth
        nstep[0] += 1
        handle = release = None
        if pipe is not None:
            if args.via_pred:
                pred, ready = pipe.submit(xs[k], fresh=False)      # the synthetic batches were made before the timed region
                pred.record_stream(s_post)
            else:
                handle, ready, release = pipe.submit_det(xs[k], args.conf, fresh=False)
            s_post.wait_event(ready)
        else:
            eng.set_single_lane(args.single_lane != 0)              # (--single-lane 0: independent branches of a forward on side streams)
            if args.via_pred:
                pred = eng.forward(x)
            else:
                slot = ws1[nstep[0] & 1]                          # two candidate workspaces: the NMS of step i runs under forward i + 1
                if slot[0] is None:
                    slot[0] = eng.det_workspace(B, args.size, args.size)
                if slot[1] is not None:
                    s_fwd.wait_event(slot[1])                    # the NMS that last read this workspace (two steps ago) is done
                handle = eng.forward_det(x, args.conf, ws=slot[0])
            if overlap:
                ready = torch.cuda.Event()
                ready.record(s_fwd)
                s_post.wait_event(ready)
                if args.via_pred:
                    pred.record_stream(s_post)
        with torch.cuda.stream(s_post):
            if handle is None:
                det, count, _ = runtime.nms_padded(pred, args.conf, args.iou, args.max_det)
            else:
                det, count, _ = runtime.nms_candidates(handle, args.iou, args.max_det)
                if release is not None:
                    release(s_post)
                else:
                    slot[1] = torch.cuda.Event()
                    slot[1].record(s_post)
            if world > 1:
                gathered = gather_detections(det, count, out=gathered, global_batch=det.shape[0] * world)   # sizes ride in the count collective: no extra exchange, no sync
            if done_events is not None:
                ev = torch.cuda.Event(enable_timing=True)
                ev.record(s_post)
                done_events.append(ev)
            if world > 1:
                return gathered
        return det, count

    def step_stats():
        """Steady-state time per step from the completion events of consecutive steps: median and p10 / p90."""
        ms = sorted(a_.elapsed_time(b_) for a_, b_ in zip(done_events[:-1], done_events[1:]))
        if not ms:
            return None
        q = lambda f: round(ms[min(len(ms) - 1, int(f * len(ms)))], 4)   # noqa: E731
        return {'median': q(0.5), 'p10': q(0.1), 'p90': q(0.9), 'n': len(ms)}

    def join():     # the measuring stream waits for everything enqueued on the other streams
        if pipe is not None:
            for st in pipe.streams:
                s_fwd.wait_stream(st)
        if s_post is not s_fwd:
            s_fwd.wait_stream(s_post)

    def timed(pipe_, depth_):
        """W untimed + K timed steps bracketed by barrier + synchronize on both sides: (seconds, host enqueue seconds, device ms)."""
        nstep[0] = 0
        for _ in range(depth_):  # set-up, not steps of the measurement: every engine binds its arena and tunes its kernel variants
            out_ = step(pipe_, depth_)
        for _ in range(args.warmup):
            out_ = step(pipe_, depth_)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        del done_events[:]
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record()
        for _ in range(args.steps):
            out_ = step(pipe_, depth_)
        join()
        ev1.record()
        t_issue_ = time.perf_counter() - t0         # host time to enqueue all steps
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        return time.perf_counter() - t0, t_issue_, ev0.elapsed_time(ev1), out_

    with torch.no_grad():
        elapsed, t_issue, dev_ms, out = timed(pipe, depth)
        stats = step_stats()
        # the same K steps with ONE batch in flight (a single engine; the NMS of a step still runs on the post stream under
        # the next forward): what a caller gets who hands over one batch at a time
        if depth > 1:
            elapsed1, _, _, _ = timed(None, 1)
            stats1 = step_stats()
        else:
            elapsed1, stats1 = elapsed, stats
    el = torch.tensor([elapsed, elapsed1], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed, elapsed1 = float(el[0].item()), float(el[1].item())
    counts = out[1].float().mean().item()

    result = None
    if rank == 0:
        # per-op device time (hipEvent pairs on torch's current stream, inside lp_engine_profile)
        ops = eng.profile(x, reps=5, inner=max(1, args.profile_inner))
        t_nms = []
        for _ in range(5):
            pred = eng.forward(x)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            runtime.nms_padded(pred, args.conf, args.iou, args.max_det)
            e1.record()
            torch.cuda.synchronize()
            t_nms.append(e0.elapsed_time(e1))
        if args.detail:
            with open(args.detail, 'w') as f:
                for i, o in enumerate(ops):
                    if o.get('carried_by') is not None:
                        f.write('%3d %-8s k%d %4d->%4d %s   (runs inside the kernel of op %d: its FLOPs are counted there)\n'
                                % (i, o['kind'], o['ksize'], o['cin'], o['cout'], o['variant'], o['carried_by']))
                        continue
                    f.write('%3d %-8s k%d %4d->%4d %s %8.1f us  %7.1f TFLOP/s  %7.1f GB/s (algorithmic)\n'
                            % (i, o['kind'], o['ksize'], o['cin'], o['cout'], o['variant'], o['ms'] * 1e3,
                               o['flops'] / max(o['ms'], 1e-9) / 1e9, o['bytes'] / max(o['ms'], 1e-9) / 1e6))
        conv3 = [o for o in ops if o['kind'] == 'conv' and o['ksize'] == 3]
        # the backbone alone (north_star's ">= 70 % MFMA roofline on the conv backbone"): all its ops, SURVEY 8(d)'s algorithmic FLOPs
        nbb = getattr(eng, 'backbone_ops', 0)
        bb_fl, bb_ms = sum(o['flops_own'] for o in ops[:nbb]), sum(o['ms'] for o in ops[:nbb])
        KNAME = {'P': 'conv3x3_pipe_kernel', 'M': 'conv3x3_pipe16_kernel', 'V': 'conv3x3_pipe16v_kernel', 'X': 'conv3x3_s2p16_kernel', 'Fz': 'stem2_fused_kernel (stem + the layer behind it)',
                 'Fp': 'pw_s2_fused_kernel (1x1 + 3x3 stride 2)', 'Pp': 'stem_planar_kernel'}
        names = {}
        for o in conv3:
            v = o['variant']
            n = KNAME.get(v[:2]) or KNAME.get(v[:1]) or 'conv_mfma_kernel<KS=3>'
            names[n] = names.get(n, 0) + 1
        allmm = [o for o in ops if o['kind'] in ('conv', 'deconv', 'head_cls', 'head_box')]
        fl3, ms3 = sum(o['flops_own'] for o in conv3), sum(o['ms'] for o in conv3)
        peak = PEAK_TFLOPS[args.dtype]
        ach = fl3 / (ms3 * 1e-3) / 1e12 if ms3 > 0 else 0.0
        total_ms = sum(o['ms'] for o in ops)
        traffic, traffic_note = None, 'no PMC file for this workload'
        tf_ok = rf_ok = None      # the profile files of this workload, when measured on this build's kernel sources
        default_workload = (args.model, args.batch, args.size, args.dtype) == ('yololps', 32, 640, 'f16')
        suffix = '' if default_workload else '_%s_%d_bs%d_%s' % (args.model, args.size, args.batch, args.dtype)
        args.traffic_file = args.traffic_file or os.path.join(ROOT, 'profiles', 'r04_pmc_traffic%s.json' % suffix)
        args.roofline_file = args.roofline_file or os.path.join(ROOT, 'profiles', 'r04_roofline%s.json' % suffix)
        if os.path.exists(args.traffic_file):     # PMC counters come from a separate rocprofv3 pass
            from yolov6.hip.srchash import source_hash
            tf = json.load(open(args.traffic_file))
            if tf.get('kernel_source_hash') == source_hash():
                tf_ok = tf
                traffic, traffic_note = tf.get('hbm_bytes_per_launch'), os.path.relpath(args.traffic_file, ROOT)
            else:     # counters of another code state would go stale silently: report none
                traffic_note = 'stale: %s was measured on kernel sources %s, this build is %s' % (
                    os.path.relpath(args.traffic_file, ROOT), tf.get('kernel_source_hash'), source_hash())
        roofline = {
            'bound': 'mfma', 'achieved': round(ach, 2), 'peak': peak, 'unit': 'TFLOP/s', 'frac': round(ach / peak, 4),
            'traffic': traffic, 'traffic_source': traffic_note,
            'kernel': '3x3 convolution layers, implicit GEMM, all %d of a step: %s' % (len(conv3), ', '.join('%s x%d' % kv for kv in sorted(names.items(), key=lambda kv: -kv[1]))),
            'backbone_frac': round(bb_fl / (bb_ms * 1e-3) / 1e12 / peak, 4) if bb_ms > 0 else None,
            'backbone': '%d ops, %.3f GFLOP, %.1f us (live hipEvent timing)' % (nbb, bb_fl / 1e9, bb_ms * 1e3),
            'backbone_flops': bb_fl, 'backbone_dispatches': sum(1 for o in ops[:nbb] if o['ms'] * 1e3 >= 3.0),   # (ops carried by a fused kernel launch nothing)
            'timing': 'hipEvent pairs on the launch stream around %d back-to-back launches of each op' % max(1, args.profile_inner),
            'launches': len(conv3), 'avg_launch_ms': round(ms3 / max(1, len(conv3)), 4),
            'conv3_layers': len(conv3), 'conv3_flops_per_step': fl3,
            'flops_per_launch': round(fl3 / max(1, len(conv3)) / 1e9, 3),
            'algorithmic_bytes_per_launch': round(sum(o['bytes_own'] for o in conv3) / max(1, len(conv3))),
            'all_mfma_kernels_tflops': round(sum(o['flops_own'] for o in allmm) / (sum(o['ms'] for o in allmm) * 1e-3) / 1e12, 2),
            'forward_device_ms': round(total_ms, 3), 'nms_device_ms': round(sorted(t_nms)[len(t_nms) // 2], 3),
            'forward_hbm_gbs': round(sum(o['bytes_own'] for o in ops) / (total_ms * 1e-3) / 1e9, 1),
        }
        # The headline fraction comes from the kernel trace (rocprofv3 --kernel-trace of this command, one batch in flight:
        # tools/roofline_from_trace.py -> profiles/) when that file was measured on this build's kernel sources; the live
        # event-timed figure stays beside it as frac_event.
        roofline['frac_event'] = roofline['frac']
        roofline['frac_source'] = 'live hipEvent timing (no kernel-trace roofline file for this workload and build)'
        if os.path.exists(args.roofline_file):
            from yolov6.hip.srchash import source_hash
            rf = json.load(open(args.roofline_file))
            if rf.get('kernel_source_hash') == source_hash():
                rf_ok = rf
                roofline['achieved_event'] = roofline['achieved']
                roofline['achieved'], roofline['frac'] = rf['achieved_tflops'], rf['frac']
                roofline['trace_us_per_step'] = rf['conv3_us_per_step']
                if rf.get('backbone_frac') is not None:
                    roofline['backbone_frac_event'] = roofline['backbone_frac']
                    roofline['backbone_frac'] = rf['backbone_frac']
                roofline['frac_source'] = '%s (rocprofv3 kernel trace, %d timed steps)' % (os.path.relpath(args.roofline_file, ROOT), rf['steps'])
            else:
                roofline['frac_source'] = 'live hipEvent timing (stale: %s was measured on kernel sources %s, this build is %s)' % (
                    os.path.relpath(args.roofline_file, ROOT), rf.get('kernel_source_hash'), source_hash())
        if args.model in HBM_BOUND:
            # bandwidth-bound configuration (AI 153 FLOP/B < machine balance 312): the roofline of the forward is HBM.
            # Algorithmic bytes (every layer reads its input once and writes its output once, weights once per batch) of the
            # whole forward / its device time; peak 8 TB/s (MI355X_MICROARCH.md).
            mb_img, mb_w = ALG_MB[args.model]
            scale = (args.size / 640.0) ** 2
            alg = (mb_img * scale * B + mb_w) * 1e6
            gbs = alg / (total_ms * 1e-3) / 1e9
            roofline.update({
                'bound': 'hbm', 'achieved': round(gbs, 1), 'peak': 8000.0, 'unit': 'GB/s', 'frac': round(gbs / 8000.0, 4),
                'frac_source': 'algorithmic bytes of the forward (SURVEY 8(d): %.2f MB per image + %.2f MB weights) / summed device time of its kernels (live hipEvent timing)' % (mb_img * scale, mb_w),
                'kernel': 'whole forward (%d launches; no single kernel dominates a bandwidth-bound configuration)' % len(ops),
                'algorithmic_bytes_per_launch': round(alg), 'launches': 1, 'avg_launch_ms': round(total_ms, 4),
                'throughput_hbm_frac': round(world * B * args.steps / elapsed * (mb_img * scale) * 1e6 / 8e12 / world, 4),
                'mfma_3x3_tflops': round(ach, 2), 'mfma_3x3_frac': roofline['frac']})
            roofline['frac_event'] = roofline['frac']
            if rf_ok is not None and rf_ok.get('forward_us_per_step'):      # the forward's kernels in the trace (one batch in flight)
                gbs_t = alg / (rf_ok['forward_us_per_step'] * 1e-6) / 1e9
                roofline.update({'achieved_event': roofline['achieved'], 'achieved': round(gbs_t, 1), 'frac': round(gbs_t / 8000.0, 4),
                                 'trace_us_per_step': rf_ok['forward_us_per_step'],
                                 'frac_source': 'algorithmic bytes of the forward / summed kernel-trace duration of its dispatches, %s (%d timed steps)'
                                                % (os.path.relpath(args.roofline_file, ROOT), rf_ok['steps'])})
            if tf_ok is not None and tf_ok.get('forward_hbm_bytes_per_step'):
                roofline['traffic'] = tf_ok['forward_hbm_bytes_per_step']
        result = {
            'metric': METRIC, 'value': round(world * B * args.steps / elapsed, 2), 'unit': 'images/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(elapsed / args.steps * 1e3, 4), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': args.dtype, 'data': 'synthetic',
            'value_inflight1': round(world * B * args.steps / elapsed1, 2),
            'step_ms': stats, 'step_ms_inflight1': stats1,
            'config': {'workload': '%s %dx%d bs=%d/GPU %s, synthetic frames resident in HBM, seeded random-init weights '
                                   '(sigma %.2f), conf %.2f iou %.2f max_det %d'
                                   % (args.model, args.size, args.size, B, args.dtype, sigma, args.conf, args.iou, args.max_det),
                       'global_batch': world * B, 'parallelism': 'dp%d: images sharded, all-gather of detections' % world,
                       'path': 'Model.forward -> pred[B,N,290] -> lp_nms' if args.via_pred else
                               'detections-only forward (head writes NMS candidates: lp_engine_forward_det) -> lp_nms_candidates',
                       'streams': ('%d batches in flight (one engine + arena + stream each%s) || NMS(+gather) on a post stream' % (depth, '' if pipe is None or pipe.single_lane else ', three execution lanes per forward')
                                   if depth > 1 else 'forward || NMS(+gather) of the previous step') if overlap else 'single stream',
                       'mean_detections_per_image': round(counts, 1),
                       'host_enqueue_ms_per_step': round(t_issue / args.steps * 1e3, 3),
                       'device_ms_per_step': round(dev_ms / args.steps, 3)},
            'roofline': roofline,
        }
        if world == 1 and not args.no_cpu_baseline:
            result['cpu_baseline'] = cpu_baseline(args.model, sigma, args.size, args.conf, args.iou, args.max_det,
                                                  args.cpu_seconds)
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
