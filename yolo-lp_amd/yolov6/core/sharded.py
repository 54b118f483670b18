"""Data-parallel inference over the GPUs of one node (new functionality: the
reference has no multi-GPU inference).

Images are the independent units: the global batch is split contiguously, every
rank runs forward + NMS on its shard with replicated weights, and the only
exchange is one all-gather of the padded detections ``[B/n, max_det, 28]`` and
their counts (RCCL over xGMI through ``torch.distributed`` backend ``nccl``; the
same code runs over ``gloo`` on CPU tensors, which is what the CPU tests use).
"""
import torch
import torch.distributed as dist


def shard_bounds(global_batch, rank, world_size):
    """Contiguous [lo, hi) image range of ``rank``; the first ``global_batch % world_size`` ranks get one more."""
    base, extra = divmod(global_batch, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


class ShardSizeError(ValueError):
    """A rank entered gather_detections with a shard that is not the one the global batch assigns to it."""


def _all_gather_flat(buf, inp, world, group):
    try:
        dist.all_gather_into_tensor(buf, inp, group=group)
    except (RuntimeError, NotImplementedError):      # backends without the flat variant
        dist.all_gather(list(buf.chunk(world)), inp, group=group)


def _check_sizes(seen, expected, where):
    seen = [int(v) for v in seen]
    if seen != list(expected):
        bad = [r for r, (a, b) in enumerate(zip(seen, expected)) if a != b]
        raise ShardSizeError('%s: ranks %s hold %s images, expected %s (per-rank shard sizes seen %s, expected %s)' % (
            where, bad, [seen[r] for r in bad], [expected[r] for r in bad], seen, list(expected)))


def gather_detections(det, count, group=None, out=None, force_collective=False, global_batch=None, check=None):
    """All-gather the padded detections of every rank's shard.

    det [b, max_det, 28] fp32 and count [b] int32 of this rank -> (det_all [B, max_det, 28], count_all [B]) on every
    rank, shards in rank order.  With ``global_batch`` = B the shards are those of ``shard_bounds`` and may differ in
    size by one image (B % world != 0): every rank pads its shard to the largest one for the collective (an
    all_gather_into_tensor needs equal contributions) and the padding rows are dropped afterwards.  Without it all shards
    must have this rank's size.  ``out`` may carry preallocated outputs of the final shapes.

    A wrong shard size never strands the other ranks in a collective: nothing is raised BEFORE the collectives.
    With ``global_batch`` every contribution has the size ``shard_bounds`` dictates whatever a rank holds (a wrong shard is
    cut or zero-padded) and each rank's true size rides in one extra element of the count collective; afterwards the faulty
    rank raises ``ShardSizeError`` from what it knows on the host, and EVERY rank raises it as soon as it reads the sizes:
    at once for CPU tensors or ``check=True``, else (device tensors: reading them is a host sync) in ``unpad`` / ``check_shards``.
    Without ``global_batch`` the contribution sizes depend on what each rank holds, so the sizes are exchanged first in a
    one-element collective and compared on the host by every rank (one sync: pass ``global_batch`` in a hot loop)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    det = det.contiguous()
    count = count.contiguous()
    b = det.shape[0]
    if world == 1 and not force_collective:
        total = b if global_batch is None else global_batch
        if total != b:
            raise ShardSizeError('gather_detections: one rank holds %d images of a global batch of %d' % (b, total))
        if out is None:
            out = (det.new_empty(det.shape), count.new_empty(b))
        det_all, count_all = out
        if tuple(det_all.shape) != tuple(det.shape) or count_all.numel() != b:
            raise ValueError('gather_detections: `out` does not have the gathered shapes')
        det_all.copy_(det)
        count_all.copy_(count)
        return det_all, count_all
    if global_batch is None:
        mine = torch.tensor([b], dtype=torch.int64, device=det.device)
        seen = torch.empty(world, dtype=torch.int64, device=det.device)
        _all_gather_flat(seen, mine, world, group)
        seen = seen.cpu().tolist()
        _check_sizes(seen, [seen[0]] * world, 'gather_detections (equal shards expected)')
        sizes = [b] * world
    else:
        sizes = [hi - lo for lo, hi in (shard_bounds(global_batch, r, world) for r in range(world))]
    total, bmax = sum(sizes), max(sizes)
    tail = tuple(det.shape[1:])
    bad_out = out is not None and (tuple(out[0].shape) != (total,) + tail or out[1].numel() != total)
    if out is None or bad_out:                       # (a wrong `out` is reported after the collectives, like a wrong shard)
        det_all, count_all = det.new_empty((total,) + tail), count.new_empty(total)
    else:
        det_all, count_all = out
    even = min(sizes) == bmax
    n = min(b, bmax)
    # counts: [bmax] values + 1 header element (this rank's true shard size)
    cnt_in = count.new_zeros(bmax + 1)
    cnt_in[:n].copy_(count[:n])
    cnt_in[bmax] = b
    cnt_buf = count.new_empty(world * (bmax + 1))
    if even and b == bmax:
        det_in, det_buf = det, det_all
    else:       # pad to the largest shard; the collective stays one flat all-gather
        det_in = det.new_zeros((bmax,) + tail)
        det_in[:n].copy_(det[:n])
        det_buf = det_all if even else det.new_empty((world * bmax,) + tail)
    _all_gather_flat(det_buf, det_in, world, group)
    _all_gather_flat(cnt_buf, cnt_in, world, group)
    cnt_rows = cnt_buf.view(world, bmax + 1)
    if even:                                         # one strided copy (the header column is dropped)
        count_all.view(world, bmax).copy_(cnt_rows[:, :bmax])
    else:
        o = 0
        for r, k in enumerate(sizes):
            det_all[o:o + k].copy_(det_buf[r * bmax:r * bmax + k])
            count_all[o:o + k].copy_(cnt_rows[r, :k])
            o += k
    if bad_out:
        raise ValueError('gather_detections: `out` does not have the gathered shapes')
    if sizes[rank] != b:                             # known on the host of the faulty rank: no sync needed
        raise ShardSizeError('gather_detections: rank %d holds %d images, shard_bounds(%s, %d, %d) says %d' % (
            rank, b, global_batch, rank, world, sizes[rank]))
    pending = (cnt_rows[:, bmax], tuple(sizes))
    if check or (check is None and not det.is_cuda):
        check_shards(pending)
    else:
        count_all._lp_shard_sizes = pending          # read by unpad() at its host sync
    return det_all, count_all


def check_shards(pending):
    """Compare the shard sizes every rank reported in the count collective with the expected ones (one host sync for
    device tensors); raises ``ShardSizeError`` on every rank alike."""
    seen, expected = pending
    _check_sizes(seen.cpu().tolist(), expected, 'gather_detections')


def unpad(det_all, count_all):
    """Padded batch -> the reference's list of [n_i, 28] tensors (one host sync; also where a deferred shard-size check
    of ``gather_detections`` is made)."""
    pending = getattr(count_all, '_lp_shard_sizes', None)
    if pending is not None:
        check_shards(pending)
        count_all._lp_shard_sizes = None
    return [det_all[i, :n] for i, n in enumerate(count_all.cpu().tolist())]
