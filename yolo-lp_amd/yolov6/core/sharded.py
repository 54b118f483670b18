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


def gather_detections(det, count, group=None, out=None, force_collective=False, global_batch=None):
    """All-gather the padded detections of every rank's shard.

    det [b, max_det, 28] fp32 and count [b] int32 of this rank -> (det_all [B, max_det, 28], count_all [B]) on every
    rank, shards in rank order.  With ``global_batch`` = B the shards are those of ``shard_bounds`` and may differ in
    size by one image (B % world != 0): every rank pads its shard to the largest one for the collective (an
    all_gather_into_tensor needs equal contributions) and the padding rows are dropped afterwards.  Without it all shards
    must have this rank's size.  ``out`` may carry preallocated outputs of the final shapes."""
    world = dist.get_world_size(group)
    det = det.contiguous()
    count = count.contiguous()
    b = det.shape[0]
    if global_batch is None:
        sizes = [b] * world
    else:
        sizes = [hi - lo for lo, hi in (shard_bounds(global_batch, r, world) for r in range(world))]
        rank = dist.get_rank(group)
        if sizes[rank] != b:
            raise ValueError('rank %d holds %d images, shard_bounds(%d, %d, %d) says %d' % (rank, b, global_batch, rank, world, sizes[rank]))
    total, bmax = sum(sizes), max(sizes)
    if out is None:
        out = (det.new_empty((total,) + tuple(det.shape[1:])), count.new_empty(total))
    det_all, count_all = out
    if tuple(det_all.shape) != (total,) + tuple(det.shape[1:]) or count_all.numel() != total:
        raise ValueError('gather_detections: `out` does not have the gathered shapes')
    if world == 1 and not force_collective:
        det_all.copy_(det)
        count_all.copy_(count)
        return det_all, count_all
    even = min(sizes) == bmax
    if even:
        det_buf, cnt_buf, det_in, cnt_in = det_all, count_all, det, count
    else:       # pad to the largest shard; the collective stays one flat all-gather
        det_in = det.new_zeros((bmax,) + tuple(det.shape[1:]))
        cnt_in = count.new_zeros(bmax)
        det_in[:b].copy_(det)
        cnt_in[:b].copy_(count)
        det_buf = det.new_empty((world * bmax,) + tuple(det.shape[1:]))
        cnt_buf = count.new_empty(world * bmax)
    try:
        dist.all_gather_into_tensor(det_buf, det_in, group=group)
        dist.all_gather_into_tensor(cnt_buf, cnt_in, group=group)
    except (RuntimeError, NotImplementedError):      # backends without the flat variant
        dist.all_gather(list(det_buf.chunk(world)), det_in, group=group)
        dist.all_gather(list(cnt_buf.chunk(world)), cnt_in, group=group)
    if not even:
        o = 0
        for r, n in enumerate(sizes):
            det_all[o:o + n].copy_(det_buf[r * bmax:r * bmax + n])
            count_all[o:o + n].copy_(cnt_buf[r * bmax:r * bmax + n])
            o += n
    return det_all, count_all


def unpad(det_all, count_all):
    """Padded batch -> the reference's list of [n_i, 28] tensors (one host sync)."""
    return [det_all[i, :n] for i, n in enumerate(count_all.cpu().tolist())]
