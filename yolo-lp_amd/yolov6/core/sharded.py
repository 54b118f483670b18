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


def gather_detections(det, count, group=None, out=None, force_collective=False):
    """All-gather padded detections of equal-sized shards.

    det [b, max_det, 28] fp32 and count [b] int32 of this rank -> (det_all [world*b, max_det, 28],
    count_all [world*b]) on every rank, shards in rank order.  ``out`` may carry preallocated outputs."""
    world = dist.get_world_size(group)
    det = det.contiguous()
    count = count.contiguous()
    if out is None:
        out = (det.new_empty((world * det.shape[0],) + tuple(det.shape[1:])), count.new_empty(world * count.shape[0]))
    det_all, count_all = out
    if world == 1 and not force_collective:
        det_all.copy_(det)
        count_all.copy_(count)
        return det_all, count_all
    try:
        dist.all_gather_into_tensor(det_all, det, group=group)
        dist.all_gather_into_tensor(count_all, count, group=group)
    except (RuntimeError, NotImplementedError):      # backends without the flat variant
        dist.all_gather(list(det_all.chunk(world)), det, group=group)
        dist.all_gather(list(count_all.chunk(world)), count, group=group)
    return det_all, count_all


def unpad(det_all, count_all):
    """Padded batch -> the reference's list of [n_i, 28] tensors (one host sync)."""
    return [det_all[i, :n] for i, n in enumerate(count_all.cpu().tolist())]
