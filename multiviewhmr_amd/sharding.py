"""Batch sharding of the un-projection path across the GPUs of a node.

Every sample of a batch is independent in `unprojection` (reference loop `models/aggregation.py:28`), so the
path shards by batch with no data-path collective; the only exchange is, when training, the gradient of the
aggregator's 1x1 conv (`process_feature`, `models/aggregation.py:108-110`): one flat fp32 all-reduce
(0.26 MB at 256->256 channels, 2.1 MB at 2048->256), latency-bound over xGMI, so one call rather than buckets.
Backend "nccl" is RCCL on ROCm; the CPU tests use gloo.
"""
import torch
import torch.distributed as dist


def shard_bounds(batch_size, world_size, rank):
    """[lo, hi) of the samples rank owns: consecutive, sizes differ by at most one (reference DDP intent, train.py:166-168)."""
    if not 0 <= rank < world_size:
        raise ValueError("rank %d outside world of %d" % (rank, world_size))
    base, extra = divmod(batch_size, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_batch(features, proj_matricies, coord_volumes, world_size=None, rank=None):
    """This rank's slice of the three inputs of `unprojection` (views of the originals, no copy)."""
    world_size = dist.get_world_size() if world_size is None else world_size
    rank = dist.get_rank() if rank is None else rank
    lo, hi = shard_bounds(features.shape[0], world_size, rank)
    return features[lo:hi], proj_matricies[lo:hi], coord_volumes[lo:hi]


def shard_batch_dict(batch, world_size=None, rank=None):
    """Slice the reference's `batch` dict (data/data_utils.py:25-27): images (B,V,H,W,3), cameras[v][b], keypoints_3d[b]."""
    world_size = dist.get_world_size() if world_size is None else world_size
    rank = dist.get_rank() if rank is None else rank
    lo, hi = shard_bounds(len(batch['keypoints_3d']), world_size, rank)
    out = dict(batch)
    out['images'] = batch['images'][lo:hi]
    out['cameras'] = [row[lo:hi] for row in batch['cameras']]
    out['keypoints_3d'] = batch['keypoints_3d'][lo:hi]
    return out


def allreduce_aggregator_grads(module, group=None, average=True):
    """Sum (then average) the gradients of `module`'s parameters across ranks with ONE all-reduce on a flat buffer.

    `module` is a VolumeGenerator (or anything with parameters); parameters without a gradient contribute zeros so
    that every rank reduces the same layout.  Returns the number of fp32 elements reduced."""
    params = [p for p in module.parameters() if p.requires_grad]
    if not params:
        return 0
    flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1).float() for p in params])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    if average:
        flat /= dist.get_world_size(group)
    off = 0
    for p in params:
        n = p.numel()
        g = flat[off:off + n].view_as(p).to(p.dtype)
        if p.grad is None:
            p.grad = g.clone()
        else:
            p.grad.copy_(g)
        off += n
    return off
