"""Multi-GPU sharding of the `image` hot path: samples are independent units, so ranks
(one process per GPU) take disjoint sample sets and never exchange data-path bytes.

The reference maps samples over a multiprocessing.Pool (commands/image.py:1281-1284) and
lets the parent merge the per-sample stats (:1144-1170); here the pool is the set of
ranks of a torch.distributed job and rank 0 merges.  The only collectives are control
plane: a barrier, a MAX-reduce of elapsed time (bench.py) and an object gather of the
small stats dicts -- over gloo on CPU tensors, or RCCL when the group is `nccl`.
"""
import os
from collections import OrderedDict


def world_info():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


def shard_indices(n_items, rank, world):
    """Round-robin shard: item i belongs to rank i % world (balanced to within one item,
    and neighbouring -- similarly sized -- samples land on different GPUs)."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    return list(range(rank, n_items, world))


def max_over_ranks(value, device=None):
    """MAX-reduce a python float over the default process group (1 rank: identity)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_stats(local_stats, dst=0):
    """Merge {sample: OrderedDict(...)} dicts of all ranks on rank `dst` (others get None),
    the counterpart of process_stats' all_stats.update (commands/image.py:1167-1168)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return OrderedDict(local_stats)
    rank, world = dist.get_rank(), dist.get_world_size()
    if dist.get_backend() == "nccl":
        # object collectives need CPU tensors: all_gather_object handles the device hop itself
        bucket = [None] * world
        dist.all_gather_object(bucket, dict(local_stats))
    else:
        bucket = [None] * world if rank == dst else None
        dist.gather_object(dict(local_stats), bucket, dst=dst)
    if rank != dst:
        return None
    merged = OrderedDict()
    for part in bucket:
        for k in sorted(part):
            merged.setdefault(k, OrderedDict()).update(part[k])
    return OrderedDict(sorted(merged.items()))
