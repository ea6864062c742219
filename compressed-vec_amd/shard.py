"""Sharding of a batch across the GPUs of a node.  Vectors are independent, so a shard is a
contiguous range of vectors chosen to balance *compressed bytes* (decode cost follows bytes, and
compressed size varies ~270x between sparse and dense vectors); no vector data ever crosses GPUs.
The only collective is the all-reduce of the match counts."""
import numpy as np


def partition_by_bytes(byte_off, world):
    """Returns world+1 vector indices: rank r owns vectors [cut[r], cut[r+1]).  Cuts are placed where
    the cumulative compressed size crosses r/world of the total."""
    bo = np.asarray(byte_off, dtype=np.uint64)
    n = bo.size - 1
    total = int(bo[-1] - bo[0])
    cuts = [0]
    for r in range(1, world):
        target = int(bo[0]) + (total * r) // world
        c = int(np.searchsorted(bo, np.uint64(target), side="left"))
        cuts.append(min(max(c, cuts[-1]), n))
    cuts.append(n)
    return cuts


def shard_of(host_batch, rank, world):
    """The sub-batch of `host_batch` that `rank` owns (byte offsets rebased to the shard's blob)."""
    from .batch import EncodedBatch
    cuts = partition_by_bytes(host_batch.byte_off, world)
    a, b = cuts[rank], cuts[rank + 1]
    bo = host_batch.byte_off[a:b + 1]
    blob = host_batch.blob[int(bo[0]):int(bo[-1])]
    return EncodedBatch(host_batch.etype, blob, bo - bo[0]), (a, b)


def allreduce_counts(counts):
    """Sums match counts (int64 tensor, on the device for RCCL or on the CPU for gloo) over all ranks,
    in place.  An 8-byte-per-predicate message: latency-bound, so one call per query."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(counts, op=dist.ReduceOp.SUM)
    return counts
