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


class Communicator:
    """An RCCL communicator made and used through the library's C entry points (cvec_comm_*, cvec_allreduce_counts):
    what a host program without RCCL bindings of its own uses.  Rank 0 creates the id (Communicator.unique_id()) and
    hands its 128 bytes to the other ranks by any means (a file, a socket, torch.distributed's broadcast); every rank
    then constructs Communicator(id, n_ranks, rank) with its GPU current."""

    def __init__(self, comm_id, n_ranks, rank):
        import ctypes as C
        from . import _native as N
        assert len(comm_id) == 128
        self._buf = C.create_string_buffer(bytes(comm_id), 128)
        h = C.c_void_p()
        N.check(N.lib().cvec_comm_init_rank(self._buf, int(n_ranks), int(rank), C.byref(h)))
        self.handle, self.n_ranks, self.rank = h, int(n_ranks), int(rank)

    @staticmethod
    def unique_id():
        import ctypes as C
        from . import _native as N
        buf = C.create_string_buffer(128)
        N.check(N.lib().cvec_comm_unique_id(buf))
        return buf.raw

    def allreduce_counts(self, d_result, n_blocks=None):
        """Reduces result blocks (int64 device tensor of 4 * n_blocks words) in place over the ranks, on the current
        stream: totals and failed-vector counts are summed, the largest status takes the maximum."""
        import ctypes as C
        import torch
        from . import _native as N
        nb = d_result.numel() // N.RESULT_WORDS if n_blocks is None else int(n_blocks)
        assert d_result.is_cuda and d_result.dtype == torch.int64 and d_result.is_contiguous() and nb * N.RESULT_WORDS <= d_result.numel()
        N.check(N.lib().cvec_allreduce_counts(C.c_void_p(d_result.data_ptr()), nb, self.handle,
                                              C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        return d_result

    def close(self):
        from . import _native as N
        if self.handle is not None and self.handle.value:
            N.check(N.lib().cvec_comm_destroy(self.handle))
        self.handle = None


class CountBucketRing:
    """Result blocks of consecutive query steps, all-reduced in buckets.

    Every step writes its result block (`words` int64: total count, failed vectors, ...) into the next
    row of a bucket; a full bucket of `bucket` rows goes out as ONE asynchronous all-reduce, beside the
    kernels of the following steps.  `nbuckets` buckets rotate; a bucket is written again only after its
    last reduction has been waited for.  drain() reduces the partial bucket of the last steps (rows not
    written since the bucket's last reduction are zeroed first) and waits for everything, so after it
    every step's counts are global sums.  Works on CUDA tensors over RCCL and on CPU tensors over gloo.

    `streams`: the torch.cuda streams steps are enqueued on (step i on stream i % len); None on the CPU
    and for a single-stream caller.  `reduce=False` keeps the bookkeeping and skips the collectives
    (one rank)."""

    def __init__(self, words, device, bucket=8, nbuckets=3, streams=None, reduce=True):
        import torch
        self.bucket, self.nbuckets = bucket, nbuckets
        self.buckets = [torch.zeros(bucket, words, dtype=torch.int64, device=device) for _ in range(nbuckets)]
        self.rows = [self.buckets[k][j] for k in range(nbuckets) for j in range(bucket)]
        self.ring = bucket * nbuckets
        self.pending = [None] * nbuckets
        self.streams = list(streams) if streams else []
        self.reduce = reduce
        self.i = 0
        self.collectives = 0

    def _stream(self, idx):
        import contextlib
        import torch
        return torch.cuda.stream(self.streams[idx]) if self.streams else contextlib.nullcontext()

    def _wait(self, k):
        if self.pending[k] is None:
            return
        if self.streams:                    # a stream-side wait on every stream that will write the bucket
            for idx in range(len(self.streams)):
                with self._stream(idx):
                    self.pending[k].wait()
        else:
            self.pending[k].wait()
        self.pending[k] = None

    def _all_reduce(self, k, on_stream):
        import torch.distributed as dist
        if self.streams:                    # the bucket's other steps may be on other streams
            for idx, st in enumerate(self.streams):
                if idx != on_stream:
                    self.streams[on_stream].wait_stream(st)
        with self._stream(on_stream):
            self.pending[k] = dist.all_reduce(self.buckets[k], op=dist.ReduceOp.SUM, async_op=True)
        self.collectives += 1

    def begin_step(self):
        """The row the next step writes, and the index of the stream to enqueue it on."""
        j = self.i % self.ring
        if j % self.bucket == 0:            # first step into a bucket: its last reduction must be over
            self._wait(j // self.bucket)
        return self.rows[j], (self.i % len(self.streams) if self.streams else 0)

    def end_step(self):
        """After the step has been enqueued: a full bucket goes out.  Returns the step's row index."""
        j = self.i % self.ring
        on_stream = self.i % len(self.streams) if self.streams else 0
        self.i += 1
        if self.reduce and j % self.bucket == self.bucket - 1:
            self._all_reduce(j // self.bucket, on_stream)
        return j

    def drain(self):
        """Reduces the partial bucket, waits for every reduction, and starts over at row 0."""
        if self.i:
            j = (self.i - 1) % self.ring
            if self.reduce and j % self.bucket != self.bucket - 1:
                k = j // self.bucket
                with self._stream(0):
                    if self.streams:
                        for st in self.streams[1:]:
                            self.streams[0].wait_stream(st)
                    self.buckets[k][j % self.bucket + 1:].zero_()
                self._all_reduce(k, 0)
        for k in range(self.nbuckets):
            if self.pending[k] is not None:
                self.pending[k].wait()
                self.pending[k] = None
        self.i = 0
