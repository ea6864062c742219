"""World-size-2 test of the N > 1 path on CPU (gloo): byte-balanced sharding of a batch, a count per
shard, all-reduce of the counts.  The per-shard count comes from the CPU oracle here (no GPU in this
suite); the sharding and the collective are the code the GPU path uses."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make_batch():
    sys.path.insert(0, ROOT)
    import oracle
    from compressed_vec_amd import synth
    from compressed_vec_amd.batch import EncodedBatch
    rng = np.random.default_rng(11)
    vals, off = [], [0]
    for v in range(40):   # dense and sparse vectors mixed, so vector counts and byte counts disagree
        n = int(rng.integers(100, 3000))
        x = synth.sinewave_varnonzeros_u32(1.0, n, 37 * v)
        if v % 3 == 0:
            x[:] = 0
            x[::97] = 3
        vals.append(x)
        off.append(off[-1] + n)
    blob, bo = oracle.batch_encode(oracle.U32, np.concatenate(vals), np.array(off, dtype=np.uint64))
    return EncodedBatch("u32", blob, bo)


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle
    from compressed_vec_amd.shard import allreduce_counts, shard_of
    host = _make_batch()
    shard, (a, b) = shard_of(host, rank, world)
    total, per, st, rc = oracle.batch_filter_count(oracle.U32, shard.blob, shard.byte_off, 3)
    assert rc == 0
    counts = torch.tensor([total, b - a, int(shard.byte_off[-1])], dtype=torch.int64)
    allreduce_counts(counts)
    np.save(os.path.join(out_dir, "rank%d.npy" % rank), counts.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_count_allreduce_world2(tmp_path, oracle):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    host = _make_batch()
    want, _, _, rc = oracle.batch_filter_count(oracle.U32, host.blob, host.byte_off, 3)
    assert rc == 0
    for r in range(world):
        got = np.load(tmp_path / ("rank%d.npy" % r))
        assert got[0] == want and got[1] == host.n_vectors and got[2] == int(host.byte_off[-1])


def test_partition_balances_bytes_not_vectors():
    sys.path.insert(0, ROOT)
    from compressed_vec_amd.shard import partition_by_bytes
    bo = np.concatenate([[0], np.cumsum([10] * 90 + [1000] * 10)]).astype(np.uint64)
    cuts = partition_by_bytes(bo, 4)
    assert cuts[0] == 0 and cuts[-1] == 100 and cuts == sorted(cuts)
    sizes = [int(bo[cuts[i + 1]] - bo[cuts[i]]) for i in range(4)]
    assert max(sizes) - min(sizes) <= 1000
    assert partition_by_bytes(np.array([0], dtype=np.uint64), 8) == [0] * 9
