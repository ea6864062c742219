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


def _ring_worker(rank, world, port, out_dir, steps):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from compressed_vec_amd.shard import CountBucketRing
    ring = CountBucketRing(4, "cpu", bucket=8, nbuckets=3)
    seen = np.zeros((steps, 4), dtype=np.int64)
    rows = []
    for rep in range(2):                      # the benchmark runs the ring twice: warm-up steps, then timed steps
        for i in range(steps):
            row, _ = ring.begin_step()
            # the stub step: what a rank's filter pass would leave in its result block
            row.copy_(torch.tensor([(rank + 1) * (i + 1), rank, i, 1], dtype=torch.int64))
            rows.append(ring.end_step())
        ring.drain()
        assert ring.i == 0 and all(p is None for p in ring.pending)
    # after drain() the rows of the last min(steps, ring) steps hold global sums
    for i in range(max(0, steps - ring.ring), steps):
        seen[i] = ring.rows[rows[steps + i]].numpy()
    np.save(os.path.join(out_dir, "ring%d.npy" % rank), seen)
    np.save(os.path.join(out_dir, "coll%d.npy" % rank), np.array([ring.collectives]))
    dist.barrier()
    dist.destroy_process_group()


def test_bucket_ring_world2_full_buckets_reuse_and_partial_drain(tmp_path):
    """The benchmark's step / bucket / drain logic under gloo with two ranks and a stub step: 29 steps =
    3 full buckets, a bucket reused (ring of 24 rows), and a partial bucket of 5 rows reduced by drain()."""
    world, steps = 2, 29
    mp.spawn(_ring_worker, args=(world, _free_port(), str(tmp_path), steps), nprocs=world, join=True)
    for r in range(world):
        seen = np.load(tmp_path / ("ring%d.npy" % r))
        # steps 24..28 reused rows 0..4 of the first bucket; drain() zeroed its rows 5..7 (steps 5..7, reduced
        # long before) so that the partial bucket's reduction does not sum them a second time
        for i in range(8, steps):
            assert list(seen[i]) == [3 * (i + 1), 1, 2 * i, 2], (r, i, seen[i])
        assert not seen[5:8].any()
        # 29 steps = 3 full buckets + 1 partial, twice
        assert int(np.load(tmp_path / ("coll%d.npy" % r))[0]) == 8


def test_bucket_ring_single_rank_skips_collectives():
    sys.path.insert(0, ROOT)
    from compressed_vec_amd.shard import CountBucketRing
    ring = CountBucketRing(4, "cpu", reduce=False)
    for i in range(30):
        row, s = ring.begin_step()
        row.fill_(i)
        assert s == 0 and ring.end_step() == i % 24
    ring.drain()
    assert ring.collectives == 0 and int(ring.rows[5][0]) == 29


def test_bench_starts_its_own_ranks_and_says_what_it_needs():
    """`python bench.py --gpus 2` without a launcher: the parent spawns two ranks before touching the GPU; on
    a machine with fewer than two GPUs the ranks leave before the rendezvous with a clear message."""
    import subprocess
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("this machine has the GPUs; the message under test is for one that has not")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "needs 2 GPUs on this node" in r.stderr
    assert "the 2 ranks ended with exit code" in r.stderr


def test_partition_balances_bytes_not_vectors():
    sys.path.insert(0, ROOT)
    from compressed_vec_amd.shard import partition_by_bytes
    bo = np.concatenate([[0], np.cumsum([10] * 90 + [1000] * 10)]).astype(np.uint64)
    cuts = partition_by_bytes(bo, 4)
    assert cuts[0] == 0 and cuts[-1] == 100 and cuts == sorted(cuts)
    sizes = [int(bo[cuts[i + 1]] - bo[cuts[i]]) for i in range(4)]
    assert max(sizes) - min(sizes) <= 1000
    assert partition_by_bytes(np.array([0], dtype=np.uint64), 8) == [0] * 9
