"""The all-reduce of result blocks through the C ABI (cvec_allreduce_counts over an RCCL communicator made with
cvec_comm_*): exercised on the one GPU with a communicator of one rank, where the reduction must leave every block
as it was.  Ranks > 1 are covered on the CPU with gloo (tests/test_multi_rank_cpu.py) through the same bookkeeping."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_allreduce_counts_over_a_one_rank_communicator():
    import torch
    assert torch.cuda.is_available(), "this test needs a GPU"
    import compressed_vec_amd as cv
    from compressed_vec_amd import _native as N
    from compressed_vec_amd.shard import Communicator
    lib = cv.lib()
    assert lib.cvec_init(0) == 0
    comm = Communicator(Communicator.unique_id(), 1, 0)
    try:
        # a real pass's result block, and a bucket of eight of them
        vals = (np.arange(20 * 1000) % 7).astype(np.uint32)
        eo = np.arange(21, dtype=np.uint64) * 1000
        dev = cv.encode_batch("u32", vals, eo).to_device()
        res = dev.filter_count_async(3).clone()
        want = res.cpu().numpy().copy()
        assert int(want[0]) == int((vals == 3).sum())
        comm.allreduce_counts(res)
        torch.cuda.synchronize()
        assert np.array_equal(res.cpu().numpy(), want)
        bucket = torch.arange(8 * N.RESULT_WORDS, dtype=torch.int64, device="cuda") * 1000003
        wantb = bucket.cpu().numpy().copy()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            comm.allreduce_counts(bucket)
        s.synchronize()
        assert np.array_equal(bucket.cpu().numpy(), wantb)
        # argument errors never reach RCCL
        assert lib.cvec_allreduce_counts(None, 1, comm.handle, None) == N.ERR_BAD_ARGUMENT
        assert lib.cvec_allreduce_counts(C.c_void_p(res.data_ptr()), 0, comm.handle, None) == N.ERR_BAD_ARGUMENT
        assert lib.cvec_allreduce_counts(C.c_void_p(res.data_ptr()), 1, None, None) == N.ERR_BAD_ARGUMENT
    finally:
        comm.close()
