"""cvec_filter_count_static_dev (the entry point that allocates nothing: its scratch is a static device
array inside the library) against cvec_filter_count_dev, through the C ABI: two calls in a row, the
second over a smaller batch, so that what the first call left in the shared scratch is in the way."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cv():
    import torch
    assert torch.cuda.is_available(), "these tests need a GPU"
    import compressed_vec_amd as cv
    assert cv.lib().cvec_init(0) == 0
    return cv


def _both_ways(cv, dev, pred):
    """(total, per-vector counts, statuses, masks) from the workspace entry point and from the static one."""
    import torch
    from compressed_vec_amd import _native as N
    from compressed_vec_amd.batch import _ptr, _stream_ptr, make_pred
    lib = cv.lib()
    p = make_pred(dev.etype, pred)
    out = []
    for static in (False, True):
        res = torch.full((N.RESULT_WORDS,), -1, dtype=torch.int64, device="cuda")
        per = torch.full((dev.n_vectors,), -1, dtype=torch.int64, device="cuda")
        st = torch.full((dev.n_vectors,), -1, dtype=torch.int32, device="cuda")
        mk = torch.full((dev.n_sections * 32,), 0x5a, dtype=torch.uint8, device="cuda")
        if static:
            # the property the entry point exists for: the call takes no device memory (the free byte count of the
            # GPU is the same before and after) ...
            torch.cuda.synchronize()
            free_before = torch.cuda.mem_get_info()[0]
            N.check(lib.cvec_filter_count_static_dev(C.byref(dev.view), C.byref(p), _ptr(res), _ptr(per), _ptr(st),
                                                     _ptr(mk), _stream_ptr()))
            torch.cuda.synchronize()
            assert torch.cuda.mem_get_info()[0] == free_before
            first = (res.clone(), per.clone(), st.clone(), mk.clone())
            # ... and can be recorded under stream capture, where an allocation or a synchronisation would end the
            # capture with an error; the replayed graph gives the same results
            for t in (res, per, st, mk):
                t.fill_(0x33)
            graph = torch.cuda.CUDAGraph()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                with torch.cuda.graph(graph, stream=side):
                    N.check(lib.cvec_filter_count_static_dev(C.byref(dev.view), C.byref(p), _ptr(res), _ptr(per), _ptr(st),
                                                             _ptr(mk), C.c_void_p(side.cuda_stream)))
                graph.replay()
            side.synchronize()
            torch.cuda.synchronize()
            for got, want in zip((res, per, st), first[:3]):
                assert torch.equal(got, want)
            res, per, st, mk = first
        else:
            ws = dev.workspace()
            N.check(lib.cvec_filter_count_dev(C.byref(dev.view), C.byref(p), _ptr(res), _ptr(per), _ptr(st), _ptr(mk),
                                              _ptr(ws), ws.numel(), _stream_ptr()))
        torch.cuda.synchronize()
        out.append((res.cpu().numpy().copy(), per.cpu().numpy().copy(), st.cpu().numpy().copy(), mk.cpu().numpy().copy()))
    return out


def test_static_scratch_entry_point_equals_workspace_entry_point(cv, oracle):
    import torch
    from compressed_vec_amd import _native as N
    rng = np.random.default_rng(7)
    # first call: ~600 ragged u32 vectors of every section kind, a few of them damaged
    vecs = []
    for i in range(600):
        n = int(rng.integers(0, 3000))
        kind = i % 4
        if kind == 0:
            v = rng.integers(0, 16, size=n)
        elif kind == 1:
            v = rng.integers(0, 1 << 30, size=n)
        elif kind == 2:
            v = np.where(rng.random(n) < 0.9, 0, rng.integers(1, 1000, size=n))
        else:
            v = rng.integers(0, 50, size=n) + 100000
        vecs.append(v.astype(np.uint32))
    eo = np.concatenate([[0], np.cumsum([v.size for v in vecs])]).astype(np.uint64)
    blob, bo = oracle.batch_encode(oracle.U32, np.concatenate(vecs), eo)
    blob = blob.copy()
    for v in (5, 77, 300):                      # damage a byte in the middle of three vectors
        a, b = int(bo[v]), int(bo[v + 1])
        if b - a > 40:
            blob[a + (b - a) // 2] ^= 0xff
    big = cv.EncodedBatch("u32", blob, bo).to_device()
    assert cv.lib().cvec_workspace_bytes(big.n_vectors, big.n_sections) <= N.STATIC_SCRATCH_BYTES
    # second call: a smaller f32 batch (the scratch still holds the first call's descriptors and results)
    from compressed_vec_amd import synth
    fv, feo = synth.batch_values("f32_xor", 40, 2048)
    small = cv.encode_batch("f32", fv, feo).to_device()
    for dev, pred, et, hb, hbo in ((big, 7, oracle.U32, blob, bo), (small, float(fv[1]), oracle.F32, None, None)):
        (r0, p0, s0, m0), (r1, p1, s1, m1) = _both_ways(cv, dev, pred)
        assert np.array_equal(r0, r1) and np.array_equal(p0, p1) and np.array_equal(s0, s1)
        # masks: the sections every vector yielded (those behind a failing section are unspecified)
        valid, _ = dev.valid_sections()          # of the workspace pass, which ran last on dev's own workspace
        sb = dev.d_sec_base.cpu().numpy()
        keep = np.zeros(dev.n_sections * 32, dtype=bool)
        for v in range(dev.n_vectors):
            keep[32 * int(sb[v]):32 * (int(sb[v]) + int(valid[v]))] = True
        assert keep.any() and np.array_equal(m0[keep], m1[keep])
        if hb is None:
            host = dev.download()
            hb, hbo = host.blob, host.byte_off
        wtotal, wper, wst, _rc = oracle.batch_filter_count(et, hb, hbo, pred)
        assert int(r1[0]) == wtotal and np.array_equal(p1.view(np.uint64), wper) and np.array_equal(s1, wst)
    # a batch beyond the static scratch is refused with a clear error, not truncated
    from compressed_vec_amd.batch import _ptr, _stream_ptr, make_pred
    v = N.DBatchView()
    C.memmove(C.byref(v), C.byref(big.view), C.sizeof(v))
    v.n_sections = 1 << 20
    res = torch.zeros(N.RESULT_WORDS, dtype=torch.int64, device="cuda")
    p = make_pred(big.etype, 7)
    rc = cv.lib().cvec_filter_count_static_dev(C.byref(v), C.byref(p), _ptr(res), None, None, None, _stream_ptr())
    assert rc == N.ERR_WORKSPACE
    assert b"static scratch" in cv.lib().cvec_last_hip_error()
