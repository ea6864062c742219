"""GPU parity tests (run with -m gpu on an MI355X).  Every check goes through the C ABI of
include/cvec.h and compares against the CPU oracle on the same seeded inputs: bit-exact for encoded
bytes, decoded values (f32 compared as raw bits), masks and counts."""
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


ET = {"u32": 0x11, "u64": 0x10, "f32": 0x12}
NPD = {"u32": np.uint32, "u64": np.uint64, "f32": np.float32}


def _bits(a):
    a = np.ascontiguousarray(a)
    return a.view(np.uint32 if a.dtype.itemsize == 4 else np.uint64)


def _random_vectors(rng, tname, n_vectors, max_len=1500):
    """Ragged vectors mixing every section kind: zeros, constants, narrow ranges around a big base,
    wide values, sparse values."""
    vals, off = [], [0]
    for _ in range(n_vectors):
        n = int(rng.integers(0, max_len))
        parts = []
        left = n
        while left > 0:
            k = int(min(left, rng.integers(1, 700)))
            mode = rng.integers(0, 6)
            if tname == "f32":
                if mode == 0:
                    p = np.zeros(k, dtype=np.float32)
                elif mode == 1:
                    p = np.full(k, rng.normal() * 100, dtype=np.float32)
                elif mode == 2:
                    p = (rng.integers(0, 16, size=k) / np.float32(1.4)).astype(np.float32)
                elif mode == 3:
                    p = rng.normal(size=k).astype(np.float32)
                elif mode == 4:
                    p = (rng.integers(0, 1 << 20, size=k).astype(np.float32) * np.float32(0.25))
                else:
                    p = rng.normal(size=k).astype(np.float32)
                    p[rng.random(k) < 0.7] = 0
            else:
                bits = 32 if tname == "u32" else 64
                if mode == 0:
                    p = np.zeros(k, dtype=np.uint64)
                elif mode == 1:
                    p = np.full(k, rng.integers(1, 1 << 31), dtype=np.uint64)
                elif mode == 2:
                    p = rng.integers(1, 16, size=k).astype(np.uint64)
                elif mode == 3:
                    nb = int(rng.integers(4, bits))
                    p = rng.integers(0, 1 << nb, size=k, dtype=np.uint64)
                elif mode == 4:
                    base = int(rng.integers(1 << 16, 1 << (bits - 2)))
                    p = (rng.integers(0, 1 << int(rng.integers(2, 14)), size=k).astype(np.uint64) + np.uint64(base))
                else:
                    nb = int(rng.integers(4, bits))
                    p = rng.integers(0, 1 << nb, size=k, dtype=np.uint64)
                    p[rng.random(k) < 0.7] = 0
                p = p.astype(NPD[tname])
            parts.append(p)
            left -= k
        v = np.concatenate(parts) if parts else np.zeros(0, dtype=NPD[tname])
        vals.append(v.astype(NPD[tname]))
        off.append(off[-1] + v.size)
    flat = np.concatenate(vals) if vals else np.zeros(0, dtype=NPD[tname])
    return flat, np.array(off, dtype=np.uint64)


def test_unaligned_loads_and_library_loads(cv):
    assert b"gfx950" in cv.lib().cvec_version()


@pytest.mark.parametrize("tname", ["u32", "u64", "f32"])
def test_hand_derived_vectors_through_gpu_encoder(cv, kats, tname):
    for k in kats["hand_derived_vectors"]:
        if k["type"] != tname:
            continue
        vals = np.array(k["values"], dtype=NPD[tname])
        b = cv.encode_batch(tname, vals, np.array([0, vals.size], dtype=np.uint64))
        assert list(b.vector(0)) == k["expected"], k["src"]


@pytest.mark.parametrize("tname", ["u32", "u64", "f32"])
def test_encode_bytes_bit_exact_vs_oracle(cv, oracle, tname):
    rng = np.random.default_rng(100 + ET[tname])
    vals, eo = _random_vectors(rng, tname, 300)
    want_blob, want_off = oracle.batch_encode(ET[tname], vals, eo)
    got = cv.encode_batch(tname, vals, eo)
    assert np.array_equal(got.byte_off, want_off)
    assert np.array_equal(got.blob, want_blob)


@pytest.mark.parametrize("tname", ["u32", "u64", "f32"])
def test_encode_runs_of_null_and_const_sections_vs_oracle(cv, oracle, tname):
    """Sections without octets (Null: one byte, src/section.rs:440-443; Const: type byte + value, :795-803) are
    written by the emit pass outside its loop over sections with octets: whole runs of them, single ones between
    value sections, at the end of a vector (a short last section is padded with zeros: Null only if its values are
    zero), signed zeros and NaN constants for f32."""
    rng = np.random.default_rng(4100 + ET[tname])
    dt = NPD[tname]
    vals, off = [], [0]
    for v in range(260):
        nsec = int(rng.integers(1, 45))
        pieces = []
        for s in range(nsec):
            k = 256 if s + 1 < nsec or rng.random() < 0.5 else int(rng.integers(1, 256))
            kind = int(rng.integers(0, 6)) if v % 3 else int(rng.integers(0, 3))
            if kind == 0:
                p = np.zeros(k, dtype=dt)
            elif kind == 1:
                c = rng.integers(1, 1 << 31) if tname != "f32" else np.float32(rng.normal())
                p = np.full(k, c, dtype=dt)
            elif kind == 2 and tname == "f32":
                bits = rng.choice(np.array([0x80000000, 0x7fc00000, 0x7f800000, 0x00000001], dtype=np.uint32))
                p = np.full(k, bits, dtype=np.uint32).view(np.float32)
            elif kind == 2:
                p = np.full(k, np.iinfo(dt).max, dtype=dt)
            elif tname == "f32":
                p = (rng.integers(0, 1 << 12, size=k) / 1.4).astype(np.float32)
            else:
                p = rng.integers(0, 1 << int(rng.integers(1, 30)), size=k).astype(dt)
            pieces.append(p)
        x = np.concatenate(pieces)
        vals.append(x)
        off.append(off[-1] + x.size)
    flat, eo = np.concatenate(vals), np.array(off, dtype=np.uint64)
    want_blob, want_off = oracle.batch_encode(ET[tname], flat, eo)
    got = cv.encode_batch(tname, flat, eo)
    assert np.array_equal(got.byte_off, want_off)
    assert np.array_equal(got.blob, want_blob)


@pytest.mark.parametrize("tname", ["u32", "u64", "f32"])
def test_decode_bit_exact_vs_oracle(cv, oracle, tname):
    rng = np.random.default_rng(200 + ET[tname])
    vals, eo = _random_vectors(rng, tname, 300)
    blob, bo = oracle.batch_encode(ET[tname], vals, eo)
    dev = cv.EncodedBatch(tname, blob, bo).to_device()
    got, out_off = dev.decode()
    want, st, rc = oracle.batch_decode(ET[tname], blob, bo, out_off)
    assert rc == 0
    assert np.array_equal(_bits(got), _bits(want))
    # round trip: the first num_elements values of every vector are the inputs
    for v in range(eo.size - 1):
        n = int(eo[v + 1] - eo[v])
        a = got[int(out_off[v]):int(out_off[v]) + n]
        assert np.array_equal(_bits(a), _bits(vals[int(eo[v]):int(eo[v + 1])])) or tname == "f32"
    # AddConstSink in front of VecSink
    base = {"u32": 4000000000, "u64": (1 << 63) + 12345, "f32": 2.5}[tname]
    got2, _ = dev.decode(add_const=base)
    want2, _, rc = oracle.batch_decode(ET[tname], blob, bo, out_off, add_base=base)
    assert rc == 0
    assert np.array_equal(_bits(got2), _bits(want2))


def test_f32_roundtrip_is_bitwise_unless_section_is_lossy(cv, oracle):
    """XOR sections are bitwise; only the writer's Const/Null choice (min == max under float ==) can
    alter bits (-0.0 / NaN), and the GPU encoder must make exactly the oracle's choice."""
    rng = np.random.default_rng(7)
    vals = rng.normal(size=256 * 40).astype(np.float32)
    vals[256:512] = 0.0
    vals[300] = -0.0                       # mixed zeros → Null section (decodes to +0.0)
    vals[512:768] = 5.0
    vals[513] = np.nan                     # NaN hidden by the fold → lossy Const section
    vals[768:1024] = 7.0
    vals[1023] = np.nan                    # trailing NaN → max = NaN → Xor section, bitwise
    vals[1024] = np.nan                    # leading NaN → min = NaN → Xor
    vals[1280:1536] = -0.0                 # all -0.0 → min == max == 0 → Null
    eo = np.array([0, vals.size], dtype=np.uint64)
    want_blob, want_off = oracle.batch_encode(0x12, vals, eo)
    got = cv.encode_batch("f32", vals, eo)
    assert np.array_equal(got.blob, want_blob)
    dec, _ = got.to_device().decode()
    assert np.array_equal(_bits(dec), _bits(oracle.decode_to_vec(0x12, want_blob)))
    assert np.array_equal(_bits(dec[:256]), _bits(vals[:256]))
    assert np.array_equal(_bits(dec[768:1280]), _bits(vals[768:1280]))


@pytest.mark.parametrize("tname", ["u32", "u64", "f32"])
def test_filter_count_and_masks_vs_oracle(cv, oracle, tname):
    rng = np.random.default_rng(300 + ET[tname])
    vals, eo = _random_vectors(rng, tname, 200)
    blob, bo = oracle.batch_encode(ET[tname], vals, eo)
    dev = cv.EncodedBatch(tname, blob, bo).to_device()
    nz = vals[vals != 0]
    preds = [0, nz[len(nz) // 3], nz[len(nz) // 2], [nz[5], 0, nz[77]], [nz[1], nz[2], nz[3], nz[4], nz[9]]]
    if tname == "f32":
        preds += [-0.0, float("nan"), np.float32(5) / np.float32(1.4)]
    else:
        preds += [3, [3, 5]]
    for p in preds:
        total, per, st, masks = dev.filter_count(p, per_vector=True, return_status=True, return_masks=True)
        wtotal, wper, wst, rc = oracle.batch_filter_count(ET[tname], blob, bo, p)
        assert rc == 0 and np.all(st == 0)
        assert total == wtotal, p
        assert np.array_equal(per, wper), p
        # per-section masks of a few vectors
        sb = dev.d_sec_base.cpu().numpy()
        for v in range(0, eo.size - 1, 37):
            _, wm = oracle.filter_count(ET[tname], blob[int(bo[v]):int(bo[v + 1])], p, return_masks=True)
            gm = masks[32 * int(sb[v]):32 * int(sb[v + 1])]
            assert np.array_equal(gm, wm), (p, v)


@pytest.mark.parametrize("tname", ["u32", "u64", "f32"])
def test_oneof_sets_larger_than_one_pass_vs_oracle(cv, oracle, tname):
    """OneOfSink over sets of 17..50 values (the reference's set is a Vec, src/filter.rs:143-175): the library
    splits them into passes of 16 and merges counts and masks on the device.  Duplicates, zero (and for f32
    both zeros and a NaN) in the set, and a damaged vector whose partial count must add up the same way."""
    rng = np.random.default_rng(900 + ET[tname])
    vals, eo = _random_vectors(rng, tname, 120)
    blob, bo = oracle.batch_encode(ET[tname], vals, eo)
    blob = blob.copy()
    v_bad = 17
    mid = (int(bo[v_bad]) + int(bo[v_bad + 1])) // 2
    blob[mid:int(bo[v_bad + 1])] = 0xEE                      # cut vector 17 short in the middle of a section
    dev = cv.EncodedBatch(tname, blob, bo).to_device()
    distinct = np.unique(vals[vals != 0])
    rng.shuffle(distinct)
    sets = [list(distinct[:17]), list(distinct[:33]) + [0], list(distinct[:24]) * 2 + [distinct[30]]]
    if tname == "f32":
        sets.append(list(distinct[:20]) + [0.0, -0.0, float("nan")] + list(distinct[20:40]))
    else:
        sets.append(list(distinct[:50]))
    for p in sets:
        p = [x.item() if hasattr(x, "item") else x for x in p]
        total, per, st, masks = dev.filter_count(p, per_vector=True, return_status=True, return_masks=True)
        wtotal, wper, wst, rc = oracle.batch_filter_count(ET[tname], blob, bo, p)
        assert total == wtotal and np.array_equal(per, wper) and np.array_equal(st, wst), len(p)
        assert st[v_bad] != 0
        sb = dev.d_sec_base.cpu().numpy()
        valid, _pf = dev.valid_sections()
        for v in list(range(0, eo.size - 1, 13)) + [v_bad]:
            _, wm = oracle.filter_count(ET[tname], blob[int(bo[v]):int(bo[v + 1])], p, return_masks=True)
            wm = np.asarray(wm, dtype=np.uint8).reshape(-1)
            gm = masks[32 * int(sb[v]):32 * (int(sb[v]) + int(valid[v]))]
            assert np.array_equal(gm, wm[:gm.size]), (len(p), v)
        assert dev.filter_count(p, return_status=True)[0] == wtotal     # count only: no per-vector / mask merge


def test_reference_test_cases_through_the_api_mirror(cv):
    """The reference's own vector/filter tests, phrased with the same API names."""
    # src/vector.rs:669-684
    app = cv.VectorU32Appender.try_new(1024)
    for i in range(400):
        app.append((i % 4) + 1)
    finished = app.finish(400)
    reader = cv.VectorReader.try_new(finished, "u32")
    assert reader.num_elements() == 400
    assert cv.count_hits(reader.filter_iter(cv.EqualsSink("u32", 3))) == 100
    # src/vector.rs:686-703
    for i in range(300):
        app.append((i % 4) + 1)
    app.append_nulls(400)
    for i in range(300):
        app.append((i % 4) + 1)
    finished = app.finish(1000)
    reader = cv.VectorReader.try_new(finished, "u32")
    assert cv.count_hits(reader.filter_iter(cv.EqualsSink("u32", 3))) == 150
    sink = cv.VecSink("u32")
    reader.decode_to_sink(sink)
    assert list(sink.vec[:1000]) == list(reader.iterate())
    # src/vector.rs:729-739
    with pytest.raises(cv.CodingError) as e:
        cv.VectorReader.try_new(finished, "u64")
    assert e.value.name == "WrongVectorType" and e.value.detail == (0x11,)
    # src/vector.rs:742-755
    data = (np.arange(280, dtype=np.float32) / np.float32(2.8)).astype(np.float32)
    fin = cv.VectorF32XorAppender.try_new(2048).encode_all(data)
    reader = cv.VectorReader.try_new(fin, "f32")
    assert reader.num_elements() == 280
    sink = cv.VecSink("f32")
    reader.decode_to_sink(sink)
    assert sink.vec[:280].tobytes() == data.tobytes()
    # src/filter.rs:332-348
    fin = cv.VectorU64Appender.try_new(1024).encode_all(np.arange(400, dtype=np.uint64) % 4 + 1)
    reader = cv.VectorReader.try_new(fin, "u64")
    assert cv.match_positions(reader.filter_iter(cv.EqualsSink("u64", 3))) == list(range(2, 400, 4))
    # src/filter.rs:351-371
    fin = cv.VectorU32Appender.try_new(1024).encode_all(np.arange(400, dtype=np.uint32) % 12 + 1)
    reader = cv.VectorReader.try_new(fin, "u32")
    pos = cv.match_positions(reader.filter_iter(cv.OneOfSink("u32", [3, 5])))
    assert pos == [p for i in range(2, 400, 12) for p in (i, i + 2)][:67]
    # README.md:58-65
    enc = cv.VectorF32XorAppender.try_new(2048).encode_all([1.0, 1.5, 2.0, 2.5])
    reader = cv.VectorReader.try_new(enc, "f32")
    vecsink = cv.VecSink("f32")
    reader.decode_to_sink(cv.AddConstSink(2.5, vecsink))
    assert list(vecsink.vec[:4]) == [3.5, 4.0, 4.5, 5.0]


def test_section_iteration_section256_sink_and_ranged_reads(cv, oracle):
    """FixedSectIterator / Section256Sink (src/section.rs:846-884, src/sink.rs:195-232) and VectorItemIter
    ranges (src/vector.rs:512-556) against the oracle's decode of the same bytes; the resident-batch ranged
    read DeviceBatch.vector_items against the same."""
    rng = np.random.default_rng(77)
    app = cv.VectorU32Appender.try_new(512)
    vals = rng.integers(0, 1000, size=700).astype(np.uint32)
    app.append_many(vals)
    app.append_nulls(256 - 700 % 256 + 512)               # rounds out section 3, then two explicit null sections
    app.append_many(np.full(256, 9, dtype=np.uint32))     # a Const section
    fin = app.finish(1800)
    kinds = [int(k) for k in oracle.reader_sections(oracle.U32, fin)]
    want = oracle.decode_to_vec(oracle.U32, fin)
    r = cv.VectorReader.try_new(fin, "u32")
    secs = r.sections()
    assert [s.kind for s in secs] == kinds and sum(s.num_bytes for s in secs) == len(fin) - 16
    assert [s.is_null() for s in secs] == [k == 0 for k in kinds] and kinds.count(0) >= 3 and 5 in kinds
    sink = cv.Section256Sink("u32")
    for s_ in secs:
        sink.reset()
        s_.decode(sink)
        assert np.array_equal(sink.values, want[256 * s_.index:256 * s_.index + 256]), s_.index
        s_.decode(sink)                                   # full: further values are ignored (src/sink.rs:214-219)
        assert np.array_equal(sink.values, want[256 * s_.index:256 * s_.index + 256])
    # through AddConstSink: wrapping add; a Null section yields the base itself
    sink.reset()
    secs[3].decode(cv.AddConstSink(7, sink))
    assert secs[3].is_null() and np.all(sink.values == 7)
    sink.reset()
    secs[0].decode(cv.AddConstSink(0xffffffff, sink))
    assert np.array_equal(sink.values, want[:256] - np.uint32(1))
    # decode_to_sink into a Section256Sink leaves the first section there
    sink.reset()
    r.decode_to_sink(sink)
    assert np.array_equal(sink.values, want[:256])
    # ranged reads
    assert np.array_equal(r.iterate(), want[:1800])
    assert np.array_equal(r.iterate(250, 1030), want[250:1030]) and r.iterate(1700, 5000).size == 100
    other = oracle.encode_all(oracle.U32, np.arange(900, dtype=np.uint32))
    dev = cv.EncodedBatch.from_vectors("u32", [other, fin, other]).to_device()
    assert np.array_equal(dev.vector_items(1, 250, 1030).cpu().numpy().view(np.uint32), want[250:1030])
    assert np.array_equal(dev.vector_items(2).cpu().numpy().view(np.uint32), np.arange(900, dtype=np.uint32))
    with pytest.raises(IndexError):
        dev.vector_items(3)
    # a damaged section: the sections before it still decode, the range behind it raises
    bad = bytearray(fin)
    bad[16 + secs[0].num_bytes + 3 + 1] = 0xf0            # second section, first octet: 16 nibbles in a u32 vector
    rb = cv.VectorReader.try_new(bytes(bad), "u32")
    assert np.array_equal(rb.iterate(0, 256), want[:256])
    with pytest.raises(cv.CodingError):
        rb.iterate(0, 300)
    with pytest.raises(cv.CodingError):
        rb.sections()[1].decode(cv.Section256Sink("u32"))


def test_appender_handle_errors_and_reuse(cv, oracle):
    """The C-ABI appender handle (cvec_appender_*): try_new's capacity check, InvalidNumRows with both numbers
    (src/vector.rs:344-349), finish() resetting the appender (:368), reset(), and bytes equal to the oracle's
    appender driven the same way."""
    with pytest.raises(cv.CodingError) as e:
        cv.VectorU64Appender.try_new(7)
    assert e.value.name == "NotEnoughSpace"
    a, o = cv.VectorF32XorAppender.try_new(64), oracle.Appender(oracle.F32, 64)
    x = (np.arange(300) / 7.0).astype(np.float32)
    for app in (a, o):
        app.append_many(x)
    with pytest.raises(cv.CodingError) as e:
        a.finish(299)
    assert e.value.name == "InvalidNumRows" and e.value.detail == (299, 300)
    with pytest.raises(cv.CodingError) as e:
        a.finish(2**32)
    assert e.value.detail == (2**32, 2**32 - 1)
    assert a.num_elements() == 300                        # a failed finish keeps the contents
    assert a.finish(1000) == o.finish(1000)
    assert a.num_elements() == 0
    for app in (a, o):                                    # the handle is reusable after finish
        app.append_nulls(700)
        app.append_many(x[:10])
    assert a.finish(710) == o.finish(710)
    a.append_many(x)
    a.reset()
    assert a.num_elements() == 0 and a.finish(0) == o.finish(0)


def test_appender_reader_sees_complete_sections_only(cv):
    """VectorAppender::reader (src/vector.rs:372-378) reads what has been encoded so far."""
    app = cv.VectorU32Appender.try_new(1024)
    app.append_many(np.arange(1, 301, dtype=np.uint32))
    r = app.reader()
    assert r.num_elements() == 256 and np.array_equal(r.iterate(), np.arange(1, 257, dtype=np.uint32))
    app.append_nulls(256 - 44 + 256)                 # fills the open section, then one explicit null section
    r = app.reader()
    assert r.num_elements() == 768 and r.get_stats()["num_null_sections"] == 1
    assert np.array_equal(r.iterate()[:300], np.arange(1, 301, dtype=np.uint32)) and not r.iterate()[300:].any()
    assert cv.VectorU32Appender.try_new(64).reader().num_elements() == 0


def test_mixed_null_appender_bytes_match_oracle(cv, oracle):
    # src/vector.rs:601-634, :637-666: explicit null sections are counted in the header
    for cap, d1n, nulls, d2n in [(1024, 100, (256 - 100) + 256 + 50, 50), (300, 300, 350, 300)]:
        a, o = cv.VectorU64Appender.try_new(cap), oracle.Appender(oracle.U64, cap)
        d1, d2 = np.arange(d1n, dtype=np.uint64), np.arange(d2n, dtype=np.uint64)
        for app in (a, o):
            app.append_many(d1)
            app.append_nulls(nulls)
            app.append_many(d2)
        total = d1n + d2n + nulls
        assert a.finish(total) == o.finish(total)
    # finish() with more rows than appended adds explicit null sections (src/vector.rs:357-359)
    a, o = cv.VectorU32Appender.try_new(64), oracle.Appender(oracle.U32, 64)
    for app in (a, o):
        app.append_many(np.arange(10, dtype=np.uint32))
    assert a.finish(1000) == o.finish(1000)


def test_appender_long_null_runs_are_counts(cv, oracle):
    """append_nulls far beyond the values (src/vector.rs:320-327: one byte per null section): the handle keeps the
    runs as counts, the bytes equal the oracle's, readers in between included."""
    a, o = cv.VectorU32Appender.try_new(64), oracle.Appender(oracle.U32, 64)
    x = np.arange(1, 1001, dtype=np.uint32) * 3
    for app in (a, o):
        app.append_many(x[:300])
        app.append_nulls(50_000_000)
        app.append_many(x[300:])
        app.append_nulls(600)
    assert a.reader().num_elements() == o.num_elements() - (o.num_elements() % 256)
    total = 2_000_000_000
    got, want = a.finish(total), o.finish(total)
    assert len(got) < 10_000_000 and got == bytes(want)


def test_appender_holds_encoded_sections_not_values(cv, oracle):
    """The handle's memory behaves like the reference appender's (src/vector.rs:270-281: a section is encoded when
    it fills): after far more than a run of 1024 sections the handle holds encoded bytes and less than one run of raw
    values; the bytes - runs encoded at different times, null runs between them, readers in between - equal the
    oracle's appender driven the same way, for every element type."""
    rng = np.random.default_rng(404)
    for tname, et in (("u32", oracle.U32), ("u64", oracle.U64), ("f32", oracle.F32)):
        a = {"u32": cv.VectorU32Appender, "u64": cv.VectorU64Appender, "f32": cv.VectorF32XorAppender}[tname].try_new(8192)
        o = oracle.Appender(et, 8192)
        w = 8 if tname == "u64" else 4
        total = 0
        for step in range(9):
            n = int(rng.integers(60_000, 140_000))
            if tname == "f32":
                x = (rng.integers(0, 1 << 12, size=n) / 1.4).astype(np.float32)
            else:
                x = rng.integers(0, 1 << int(rng.integers(3, 30)), size=n).astype(NPD[tname])
            for app in (a, o):
                app.append_many(x)
            total += n
            if step % 3 == 1:
                k = int(rng.integers(1, 3000))
                for app in (a, o):
                    app.append_nulls(k)
                total += k
            if step % 4 == 2:
                r = a.reader()
                assert r.num_elements() == total - total % 256
            enc, raw = a.stats()
            assert raw < (1024 + 1) * 256 * w, (tname, step, raw)
        enc, raw = a.stats()
        assert enc > 0 and total > 2 * 1024 * 256                   # at least two runs were encoded on the way
        got, want = a.finish(total + 777), o.finish(total + 777)
        assert got == bytes(want), tname
        assert a.stats() == (0, 0)


def test_null_section_through_addconst_keeps_base_bits(cv, oracle):
    # src/sink.rs:314-317
    enc = cv.VectorF32XorAppender.try_new(64).encode_all(np.zeros(300, dtype=np.float32))
    reader = cv.VectorReader.try_new(enc, "f32")
    s = cv.VecSink("f32")
    reader.decode_to_sink(cv.AddConstSink(-0.0, s))
    assert np.all(s.vec.view(np.uint32) == 0x80000000)


def test_empty_batch_and_empty_vectors(cv, oracle):
    for tname in ("u32", "f32"):
        b = cv.encode_batch(tname, np.zeros(0, dtype=NPD[tname]), np.array([0], dtype=np.uint64))
        assert b.n_vectors == 0 and b.blob.size == 0
        vals = np.arange(5, dtype=NPD[tname])
        eo = np.array([0, 0, 5, 5], dtype=np.uint64)
        got = cv.encode_batch(tname, vals, eo)
        want_blob, want_off = oracle.batch_encode(ET[tname], vals, eo)
        assert np.array_equal(got.blob, want_blob) and np.array_equal(got.byte_off, want_off)
        dev = got.to_device()
        assert dev.n_sections == 1
        assert dev.filter_count(3) == 1
        dec, off = dev.decode()
        assert dec.size == 256 and list(off) == [0, 0, 256, 256]


def test_malformed_vectors_report_status_and_partial_counts(cv, oracle):
    good = oracle.encode_all(oracle.U32, (np.arange(1000, dtype=np.uint32) % 7) + 1)
    vecs = [good]
    bad = bytearray(good)
    bad[5] = 0x10                                   # wrong subtype
    vecs.append(bytes(bad))
    bad = bytearray(good)
    pos = 16
    _, nb = oracle.section_parse(oracle.U32, good[pos:])
    bad[pos + nb] = 9                               # second section: invalid type byte
    vecs.append(bytes(bad))
    bad = bytearray(good)
    bad[pos + nb + 3 + 1] = 0xf0                    # second section, first octet: 16 nibbles in a u32 vector
    vecs.append(bytes(bad))
    vecs.append(good[:10])                          # truncated header
    vecs.append(good[:len(good) - 40])              # truncated body
    xor = oracle.encode_all(oracle.F32, (np.arange(300) / 1.3).astype(np.float32))
    bad = bytearray(xor)
    bad[5] = 0x11                                   # an Xor section inside a vector that claims u32
    vecs.append(bytes(bad))
    vecs.append(good)
    host = cv.EncodedBatch.from_vectors("u32", vecs)
    dev = host.to_device()
    total, per, st = dev.filter_count(3, per_vector=True, return_status=True)
    wtotal, wper, wst, _ = oracle.batch_filter_count(oracle.U32, host.blob, host.byte_off, 3)
    assert list(st) == list(wst)
    assert np.array_equal(per, wper)
    assert total == wtotal
    assert st[0] == 0 and st[-1] == 0 and st[1] == 7 and st[2] == 4 and st[3] == 5 and st[6] == 5
    # the payloads of the reference's CodingError (src/error.rs:2-11) and where each iteration stopped
    det = dev.error_details(st)
    for v, vec in enumerate(vecs):
        rc, k, payload = oracle.error_detail(oracle.U32, vec)
        assert rc == st[v] and (int(det[v, 0]), int(det[v, 1])) == (k, payload), v
    assert tuple(det[1]) == (2**64 - 1, 0x10)         # WrongVectorType(0x10), found in the vector header
    assert tuple(det[2]) == (1, 9)                    # InvalidSectionType(9) at the second section
    assert int(det[3, 0]) == 1 and int(det[3, 1]) == 16 + nb      # second section fails to decode: its offset
    assert not det[0].any() and not det[-1].any()
    _, _, dst = dev.decode(return_status=True)
    assert list(dst) == list(wst)
    assert np.array_equal(dev.error_details(dst), det)


def test_total_only_pass_equals_the_per_vector_pass(cv, oracle):
    """A pass that asks for nothing but the result block takes the section kernel's own sums as the total when no
    vector header and no section failed, and folds per vector otherwise.  Both ways must give the result block of
    a pass with per-vector outputs: on a clean batch, on batches with a damaged header, a damaged section, a
    section of an illegal kind - and again on the clean batch afterwards (the failure mark of an earlier pass
    must not stick to the workspace)."""
    rng = np.random.default_rng(5)
    good = [oracle.encode_all(oracle.U32, rng.integers(0, 9, size=int(rng.integers(1, 3000))).astype(np.uint32))
            for _ in range(40)]

    def blocks(vecs):
        dev = cv.EncodedBatch.from_vectors("u32", vecs).to_device()
        r_total = dev.filter_count_async(3).cpu().numpy().copy()
        pv = torch.empty(dev.n_vectors, dtype=torch.int64, device=dev.device)
        st = torch.empty(dev.n_vectors, dtype=torch.int32, device=dev.device)
        r_full = dev.filter_count_async(3, pv, st).cpu().numpy().copy()
        r_again = dev.filter_count_async(3).cpu().numpy().copy()
        d_total = dev.decode_async(torch.zeros(dev.n_sections * 256, dtype=torch.int32, device=dev.device)).cpu().numpy().copy()
        d_full = dev.decode_async(torch.zeros(dev.n_sections * 256, dtype=torch.int32, device=dev.device), status=st).cpu().numpy().copy()
        assert np.array_equal(r_total[:3], r_full[:3]) and np.array_equal(r_again[:3], r_full[:3]), (r_total, r_full, r_again)
        assert np.array_equal(d_total[:3], d_full[:3]), (d_total, d_full)
        host = dev.download()
        wtotal, _, wst, _ = oracle.batch_filter_count(oracle.U32, host.blob, host.byte_off, 3)
        assert int(r_total[0]) == wtotal and int(r_total[1]) == int((np.asarray(wst) != 0).sum())
        return r_total

    import torch
    clean = blocks(good)
    assert clean[1] == 0
    bad_header = bytearray(good[7]); bad_header[5] = 0x10
    assert blocks(good[:7] + [bytes(bad_header)] + good[8:])[1] == 1
    big = max(range(len(good)), key=lambda i: len(good[i]))
    _, nb = oracle.section_parse(oracle.U32, good[big][16:])
    bad_octet = bytearray(good[big]); bad_octet[16 + nb + 3 + 1] = 0xf0      # second section: an octet of 16 nibbles
    assert blocks(good[:big] + [bytes(bad_octet)] + good[big + 1:])[1] == 1
    xor = bytearray(oracle.encode_all(oracle.F32, (np.arange(600) / 1.3).astype(np.float32))); xor[5] = 0x11
    assert blocks(good[:3] + [bytes(xor)] + good[3:])[1] == 1              # Xor sections in a u32 vector
    short = good[11][:len(good[11]) - 1] if len(good[11]) > 40 else good[12][:len(good[12]) - 1]
    assert blocks(good[:20] + [short] + good[20:])[1] == 1                 # truncated body
    again = blocks(good)
    assert np.array_equal(again[:3], clean[:3])


@pytest.mark.parametrize("tname", ["u32", "f32", "u64"])
def test_big_ragged_batch_with_damage_vs_oracle(cv, oracle, tname):
    """More than 65536 sections: the section kernel hands out its slot ranges on demand (its waves draw them from
    counters and move on to other counters at the end).  40000 ragged vectors, every tenth one with damaged bytes:
    statuses, per-vector counts, masks and decoded values of every vector must equal the oracle's, the total-only
    pass must equal the per-vector pass, and a second pass must give the same again."""
    rng = np.random.default_rng(404)
    et = ET[tname]
    uniq = []
    for i in range(1500):
        flat, off = _random_vectors(rng, tname, 1, max_len=int(rng.integers(200, 2500)))
        vec = bytearray(oracle.encode_all(et, flat))
        if i % 10 == 3 and len(vec) > 24:
            for _ in range(int(rng.integers(1, 4))):
                at = int(rng.integers(0, len(vec))) if rng.random() < 0.5 else 16 + int(rng.integers(0, min(12, len(vec) - 16)))
                vec[at] = int(rng.integers(0, 256))
        uniq.append(bytes(vec))
    pick = rng.integers(0, len(uniq), size=40000)
    host = cv.EncodedBatch.from_vectors(tname, [uniq[i] for i in pick])
    dev = host.to_device()
    assert dev.n_sections > 65536
    pred = NPD[tname](0)
    # the oracle once per distinct vector
    want = {}
    for i in set(int(x) for x in pick):
        blob = np.frombuffer(uniq[i], dtype=np.uint8)
        bo = np.array([0, blob.size], dtype=np.uint64)
        wtotal, wper, wst, _ = oracle.batch_filter_count(et, blob, bo, pred)
        want[i] = (int(wper[0]), int(wst[0]))
    sb = dev.d_sec_base.cpu().numpy().view(np.uint64)
    for rep in range(2):
        total, per, st, masks = dev.filter_count(pred, per_vector=True, return_status=True, return_masks=True)
        assert int(total) == int(np.asarray(per).sum())
        wper = np.array([want[int(i)][0] for i in pick], dtype=np.uint64)
        wst = np.array([want[int(i)][1] for i in pick], dtype=np.int32)
        ok = np.asarray(st) != 100            # (a damaged element count is reported as a section-count mismatch)
        assert np.array_equal(np.asarray(st)[ok], wst[ok]) and np.array_equal(np.asarray(per)[ok], wper[ok])
        assert int((np.asarray(st) != 0).sum()) > 1000
        r = dev.filter_count_async(pred).cpu().numpy()
        assert int(r[0]) == int(total) and int(r[1]) == int((np.asarray(st) != 0).sum())
    # masks and decoded values of a sample of healthy vectors
    dec, doff, dst = dev.decode(return_status=True)
    assert np.array_equal(np.asarray(dst)[ok], wst[ok])
    for v in rng.integers(0, len(pick), size=300):
        v = int(v)
        if st[v] != 0:
            continue
        blob = np.frombuffer(uniq[int(pick[v])], dtype=np.uint8)
        _, wm = oracle.filter_count(et, blob, pred, return_masks=True)
        assert np.array_equal(np.asarray(masks[32 * int(sb[v]):32 * int(sb[v + 1])]), np.asarray(wm, dtype=np.uint8).reshape(-1)), v
        wdec, _, rc = oracle.batch_decode(et, blob, np.array([0, blob.size], dtype=np.uint64), np.array([0, 256 * int(sb[v + 1] - sb[v])], dtype=np.uint64))
        a, b = int(doff[v]), int(doff[v + 1])
        assert rc == 0 and np.array_equal(_bits(np.asarray(dec[a:b])), _bits(np.asarray(wdec)))


def test_damaged_element_count_is_bounded_by_the_bytes(cv, oracle):
    """A header whose num_elements (src/vector.rs:165) claims far more sections than the vector's bytes
    could hold must not size any buffer: the vector gets one slot more than its bytes allow, reports a
    section-count mismatch, and its neighbours decode and count as usual."""
    good = oracle.encode_all(oracle.U32, (np.arange(600, dtype=np.uint32) % 5) + 1)
    bad = bytearray(good)
    bad[8:12] = (0xffffffff).to_bytes(4, "little")
    host = cv.EncodedBatch.from_vectors("u32", [good, bytes(bad), good])
    dev = host.to_device()
    sb = dev.d_sec_base.cpu().numpy()
    assert int(sb[1] - sb[0]) == 3 and int(sb[2] - sb[1]) == len(good) - 16 + 1 and int(sb[3] - sb[2]) == 3
    total, per, st = dev.filter_count(3, per_vector=True, return_status=True)
    assert list(st) == [0, 100, 0]
    assert per[0] == per[2] == oracle.filter_count(oracle.U32, good, 3)
    dec, off, dst = dev.decode(return_status=True)
    assert list(dst) == [0, 100, 0] and dec.size == 256 * int(sb[3])
    want = oracle.decode_to_vec(oracle.U32, good)
    assert np.array_equal(dec[:768], want) and np.array_equal(dec[int(off[2]):int(off[2]) + 768], want)
    vals, hoff, hst = cv.decode_host(host, return_status=True)
    assert list(hst) == [0, 100, 0] and np.array_equal(hoff, off) and np.array_equal(vals[:768], want)


def _tool(name):
    """Imports tools/<name>.py (the soak programs; the tests run a short seeded stretch of each)."""
    import importlib
    import os
    import sys
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools")
    if d not in sys.path:
        sys.path.insert(0, d)
    return importlib.import_module(name)


def test_randomly_damaged_batches_vs_oracle(cv, oracle):
    """Random ragged batches with 0, 1, 3 or 12 damaged bytes; statuses, partial counts, masks and
    decoded values of every vector against the oracle."""
    fz = _tool("fuzz_parity")
    for seed in range(5000, 5120):
        for tname in ("u32", "u64", "f32"):
            assert fz.one_round(seed, tname, [0, 1, 3, 12][seed % 4]) == [], (seed, tname)
    assert fz.STATS["bad_status"] > 50 and fz.STATS["details"] > 50


def test_encoder_on_adversarial_values_vs_oracle(cv, oracle):
    """Extremes of the integer range, ranges that need every bit, chosen zero nibbles, NaN payloads,
    signed zeros, infinities, denormals: device-encoded bytes equal the oracle's, and decode back."""
    fz = _tool("fuzz_encode")
    for seed in range(7000, 7100):
        for tname in ("u32", "u64", "f32"):
            assert fz.one_round(seed, tname)[0] == [], (seed, tname)
            assert fz.appender_round(seed, tname) == [], (seed, tname)   # append / append_nulls / reader / finish


def test_streams_and_columns_randomised_vs_oracle(cv, oracle):
    """Stream codecs on adversarial values and on damaged or cut packed bytes, the multi-column filter over
    columns whose vectors differ in length, and series of histograms (some damaged) through the
    delta-diff re-packer, against the oracle."""
    fz = _tool("fuzz_streams")
    for seed in range(9000, 9060):
        for kind in ("u64", "u64_delta", "f64_xor"):
            assert fz.streams_round(seed, kind) == [], (seed, kind)
        assert fz.columns_round(seed) == [], seed
        assert fz.hist_round(seed) == [], seed
    assert fz.STATS["gated_rounds"] > 10 and fz.STATS["hist_failed"] > 10


def test_resident_index_gives_the_same_results(cv, oracle):
    """build_index() walks the section headers once; later ungated filter passes reuse the descriptors
    and must return exactly what the full path returns (damaged vectors included); a gated pass in
    between overwrites the index and the next ungated pass rebuilds it."""
    rng = np.random.default_rng(41)
    for tname in ("u32", "u64", "f32"):
        vals, eo = _random_vectors(rng, tname, 120)
        blob, bo = oracle.batch_encode(ET[tname], vals, eo)
        blob = np.array(blob, dtype=np.uint8)
        for v in (5, 17, 60):                               # damage: a type byte, a length, an octet header
            lo = int(bo[v])
            if int(bo[v + 1]) - lo > 40:
                blob[lo + 16 + (0, 1, 4)[v % 3]] ^= 0x5a
        host = cv.EncodedBatch(tname, blob, np.asarray(bo, dtype=np.uint64))
        plain, indexed = host.to_device(), host.to_device().build_index()
        nz = vals[vals != 0]
        for p in (0, nz[3], [nz[1], 0, nz[7]]):
            a = plain.filter_count(p, per_vector=True, return_status=True, return_masks=True)
            b = indexed.filter_count(p, per_vector=True, return_status=True, return_masks=True)
            assert indexed._index_valid
            assert a[0] == b[0] and all(np.array_equal(x, y) for x, y in zip(a[1:], b[1:])), (tname, p)
            w = oracle.batch_filter_count(ET[tname], blob, bo, p)
            assert list(b[2]) == list(w[2]) and np.array_equal(b[1], w[1])
        # a gated pass over the index reads the gate in the section kernel: same masks as the gated full
        # path, and the index stays as it is
        gate = plain.filter_masks(0)
        g_idx, g_full = indexed.filter_masks(nz[3], gate=gate), plain.filter_masks(nz[3], gate=gate)
        assert indexed._index_valid and not plain._index_valid
        assert np.array_equal(g_idx.masks.cpu().numpy(), g_full.masks.cpu().numpy())
        assert np.array_equal(g_idx.valid.cpu().numpy(), g_full.valid.cpu().numpy())
        assert np.array_equal(g_idx.result.cpu().numpy(), g_full.result.cpu().numpy())
        a = plain.filter_count(nz[3], per_vector=True, return_status=True)
        b = indexed.filter_count(nz[3], per_vector=True, return_status=True)
        assert indexed._index_valid and a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
        b = indexed.filter_count(nz[3], per_vector=True, return_status=True)
        assert a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
        # decode leaves a valid index behind too
        d1 = plain.decode(return_status=True)
        d2 = indexed.decode(return_status=True)
        assert np.array_equal(_bits(d1[0]), _bits(d2[0])) and np.array_equal(d1[2], d2[2])
        b = indexed.filter_count(0, per_vector=True, return_status=True)
        a = plain.filter_count(0, per_vector=True, return_status=True)
        assert a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    # an empty batch
    e = cv.EncodedBatch.from_vectors("u32", []).to_device().build_index()
    assert e.filter_count(3) == 0
    # a workspace that does not hold this batch's index (never indexed, or indexed for another batch of the
    # same shape) is refused on the device, not followed: every vector reports the workspace error
    import torch
    other = host.to_device().build_index()
    for ws in (torch.zeros_like(other.workspace()), other.workspace()):
        stranger = host.to_device()
        stranger._ws, stranger._keep_index, stranger._index_valid = ws, True, True
        total, per, st = stranger.filter_count(0, per_vector=True, return_status=True)
        assert total == 0 and not per.any() and set(st.tolist()) == {104}
        with pytest.raises(cv.CodingError):
            stranger.filter_count(0)
    assert other.filter_count(0, return_status=True)[0] == plain.filter_count(0, return_status=True)[0]   # the owner's index is untouched


def test_overwide_shift_octets_take_the_exact_path(cv, oracle):
    """Hand-made 32-bit sections whose octets carry 8 or more trailing zero nibbles (an encoder never
    writes them: the lanes shift out to 0, src/nibblepack_simd.rs:581) must decode like the oracle;
    the kernel's common path only detects them and a second exact pass handles them."""
    def section(kind, octets):
        body = b"".join(octets) + b"\x00" * (32 - len(octets))
        if kind == 1:                               # NibblePack: length excludes the 3-byte header
            return bytes([1]) + len(body).to_bytes(2, "little") + body
        return bytes([6]) + (len(body) + 3).to_bytes(2, "little") + body   # Xor: length includes it
    octs = [bytes([0b00000101, 0x19, 0xab, 0xcd]),   # n = 2, t = 9: both stored lanes shift out
            bytes([0xff, 0x07] + [0x21, 0x43, 0x65, 0x87]),   # n = 1, t = 7: ordinary
            bytes([0b10000001, 0x3c, 0x34, 0x12, 0x78, 0x56])]  # n = 4, t = 12
    for tname, kind in (("u32", 1), ("f32", 6)):
        sect = section(kind, octs)
        body = sect + sect                         # two sections: 512 elements
        hdr = (len(body) + 12).to_bytes(4, "little") + bytes([0x10, ET[tname]]) + b"\x00\x00" + \
            (512).to_bytes(4, "little") + b"\x00\x00\x00\x00"
        vec = hdr + body
        want = oracle.decode_to_vec(ET[tname], vec)
        plain = oracle.encode_all(ET[tname], np.arange(700).astype(NPD[tname]))
        host = cv.EncodedBatch.from_vectors(tname, [plain, vec, plain])
        dev = host.to_device()
        dec, off, st = dev.decode(return_status=True)
        assert list(st) == [0, 0, 0]
        assert np.array_equal(_bits(dec[int(off[1]):int(off[1]) + 512]), _bits(want))
        for pred in (0, want[8] if want[8] != 0 else 1):
            total, per, st = dev.filter_count(pred, per_vector=True, return_status=True)
            wtotal, wper, wst, _ = oracle.batch_filter_count(ET[tname], host.blob, host.byte_off, pred)
            assert list(st) == list(wst) and np.array_equal(per, wper) and total == wtotal


def test_multi_vector_filter_and_positions_vs_oracle(cv, oracle):
    """MultiVectorFilter + count_hits (src/filter.rs:239-299) and match_positions over device-resident masks."""
    rng = np.random.default_rng(77)
    n = 60
    lens = rng.integers(1, 1500, size=n)
    cols_vals, blobs = [], []
    for c in range(3):
        vecs = []
        for v in range(n):
            ln = int(lens[v]) if c < 2 else max(1, int(lens[v]) - 300 * (v % 5 == 0))   # column 2: some shorter vectors
            x = rng.integers(0, 4, size=ln).astype(np.uint32)
            if v % 7 == c:
                x[:] = 0                                                             # Null sections: short-circuit rows
            vecs.append(oracle.encode_all(oracle.U32, x))
        blobs.append(vecs)
    preds = [1, 2, 1]
    devs = [cv.EncodedBatch.from_vectors("u32", b).to_device() for b in blobs]
    cols = [d.filter_masks(p) for d, p in zip(devs, preds)]
    total, per_row, masks, row_sections = cv.multi_filter_count(cols, per_row=True, return_masks=True)
    sb0 = devs[0].d_sec_base.cpu().numpy()
    want_total = 0
    for v in range(n):
        cnt, wm = oracle.multi_filter_count(oracle.U32, [blobs[c][v] for c in range(3)], preds, return_masks=True)
        want_total += cnt
        assert per_row[v] == cnt, v
        ns = wm.size // 32
        assert row_sections[v] == ns, v
        assert np.array_equal(masks[32 * int(sb0[v]):32 * (int(sb0[v]) + ns)], wm), v
    assert total == want_total
    # the API mirror: two filters over one vector each
    r0, r1 = cv.VectorReader.try_new(blobs[0][3], "u32"), cv.VectorReader.try_new(blobs[1][3], "u32")
    mf = cv.MultiVectorFilter([r0.filter_iter(cv.EqualsSink("u32", 1)), r1.filter_iter(cv.EqualsSink("u32", 2))])
    assert cv.count_hits(mf) == oracle.multi_filter_count(oracle.U32, [blobs[0][3], blobs[1][3]], [1, 2])
    # match_positions of a single column, every vector
    pos, off = cols[0].match_positions()
    for v in range(n):
        _, wm = oracle.filter_count(oracle.U32, blobs[0][v], preds[0], return_masks=True)
        want = oracle.match_positions(wm)
        assert np.array_equal(pos[int(off[v]):int(off[v + 1])].astype(np.uint64), want), v
    # left on the device, twice into the same buffers
    dpos, doff = cols[0].match_positions_async()
    dpos.fill_(-1)
    dpos2, doff2 = cols[0].match_positions_async(dpos, doff)
    assert dpos2 is dpos and doff2 is doff
    assert np.array_equal(doff.cpu().numpy().view(np.uint64), off)
    assert np.array_equal(dpos[:int(off[-1])].cpu().numpy().view(np.uint32), pos)


@pytest.mark.parametrize("tname", ["u32", "u64", "f32"])
def test_gated_columns_equal_ungated_and_oracle(cv, oracle, tname):
    """Columns behind a selective one are gated by its masks (sections it ruled out are parsed, not
    decoded, src/filter.rs:277-282): totals, per-row counts and ANDed masks equal the ungated path and
    the oracle; a decode error inside a gated-off section goes unnoticed, as in the reference."""
    rng = np.random.default_rng(300 + ET[tname])
    n, length = 400, 1536                      # same layout in every column: 6 sections per vector
    def column(p_hit):
        x = rng.integers(1, 6, size=(n, length)).astype(NPD[tname])
        hit = rng.random((n, length // 256)) < p_hit             # sections that contain the value 9 at all
        x[np.repeat(hit, 256, axis=1) & (rng.random((n, length)) < 0.3)] = 9
        return x
    cols_np = [column(0.08), column(0.5), column(0.5)]
    eo = np.arange(n + 1, dtype=np.uint64) * length
    devs = [cv.encode_batch(tname, c.reshape(-1), eo).to_device() for c in cols_np]
    preds = [9, 9, 3]
    total, per_row, masks, rows = cv.multi_filter(devs, preds, per_row=True, return_masks=True)
    plain = [d.filter_masks(p) for d, p in zip(devs, preds)]
    t2, per2, masks2, rows2 = cv.multi_filter_count(plain, per_row=True, return_masks=True)
    assert total == t2 and np.array_equal(per_row, per2) and np.array_equal(masks, masks2) and np.array_equal(rows, rows2)
    want = (cols_np[0] == 9) & (cols_np[1] == 9) & (cols_np[2] == 3)
    assert total == int(want.sum()) and np.array_equal(per_row, want.sum(axis=1).astype(np.uint64))
    host = [d.download() for d in devs]
    for v in (0, 7, n - 1):
        cnt = oracle.multi_filter_count(ET[tname], [h.vector(v) for h in host], preds)
        assert cnt == per_row[v]
    # the gate really switches sections off: the gated second column matches less often than the plain one
    g1 = devs[1].filter_masks(preds[1], gate=plain[0])
    assert int(g1.result[0].item()) < int(plain[1].result[0].item())
    assert torch_equal_where_gate_open(g1.masks, plain[1].masks, plain[0].masks)
    # damage an octet of a section of column 1 that column 0 rules out: unnoticed behind the gate
    m0 = plain[0].masks.cpu().numpy().reshape(-1, 32)
    closed = np.flatnonzero(~m0.any(axis=1))
    s = int(closed[len(closed) // 2])
    v, k = divmod(s, length // 256)
    h1 = host[1]
    vec = bytearray(h1.vector(v))
    pos = 16
    for _ in range(k):
        pos += oracle.section_parse(ET[tname], bytes(vec[pos:]))[1]
    kind, nb = oracle.section_parse(ET[tname], bytes(vec[pos:]))
    hdr = 12 if kind == 3 else 3
    vec[pos + hdr + 1] = 0xf0                                      # 16 nibbles per value: an invalid octet
    vecs = [h1.vector(i) for i in range(n)]
    vecs[v] = bytes(vec)
    bad = cv.EncodedBatch.from_vectors(tname, vecs).to_device()
    _, _, st = bad.filter_count(preds[1], per_vector=True, return_status=True)
    if tname != "u64":
        assert st[v] != 0                                          # the plain pass does see it
    tg, perg = cv.multi_filter([devs[0], bad, devs[2]], preds, per_row=True)
    assert tg == total and np.array_equal(perg, per_row)
    # the same columns with resident section indexes: the gate is applied inside the section kernel
    for d in devs + [bad]:
        d.build_index()
    t3, per3, masks3, rows3 = cv.multi_filter(devs, preds, per_row=True, return_masks=True)
    assert t3 == total and np.array_equal(per3, per_row) and np.array_equal(masks3, masks) and np.array_equal(rows3, rows)
    g1i = devs[1].filter_masks(preds[1], gate=plain[0])
    assert np.array_equal(g1i.masks.cpu().numpy(), g1.masks.cpu().numpy())
    tg, perg = cv.multi_filter([devs[0], bad, devs[2]], preds, per_row=True)
    assert tg == total and np.array_equal(perg, per_row)


def torch_equal_where_gate_open(got, plain, gate):
    g, p, m = (x.cpu().numpy().reshape(-1, 32) for x in (got, plain, gate))
    open_ = m.any(axis=1)
    return np.array_equal(g[open_], p[open_]) and not g[~open_].any()


def test_legacy_stream_codecs_vs_oracle(cv, oracle):
    """pack_u64_delta / DeltaSink and pack_f64_xor / DoubleXorSink (src/nibblepacking.rs:15-45, :239-325,
    :392-400) over batches of ragged streams: packed bytes and unpacked values equal the oracle's."""
    rng = np.random.default_rng(19)
    # --- delta: increasing counters (with a few decreases, which the packer clips to 0, :18) -----------
    streams = []
    for n in [1, 7, 8, 9, 255, 256, 257, 1000, 5000, 0, 33]:
        inc = rng.integers(0, 1 << int(rng.integers(1, 40)), size=n).astype(np.uint64)
        if n > 20:
            inc[rng.integers(0, n, size=n // 4)] = 0
        v = np.cumsum(inc, dtype=np.uint64)
        if n > 50:
            v[n // 2] = v[n // 2 - 1] - np.uint64(1)                     # a decrease
        streams.append(v)
    streams.append(np.array([1, 1 << 63, (1 << 64) - 1], dtype=np.uint64))
    eo = np.concatenate([[0], np.cumsum([s.size for s in streams])]).astype(np.uint64)
    vals = np.concatenate(streams) if streams else np.zeros(0, np.uint64)
    blob, bo, st = cv.stream.pack_batch("u64_delta", vals, eo)
    assert not st.any()
    for i, sv in enumerate(streams):
        want = oracle.pack_u64_delta(sv)
        assert blob[int(bo[i]):int(bo[i + 1])].tobytes() == bytes(want), i
    got, st = cv.stream.unpack_batch("u64_delta", blob, bo, eo)
    assert not st.any()
    for i, sv in enumerate(streams):
        want = oracle.unpack_u64_delta(blob[int(bo[i]):int(bo[i + 1])], sv.size)[0][:sv.size] if sv.size else sv
        assert np.array_equal(got[int(eo[i]):int(eo[i + 1])], want), i
    # --- f64 XOR -----------------------------------------------------------------------------------------
    fstreams = []
    for n in [1, 2, 8, 9, 10, 257, 258, 3000]:
        x = np.cumsum(rng.normal(size=n)) + 100.0
        if n > 100:
            x[50:80] = x[49]                                             # repeated values: zero octets
        fstreams.append(x.astype(np.float64))
    fstreams.append(np.array([0.0, -0.0, np.nan, np.inf, 1e-310], dtype=np.float64))
    feo = np.concatenate([[0], np.cumsum([s.size for s in fstreams])]).astype(np.uint64)
    fvals = np.concatenate(fstreams)
    blob, bo, st = cv.stream.pack_batch("f64_xor", fvals, feo)
    assert not st.any()
    for i, sv in enumerate(fstreams):
        assert blob[int(bo[i]):int(bo[i + 1])].tobytes() == bytes(oracle.pack_f64_xor(sv)), i
    got, st = cv.stream.unpack_batch("f64_xor", blob, bo, feo)
    assert not st.any()
    assert got.view(np.uint64).tobytes() == fvals.view(np.uint64).tobytes()          # bitwise round trip
    for i, sv in enumerate(fstreams):
        want = oracle.unpack_f64_xor(blob[int(bo[i]):int(bo[i + 1])], sv.size)[0][:sv.size]
        assert np.array_equal(got[int(feo[i]):int(feo[i + 1])].view(np.uint64), want.view(np.uint64)), i
    # --- errors: an empty XOR stream cannot be packed (:37); truncated streams report NotEnoughSpace ------
    _, _, st = cv.stream.pack_batch("f64_xor", np.zeros(3), np.array([0, 0, 3], dtype=np.uint64))
    assert list(st) == [2, 0]
    cut = np.concatenate([blob[int(bo[7]):int(bo[8]) - 5], blob[int(bo[2]):int(bo[3])]])
    cbo = np.array([0, int(bo[8] - bo[7]) - 5, int(bo[8] - bo[7]) - 5 + int(bo[3] - bo[2])], dtype=np.uint64)
    ceo = np.array([0, fstreams[7].size, fstreams[7].size + fstreams[2].size], dtype=np.uint64)
    got, st = cv.stream.unpack_batch("f64_xor", cut, cbo, ceo)
    assert st[0] == 1 and st[1] == 0
    assert np.array_equal(got[int(ceo[1]):].view(np.uint64), fstreams[2].view(np.uint64))
    # the single-stream mirror of the reference's function names
    enc = cv.stream.pack_u64_delta([0, 1000, 1001, 1002, 1003, 2005, 2010, 3034, 4045, 5056, 6067, 7078])
    assert list(cv.stream.unpack_u64_delta(enc, 12)) == [0, 1000, 1001, 1002, 1003, 2005, 2010, 3034, 4045, 5056, 6067, 7078]


def test_histogram_delta_diff_pack_vs_oracle(cv, oracle, kats):
    """DeltaDiffPackSink (src/histogram.rs:92-171) over series of histograms, and
    compress_geom_nonincreasing (:55-76), against the oracle; the reference's own test data first."""
    hist = cv.histogram
    d = kats["histogram"]["delta_diffpack"]
    inputs = [np.array(x, dtype=np.uint64) for x in d["inputs"]]
    sink, osink = hist.DeltaDiffPackSink(12), oracle.DeltaDiffPackSink(12)
    for inp in inputs:
        packed = oracle.pack_u64_delta(inp)
        want, wdrop, _ = osink.feed(packed)
        assert sink.feed(packed) == want and sink.value_dropped == wdrop
    assert hist.compress_geom_nonincreasing(64, 2.0, 2.0, hist.FORMAT_GEOMETRIC_DELTA, inputs[1]) == \
        oracle.compress_geom_nonincreasing(64, 2.0, 2.0, 1, inputs[1])
    assert cv.stream.pack_u64(inputs[2]) == oracle.pack_u64(inputs[2])
    assert np.array_equal(cv.stream.unpack_u64(oracle.pack_u64(inputs[2]), 12), inputs[2])
    rng = np.random.default_rng(99)
    for nb in (1, 7, 8, 64, 247, 248, 256, 300, 700):
        # series of different lengths (one empty), buckets that grow, stall and sometimes reset
        lens = [5, 0, 1, 9, 3]
        series_off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
        packed, want_out, want_drop = [], [], []
        for ln in lens:
            osink = oracle.DeltaDiffPackSink(nb)
            counts = np.zeros(nb, dtype=np.uint64)
            for t in range(ln):
                counts = counts + rng.integers(0, 1 << int(rng.integers(1, 40)), nb).astype(np.uint64) * (rng.random(nb) < 0.6)
                if rng.random() < 0.25:
                    counts[rng.integers(0, nb)] = 0                       # a counter reset
                deltas = np.diff(np.concatenate([[0], np.sort(counts)])).astype(np.uint64)
                if rng.random() < 0.3:
                    deltas = counts                                       # not increasing bucket to bucket
                p = oracle.pack_u64(deltas)
                packed.append(p)
                o, dr, used = osink.feed(p)
                assert used == len(p)
                want_out.append(o)
                want_drop.append(dr)
        off = np.concatenate([[0], np.cumsum([len(p) for p in packed])]).astype(np.uint64)
        blob = np.frombuffer(b"".join(packed), dtype=np.uint8)
        out, oo, dropped, st = hist.delta_diff_batch(blob, off, series_off, nb)
        assert not st.any()
        assert [out[int(oo[h]):int(oo[h + 1])].tobytes() for h in range(len(packed))] == want_out
        assert list(dropped) == want_drop
    # a histogram cut short fails, and so do the later ones of its series; other series are untouched
    nb = 64
    good = [oracle.pack_u64(rng.integers(1, 1 << 20, nb).astype(np.uint64)) for _ in range(6)]
    cut = list(good)
    cut[1] = good[1][:len(good[1]) - 9]
    off = np.concatenate([[0], np.cumsum([len(p) for p in cut])]).astype(np.uint64)
    out, oo, dropped, st = hist.delta_diff_batch(np.frombuffer(b"".join(cut), dtype=np.uint8), off, [0, 3, 6], nb)
    assert st[0] == 0 and st[1] != 0 and st[2] == st[1] and not st[3:].any()
    assert oo[2] == oo[1] and oo[3] == oo[2]
    osink = oracle.DeltaDiffPackSink(nb)
    assert [out[int(oo[h]):int(oo[h + 1])].tobytes() for h in (3, 4, 5)] == [osink.feed(good[h])[0] for h in (3, 4, 5)]


@pytest.mark.parametrize("tname", ["u32", "u64", "f32"])
def test_large_ragged_batch_vs_oracle(cv, oracle, tname):
    """Tens of thousands of sections: every wave of the section kernel runs many chunks, chunk boundaries
    fall everywhere in the vectors, and 64-bit worst-case sections force short chunks."""
    rng = np.random.default_rng(4000 + ET[tname])
    vals, eo = _random_vectors(rng, tname, 5000, max_len=3000)
    blob, bo = oracle.batch_encode(ET[tname], vals, eo)
    got = cv.encode_batch(tname, vals, eo)
    assert np.array_equal(got.byte_off, bo) and np.array_equal(got.blob, blob)
    dev = got.to_device()
    dec, out_off = dev.decode()
    want, st, rc = oracle.batch_decode(ET[tname], blob, bo, out_off)
    assert rc == 0 and np.array_equal(_bits(dec), _bits(want))
    nz = vals[vals != 0]
    for p in (0, nz[7], [nz[1], nz[2], 0]):
        total, per = dev.filter_count(p, per_vector=True)
        wtotal, wper, wst, rc = oracle.batch_filter_count(ET[tname], blob, bo, p)
        assert rc == 0 and total == wtotal and np.array_equal(per, wper)


@pytest.mark.parametrize("chunk_bytes", ["3000", "70000", None])
def test_host_buffer_entry_points_chunked_vs_oracle(cv, oracle, monkeypatch, chunk_bytes):
    """cvec_{encode,decode,filter_count}_batch on host arrays: the library pipelines chunks of whole
    vectors through a few device slots; tiny chunks (many per batch, some a single vector), medium
    ones and the default (one chunk) must all give the oracle's bytes, values, counts and statuses."""
    if chunk_bytes is None:
        monkeypatch.delenv("CVEC_PIPE_CHUNK_BYTES", raising=False)
    else:
        monkeypatch.setenv("CVEC_PIPE_CHUNK_BYTES", chunk_bytes)
    for tname in ("u32", "u64", "f32"):
        rng = np.random.default_rng(77 + ET[tname])
        vals, eo = _random_vectors(rng, tname, 300, max_len=2500)
        blob, bo = oracle.batch_encode(ET[tname], vals, eo)
        got = cv.encode_host(tname, vals, eo)
        assert np.array_equal(got.byte_off, bo) and np.array_equal(got.blob, blob)
        dec, out_off, st = cv.decode_host(got, add_const=None if tname == "u64" else 3, return_status=True)
        want, wst, rc = oracle.batch_decode(ET[tname], blob, bo, out_off, add_base=None if tname == "u64" else 3)
        assert rc == 0 and not st.any() and np.array_equal(_bits(dec), _bits(want))
        nz = vals[vals != 0]
        for p in (0, nz[3], [nz[1], 0, nz[5]]):
            total, per = cv.filter_count_host(got, p, per_vector=True)
            wtotal, wper, _, rc = oracle.batch_filter_count(ET[tname], blob, bo, p)
            assert rc == 0 and total == wtotal and np.array_equal(per, wper)
    # a damaged vector in the middle: its status and the partial counts travel through the chunking
    rng = np.random.default_rng(5)
    vals, eo = _random_vectors(rng, "u32", 40, max_len=1200)
    blob, bo = oracle.batch_encode(ET["u32"], vals, eo)
    bad = blob.copy()
    v = 17
    bad[int(bo[v]) + 16] = 0x09                      # first section type byte of vector 17: not a section type
    total, per, st = cv.filter_count_host(cv.EncodedBatch("u32", bad, bo), 0, per_vector=True, return_status=True)
    wtotal, wper, wst, rc = oracle.batch_filter_count(ET["u32"], bad, bo, 0)
    assert total == wtotal and np.array_equal(per, wper) and np.array_equal(st, wst) and st[v] != 0
    # pinned caller buffers take the same path
    pv, pe = cv.hostpath.pin(vals), cv.hostpath.pin(eo)
    assert np.array_equal(cv.encode_host("u32", pv, pe).blob, blob)
    # empty batch
    e = cv.encode_host("u32", np.zeros(0, dtype=np.uint32), np.zeros(1, dtype=np.uint64))
    assert e.n_vectors == 0 and e.blob.size == 0 and cv.filter_count_host(e, 1) == 0


@pytest.mark.parametrize("tname", ["u32", "u64", "f32"])
def test_vectors_far_beyond_64k_bytes_vs_oracle(cv, oracle, tname):
    """Vectors of hundreds of sections (several hundred KiB each): section offsets and remaining
    lengths do not fit 16 bits (the reference's length checks wrap there, src/section.rs:537, :616,
    :694; this library compares in 64 bits) and one vector spans many waves of the section kernel."""
    rng = np.random.default_rng(900 + ET[tname])
    lens = [300_000, 1, 70_001, 256 * 300]
    parts = []
    for ln in lens:
        if tname == "f32":
            x = (rng.integers(0, 2000, ln) / np.float32(7)).astype(np.float32)
        else:
            x = (rng.integers(0, 1 << 20, ln) + (1 << 33 if tname == "u64" else 0)).astype(NPD[tname])
        x[ln // 3:ln // 3 + 700] = 0                                  # a run of Null sections inside
        parts.append(x)
    vals = np.concatenate(parts)
    eo = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    blob, bo = oracle.batch_encode(ET[tname], vals, eo)
    assert int(bo[1] - bo[0]) > 65536
    got = cv.encode_batch(tname, vals, eo)
    assert np.array_equal(got.byte_off, bo) and np.array_equal(got.blob, blob)
    dev = got.to_device()
    dec, out_off = dev.decode()
    for v, ln in enumerate(lens):
        assert np.array_equal(_bits(dec[int(out_off[v]):int(out_off[v]) + ln]), _bits(parts[v]))
    for p in (0, parts[0][5]):
        total, per = dev.filter_count(p, per_vector=True)
        wtotal, wper, _, rc = oracle.batch_filter_count(ET[tname], blob, bo, p)
        assert rc == 0 and total == wtotal and np.array_equal(per, wper)


def test_resident_batch_grows_by_appended_vectors(cv, oracle):
    """A batch in HBM extended twice by newly encoded vectors equals the encode of all vectors at once
    (bytes, offsets, section bases) and filters / decodes like it."""
    import torch
    rng = np.random.default_rng(4242)
    vals, eo = _random_vectors(rng, "u32", 90, max_len=1400)
    cuts = [0, 25, 26, 90]
    dev = None
    for a, b in zip(cuts[:-1], cuts[1:]):
        part = torch.from_numpy(vals[int(eo[a]):int(eo[b])].view(np.int32)).cuda()
        peo = torch.from_numpy((eo[a:b + 1] - eo[a]).view(np.int64)).cuda()
        if part.numel() == 0:
            part = torch.zeros(4, dtype=torch.int32, device="cuda")
        dev = cv.encode_device("u32", part, peo, int(eo[b] - eo[a])) if dev is None else dev.append_values(part, peo)
    blob, bo = oracle.batch_encode(ET["u32"], vals, eo)
    host = dev.download()
    assert np.array_equal(host.byte_off, bo) and np.array_equal(host.blob, blob)
    whole = cv.EncodedBatch("u32", blob, bo).to_device()
    assert torch.equal(dev.d_sec_base, whole.d_sec_base) and dev.n_sections == whole.n_sections
    nz = vals[vals != 0]
    total, per = dev.filter_count(nz[2], per_vector=True)
    wtotal, wper, _, rc = oracle.batch_filter_count(ET["u32"], blob, bo, nz[2])
    assert rc == 0 and total == wtotal and np.array_equal(per, wper)
    dec, off = dev.decode()
    want, _, rc = oracle.batch_decode(ET["u32"], blob, bo, off)
    assert rc == 0 and np.array_equal(dec, want)


@pytest.mark.parametrize("name", ["u32 dense", "u32 sparse", "f32 xor 60%", "u64 delta"])
def test_reference_bench_configurations_vs_oracle(cv, oracle, name):
    """The reference's `cargo bench vector` workloads (benches/bench.rs:131-170: 10000-element vectors;
    dense u32 sinusoid, the same with all but two runs of 50 values nulled, f32 at 60 % density / 1.4,
    u64 = dense + 100000) at a size the oracle finishes in seconds: encoded bytes, filter + count,
    decode through AddConstSink -> VecSink and the encode_all -> decode round trip against the oracle."""
    from compressed_vec_amd import synth
    nvec, length = 48, synth.VECTOR_LENGTH
    tname, kind, pred, base = {"u32 dense": ("u32", "u32_dense", 3, 7), "u32 sparse": ("u32", "u32_dense", 15, 7),
                               "f32 xor 60%": ("f32", "f32_xor", np.float32(3) / np.float32(1.4), 2.5),
                               "u64 delta": ("u64", "u64_delta", 100003, 5)}[name]
    vals, eo = synth.batch_values(kind, nvec, length)
    if name == "u32 sparse":                                    # benches/bench.rs:160-170
        keep = np.zeros(length, dtype=bool)
        keep[2475:2525] = keep[7475:7525] = True
        vals = np.where(np.tile(keep, nvec), vals, 0).astype(np.uint32)
    want_blob, want_off = oracle.batch_encode(ET[tname], vals, eo)
    got = cv.encode_batch(tname, vals, eo)
    assert np.array_equal(got.byte_off, want_off) and np.array_equal(got.blob, want_blob)
    dev = got.to_device()
    total, per = dev.filter_count(pred, per_vector=True)
    wtotal, wper, wst, rc = oracle.batch_filter_count(ET[tname], want_blob, want_off, pred)
    assert rc == 0 and total == wtotal and np.array_equal(per, wper)
    assert total == int((vals == NPD[tname](pred)).sum())       # 10000 = 39 sections + 16 elements: the padding is zero
    dec, off = dev.decode(add_const=base)
    wdec, _, rc = oracle.batch_decode(ET[tname], want_blob, want_off, off, add_base=base)
    assert rc == 0 and np.array_equal(_bits(dec), _bits(wdec))
    plain, _ = dev.decode()
    for v in (0, nvec // 2, nvec - 1):
        a = plain[int(off[v]):int(off[v]) + length]
        assert np.array_equal(_bits(a), _bits(vals[v * length:(v + 1) * length]))
    if name == "u32 sparse":
        kinds = oracle.reader_sections(ET[tname], want_blob[int(want_off[0]):int(want_off[1])])
        assert (kinds == 0).sum() >= 36                         # Null sections: the short-circuit path


@pytest.mark.parametrize("top,lo,hi", [(200, 0, 378), (1 << 15, 379, 700)])
def test_small_region_shape_is_chosen_by_section_size_and_agrees(cv, oracle, top, lo, hi):
    """Count-only passes pick 6 KiB staging regions (four workgroups per CU) or 12 KiB ones (three) when 16
    average sections fit them, 18 KiB ones otherwise; a batch of small sections with a few vectors of
    incompressible ones mixed in runs a small shape with chunks that break early, and must count like the
    oracle either way."""
    rng = np.random.default_rng(77)
    small = [rng.integers(1, top, size=int(rng.integers(200, 3000))).astype(np.uint32) for _ in range(300)]
    big = [rng.integers(0, 1 << 32, size=2048, dtype=np.uint64).astype(np.uint32) for _ in range(6)]
    vecs = small[:100] + big[:3] + small[100:] + big[3:]
    eo = np.concatenate([[0], np.cumsum([v.size for v in vecs])]).astype(np.uint64)
    vals = np.concatenate(vecs)
    blob, bo = oracle.batch_encode(oracle.U32, vals, eo)
    dev = cv.EncodedBatch("u32", blob, bo).to_device()
    assert lo <= dev.n_bytes // dev.n_sections <= hi           # the launcher's rule picks the 6 KiB / the 12 KiB shape
    for p in (7, [7, 9, 150], int(big[0][5])):
        total, per = dev.filter_count(p, per_vector=True)
        wtotal, wper, _, rc = oracle.batch_filter_count(oracle.U32, blob, bo, p)
        assert rc == 0 and total == wtotal and np.array_equal(per, wper), p
    onlybig = cv.encode_batch("u32", np.concatenate(big), np.arange(7, dtype=np.uint64) * 2048).to_device()
    assert onlybig.n_bytes / onlybig.n_sections > 1000         # the large shape
    assert onlybig.filter_count(int(big[1][9])) == int((np.concatenate(big) == big[1][9]).sum())


@pytest.mark.parametrize("regime", ["small", "middle", "large"])
@pytest.mark.parametrize("tname", ["u32", "u64", "f32"])
def test_wide_decode_stagings_vs_oracle(cv, oracle, tname, regime):
    """The decode pass picks its loop by the batch's average section size: wide stagings (up to 64 sections staged
    at once and worked off in rounds of 16) for small and, with 32-bit elements, for large sections, stagings of 16
    with paired 16-byte stores (32-bit) or plain ones (64-bit) in between.  Section sizes are mixed so that stagings
    end after every possible number of sections (tiny sections in runs, dense ones, incompressible vectors in
    between, ragged vector ends, Null and Const sections), bytes are damaged in a few vectors, and both tail rules
    run; values, statuses and per-vector element counts against the oracle."""
    et, npd = ET[tname], NPD[tname]
    rng = np.random.default_rng(4242 + et + len(regime))
    # which of the ten value kinds below a vector takes: mostly narrow ones, an even mix, mostly wide ones
    kinds = {"small": (0, 0, 0, 0, 0, 1, 1, 1, 2, 3), "middle": (0, 1, 2, 3, 4, 5, 6, 7, 8, 9), "large": (8, 9, 8, 9, 8, 9, 9, 9, 7, 0)}[regime]
    parts = []
    for v in range(2600):
        n = int(rng.integers(1, 1800))
        kind = kinds[int(rng.integers(0, 10))]
        if tname == "f32":
            if kind < 5:
                x = (rng.integers(0, 4, size=n) * 0.5).astype(np.float32)            # few distinct bit patterns
                if regime == "small":
                    x = np.repeat(x[: n // 40 + 1], 40)[:n]                               # in runs: mostly empty octets
            elif kind < 8:
                x = rng.integers(0, 1 << 12, size=n).astype(np.float32)
            else:
                x = rng.normal(size=n).astype(np.float32)                               # incompressible
        else:
            top = (1 << 4, 1 << 4, 1 << 9, 1 << 9, 1 << 14, 1 << 14, 1 << 20, 1 << 20, 1 << 31, 1 << 31)[kind]
            x = rng.integers(0, top, size=n).astype(npd)
            if kind == 9 and tname == "u64" and v % 3 == 0:
                x = rng.integers(0, 1 << 63, size=n, dtype=np.uint64)
            if v % 6 == 0:
                x += npd(100000)
        if v % 9 == 0:
            x[: min(n, 512)] = 0                                                        # Null sections in front
        if v % 13 == 0 and n > 300:
            x[-300:] = x[-301]                                                          # a Const tail
        parts.append(x)
    eo = np.concatenate([[0], np.cumsum([p.size for p in parts])]).astype(np.uint64)
    vals = np.concatenate(parts)
    blob, bo = oracle.batch_encode(et, vals, eo)
    blob = blob.copy()
    for v in (3, 500, 501, 1999, 2599):                                                 # damage a byte of five vectors
        a, b = int(bo[v]), int(bo[v + 1])
        if b - a > 60:
            blob[a + (b - a) * 2 // 3] ^= 0x5a
    dev = cv.EncodedBatch(tname, blob, bo).to_device()
    # the launcher's rule: wide stagings up to 260 B per section (64-bit: 380) and, 32-bit only, above 760 B
    avg = dev.n_bytes // dev.n_sections
    assert dev.n_sections > 6000
    if regime == "small":
        assert avg <= 260
    elif regime == "middle":
        assert 380 < avg <= 760
    else:
        assert avg > 760
    base = {"u32": 4000000001, "u64": (1 << 63) + 77, "f32": -1.25}[tname]
    for rule in ("format", "reference"):
        for add in (None, base):
            got, out_off, st = dev.decode(add_const=add, return_status=True, tail_rule=rule)
            if rule == "reference":
                with oracle.ref_tail_quirk():
                    want, wst, _rc = oracle.batch_decode(et, blob, bo, out_off, add_base=add)
            else:
                want, wst, _rc = oracle.batch_decode(et, blob, bo, out_off, add_base=add)
            assert np.array_equal(st, wst), (rule, add)
            ok = np.flatnonzero(wst == 0)
            assert ok.size >= 2590
            for v in ok[:: 7]:
                a, b = int(out_off[v]), int(out_off[v + 1])
                assert np.array_equal(_bits(got[a:b]), _bits(want[a:b])), (rule, add, int(v))
            # every good vector, in one comparison
            keep = np.zeros(got.size, dtype=bool)
            for v in ok:
                keep[int(out_off[v]):int(out_off[v + 1])] = True
            assert np.array_equal(_bits(got)[keep], _bits(want)[keep]), (rule, add)


def test_batches_of_millions_of_vectors_take_the_long_scan(cv):
    """The exclusive scans behind sec_base, the encoder's offsets and match_positions add up block sums inside
    their second kernel for up to 2048 blocks of 2048 entries and scan them in a kernel of their own beyond:
    4.3 million one-element vectors reach the second form; counts, decoded values and positions against the
    values."""
    import torch
    n = 4_300_000
    vals = (torch.arange(n, device="cuda", dtype=torch.int64) * 2654435761 % 7).to(torch.int32)
    eo = torch.arange(n + 1, device="cuda", dtype=torch.int64)
    dev = cv.encode_device("u32", vals, eo)
    assert dev.n_vectors == n and dev.n_sections == n
    want = int((vals == 3).sum().item())
    total, per = dev.filter_count(3, per_vector=True)
    assert total == want and int(np.asarray(per).sum()) == want
    out = torch.empty(n * 256, dtype=torch.int32, device="cuda")
    dev.decode_async(out)
    torch.cuda.synchronize()
    assert torch.equal(out.view(n, 256)[:, 0], vals) and int(out.view(n, 256)[:, 1:].abs().sum().item()) == 0
    del out
    pos, off = dev.filter_masks(3).match_positions_async()
    off = off.cpu().numpy()
    assert int(off[-1]) == want and int(pos[:want].abs().sum().item()) == 0      # every match is element 0 of its vector
    hit = np.flatnonzero((vals == 3).cpu().numpy())
    assert np.array_equal(np.flatnonzero(np.diff(off.astype(np.int64))), hit)


def test_full_size_properties_cfg5_shape(cv):
    """1 Mi elements per call at the benchmark's vector shape (2048-element f32 vectors): encode →
    decode is the identity on bits, counts are additive over predicates that partition the values,
    and per-vector counts sum to the total."""
    import torch
    from compressed_vec_amd import synth
    nvec, length = 4096, 2048
    vals, eo = synth.batch_values_torch("f32_xor", nvec, length)
    dev = cv.encode_device("f32", vals, eo)
    out = torch.zeros(dev.n_sections * 256, dtype=torch.float32, device="cuda")
    dev.decode_async(out)
    torch.cuda.synchronize()
    assert torch.equal(out.view(torch.int32), vals.view(torch.int32))
    distinct = torch.unique(vals).cpu().numpy()
    total = 0
    for d in distinct:
        total += dev.filter_count(float(d))
    assert total == nvec * length
    t, per = dev.filter_count(float(distinct[3]), per_vector=True)
    assert int(per.sum()) == t
    assert dev.filter_count([float(x) for x in distinct[:8]]) == sum(dev.filter_count(float(x)) for x in distinct[:8])


def test_pipeline_error_exit_drains_and_the_next_call_is_clean(cv, oracle, monkeypatch):
    """A *_batch call that fails at a chunk in the middle (injected after that chunk's copies and kernels
    were queued) must return the error only when nothing of it is in flight any more: the next call on
    the same thread reuses the slots at once and has to produce the oracle's results; a too small output
    buffer is a call-level error (>= 101), never a per-vector status."""
    from compressed_vec_amd import _native as N
    monkeypatch.setenv("CVEC_PIPE_CHUNK_BYTES", "20000")
    rng = np.random.default_rng(99)
    vals, eo = _random_vectors(rng, "u32", 200, max_len=2500)
    blob, bo = oracle.batch_encode(ET["u32"], vals, eo)
    host = cv.EncodedBatch("u32", cv.hostpath.pin(blob), bo)
    for fail_at in (0, 3, 7):
        for op in ("decode", "filter", "encode"):
            assert cv.lib().cvec_debug_fail_chunk(fail_at) == 0
            with pytest.raises(cv.CodingError) as e:
                if op == "decode":
                    cv.decode_host(host)
                elif op == "filter":
                    cv.filter_count_host(host, 5)
                else:
                    cv.encode_host("u32", vals, eo)
            assert e.value.code == N.ERR_HIP
            # straight into the next call, same slots
            dec, out_off = cv.decode_host(host)
            want, _, rc = oracle.batch_decode(ET["u32"], blob, bo, out_off)
            assert rc == 0 and np.array_equal(dec, want)
            total, per = cv.filter_count_host(host, 5, per_vector=True)
            wtotal, wper, _, rc = oracle.batch_filter_count(ET["u32"], blob, bo, 5)
            assert rc == 0 and total == wtotal and np.array_equal(per, wper)
            got = cv.encode_host("u32", vals, eo)
            assert np.array_equal(got.blob, blob) and np.array_equal(got.byte_off, bo)
    assert cv.lib().cvec_debug_fail_chunk(-1) == 0
    small = np.zeros(256, dtype=np.uint32)
    with pytest.raises(cv.CodingError) as e:
        cv.decode_host(host, out=small, return_status=True)
    assert e.value.code == N.ERR_CAPACITY


def test_legacy_streams_many_short_streams_vs_oracle(cv, oracle):
    """Batches of thousands of streams take the one-pass scan (a wave per stream, the running sum / XOR carried from
    block to block): ragged lengths around the 256-value block size, empty streams, against the oracle's packers and
    the values themselves."""
    from compressed_vec_amd import stream
    rng = np.random.default_rng(41)
    n = 2600
    lens = rng.integers(0, 900, size=n)
    lens[::97] = 0
    lens[1::97] = 256
    lens[2::97] = 257
    eo = np.zeros(n + 1, dtype=np.uint64)
    eo[1:] = np.cumsum(lens)
    total = int(eo[-1])
    inc = rng.integers(0, 1 << 20, size=total).astype(np.uint64)
    inc[rng.random(total) < 0.3] = 0
    u = np.concatenate([np.cumsum(inc[int(eo[s]):int(eo[s + 1])]) for s in range(n)] + [np.zeros(0, np.uint64)]).astype(np.uint64)
    blob, bo, st = stream.pack_batch("u64_delta", u, eo)
    assert not st.any()
    for s in range(0, n, 131):
        assert blob[int(bo[s]):int(bo[s + 1])].tobytes() == bytes(oracle.pack_u64_delta(u[int(eo[s]):int(eo[s + 1])]))
    got, st = stream.unpack_batch("u64_delta", blob, bo, eo)
    assert not st.any() and np.array_equal(got, u)
    # f64 XOR: a stream needs at least one value
    lens_f = np.maximum(lens, 1)
    eof = np.zeros(n + 1, dtype=np.uint64)
    eof[1:] = np.cumsum(lens_f)
    f = (rng.standard_normal(int(eof[-1])) * 1e3).astype(np.float64)
    f[rng.random(f.size) < 0.2] = 0.0
    blob, bo, st = stream.pack_batch("f64_xor", f, eof)
    assert not st.any()
    for s in range(0, n, 131):
        assert blob[int(bo[s]):int(bo[s + 1])].tobytes() == bytes(oracle.pack_f64_xor(f[int(eof[s]):int(eof[s + 1])]))
    got, st = stream.unpack_batch("f64_xor", blob, bo, eof)
    assert not st.any() and np.array_equal(np.asarray(got).view(np.uint64), f.view(np.uint64))
