"""GPU parity of the series appender (cvec_series_appender_*, through the C ABI): n appenders driven side by side
on the device against n oracle appenders (the CPU restatement of VectorAppender, src/vector.rs:207-379) that
receive the same calls.  Bit-exact bytes, element counts and null-section counts."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

NPD = {"u32": np.uint32, "u64": np.uint64, "f32": np.float32}


@pytest.fixture(scope="module")
def cv():
    import torch
    assert torch.cuda.is_available(), "these tests need a GPU"
    import compressed_vec_amd as cv
    assert cv.lib().cvec_init(0) == 0
    return cv


def _values(rng, tname, n):
    mode = int(rng.integers(0, 5))
    if tname == "f32":
        if mode == 0:
            return np.zeros(n, dtype=np.float32)
        if mode == 1:
            return np.full(n, 2.5, dtype=np.float32)
        return (rng.standard_normal(n) * 10.0 ** int(rng.integers(-3, 6))).astype(np.float32)
    dt = NPD[tname]
    if mode == 0:
        return np.zeros(n, dtype=dt)
    if mode == 1:
        return np.full(n, 77, dtype=dt)
    if mode == 2:
        return (rng.integers(0, 50, size=n) + (1 << 20)).astype(dt)
    hi = 2 ** (31 if tname == "u32" else 62)
    return rng.integers(0, hi, size=n).astype(dt)


def _same_vectors(host, oracles, total=None, reader=False):
    for v, o in enumerate(oracles):
        want = o.reader_bytes() if reader else o.finish(total)
        assert host.vector(v) == bytes(want), "series %d" % v


@pytest.mark.parametrize("tname", ["u32", "u64", "f32"])
def test_series_appender_ragged_appends_and_nulls_vs_oracle(cv, oracle, tname):
    """Ragged appends, ragged and uniform nulls, readers in between, finish with padding; then the handle again."""
    rng = np.random.default_rng({"u32": 5, "u64": 6, "f32": 7}[tname])
    n = 37
    et = {"u32": oracle.U32, "u64": oracle.U64, "f32": oracle.F32}[tname]
    app = cv.SeriesAppender(tname, n)
    for round_ in range(2):
        oracles = [oracle.Appender(et, 64) for _ in range(n)]
        seen = [[] for _ in range(n)]
        for step in range(9):
            kind = step % 4
            if kind in (0, 1, 3):
                counts = rng.integers(0, 400 if kind != 3 else 3, size=n)
                counts[rng.integers(0, n)] = 0
                off = np.zeros(n + 1, dtype=np.int64)
                off[1:] = np.cumsum(counts)
                vals = np.concatenate([_values(rng, tname, int(c)) for c in counts]) if off[-1] else np.zeros(0, NPD[tname])
                app.append(vals, off)
                for v, o in enumerate(oracles):
                    o.append_many(vals[off[v]:off[v + 1]])
                    seen[v].append(vals[off[v]:off[v + 1]])
            else:
                counts = rng.integers(0, 900, size=n)
                counts[::5] = 0
                app.append_nulls(counts)
                for v, o in enumerate(oracles):
                    if counts[v]:
                        o.append_nulls(int(counts[v]))
                    seen[v].append(np.zeros(int(counts[v]), NPD[tname]))
            if step in (3, 7):
                _same_vectors(app.reader().download(), oracles, reader=True)
        app.append_nulls(300)
        for o in oracles:
            o.append_nulls(300)
        longest = max(o.num_elements() for o in oracles)
        with pytest.raises(cv.CodingError) as e:
            app.finish(longest - 1)
        assert e.value.name == "InvalidNumRows" and e.value.detail == (longest - 1, longest)
        total = longest + (0 if round_ else 700)
        batch = app.finish(total)
        _same_vectors(batch.download(), oracles, total=total)
        assert app.stats()["segments"] == 0            # finish() leaves the appender empty
        # what came out is a batch like any other: decode it and compare with the values that went in
        per = (total + 255) // 256
        assert batch.n_sections == n * per
        dec = batch.decode()[0].reshape(n, per * 256)
        bits = np.uint32 if tname != "u64" else np.uint64
        for v in range(n):
            went_in = np.concatenate(seen[v] + [np.zeros(0, NPD[tname])])
            assert np.array_equal(dec[v, :went_in.size].view(bits), went_in.view(bits)) and not dec[v, went_in.size:].any()


def test_series_appender_uniform_scrapes_vs_oracle(cv, oracle):
    """The scrape pattern: every series receives k values per call; sections complete for all series at once."""
    import torch
    rng = np.random.default_rng(11)
    n, k, calls = 300, 48, 13                          # 624 values per series: two complete sections + 112 open
    app = cv.SeriesAppender("f32", n)
    oracles = [oracle.Appender(oracle.F32, 64) for _ in range(n)]
    base = rng.standard_normal(n).astype(np.float32) * 100
    blocks = []
    for c in range(calls):
        block = (base[:, None] + np.sin(np.arange(c * k, (c + 1) * k, dtype=np.float32) / 9.0)[None, :]).astype(np.float32)
        blocks.append(block)
        app.append(torch.from_numpy(block).cuda())
        for v, o in enumerate(oracles):
            o.append_many(block[v])
    st = app.stats()
    assert st["segments"] == 2                          # calls 6 and 11 complete a section of every series
    _same_vectors(app.reader().download(), oracles, reader=True)
    batch = app.finish(calls * k)
    _same_vectors(batch.download(), oracles, total=calls * k)
    # the batch is usable as it is
    got = batch.decode()[0].reshape(n, -1)[:, :calls * k]
    assert got.shape == (n, calls * k)
    assert np.array_equal(np.ascontiguousarray(got).view(np.uint32), np.concatenate(blocks, 1).view(np.uint32))
    assert int(batch.filter_count(float(got[3, 5]))) == int((got == got[3, 5]).sum())


def test_series_appender_empty_and_argument_errors(cv, oracle):
    app = cv.SeriesAppender("u32", 5)
    empty = app.finish(0).download()
    o = oracle.Appender(oracle.U32, 64)
    want = bytes(o.finish(0))
    assert all(empty.vector(v) == want for v in range(5))
    assert app.reader().n_sections == 0
    with pytest.raises(cv.CodingError):
        app.append(np.zeros((4, 3), dtype=np.uint32))                     # wrong number of series
    with pytest.raises(cv.CodingError):
        app.append(np.zeros(3, dtype=np.uint32), [0, 1, 2, 3, 4, 5])      # offsets reach past the values
    with pytest.raises(cv.CodingError):
        app.append(np.zeros((5, 3), dtype=np.float32))                    # element type
    with pytest.raises(cv.CodingError) as e:
        app.finish(2 ** 32)
    assert e.value.detail == (2 ** 32, 2 ** 32 - 1)
    zero = cv.SeriesAppender("u64", 0)
    assert zero.finish(10).n_vectors == 0


def test_series_appender_at_shard_width_equals_one_shot_encode(cv):
    """Size-independent property at the headline vector length: 16384 series x 2048 f32 values, appended 96 per call
    (sections complete in the middle of calls), finish() == encode of the whole matrix at once, byte for byte,
    offsets and section bases included; the handle, used again, gives the same batch."""
    import torch
    from compressed_vec_amd import synth
    n, length, k = 16384, 2048, 96
    mat = torch.cat([synth.batch_values_torch("f32_xor", 4096, length, v_first=c0)[0] for c0 in range(0, n, 4096)]).reshape(n, length)
    eo = torch.arange(n + 1, device="cuda", dtype=torch.int64) * length
    want = cv.encode_device("f32", mat.reshape(-1), eo)
    app = cv.SeriesAppender("f32", n)
    for _ in range(2):
        for c0 in range(0, length, k):
            app.append(mat[:, c0:c0 + k].contiguous())
        got = app.finish(length)
        assert got.n_bytes == want.n_bytes and got.n_sections == want.n_sections
        assert torch.equal(got.d_bytes[:got.n_bytes], want.d_bytes[:want.n_bytes])
        assert torch.equal(got.d_byte_off, want.d_byte_off) and torch.equal(got.d_sec_base, want.d_sec_base)


def test_series_appender_long_null_runs_cost_a_count_not_their_length(cv, oracle):
    """append_nulls and finish padding far beyond what the values take: whole null sections are one byte each in the
    vector (src/vector.rs:320-327) and a count per series in the handle."""
    rng = np.random.default_rng(23)
    n = 19
    app = cv.SeriesAppender("u64", n)
    oracles = [oracle.Appender(oracle.U64, 64) for _ in range(n)]
    first = rng.integers(0, 300, size=n)
    off = np.zeros(n + 1, dtype=np.int64)
    off[1:] = np.cumsum(first)
    vals = rng.integers(1, 1 << 40, size=int(off[-1])).astype(np.uint64)
    app.append(vals, off)
    for v, o in enumerate(oracles):
        o.append_many(vals[off[v]:off[v + 1]])
    runs = rng.integers(0, 40_000_000, size=n)
    runs[3], runs[4] = 0, 255
    app.append_nulls(runs)
    for v, o in enumerate(oracles):
        if runs[v]:
            o.append_nulls(int(runs[v]))
    held = app.stats()
    assert held["encoded_bytes"] < 1 << 20                  # not the gigabytes the zeros would take
    tail = rng.integers(1, 1 << 20, size=(n, 70)).astype(np.uint64)
    app.append(tail)
    for v, o in enumerate(oracles):
        o.append_many(tail[v])
    _same_vectors(app.reader().download(), oracles, reader=True)
    total = 3_000_000_000                                    # finish pads every series to three billion rows
    batch = app.finish(total)
    assert batch.n_sections == n * ((total + 255) // 256)
    _same_vectors(batch.download(), oracles, total=total)


@pytest.mark.parametrize("tname", ["u32", "f32"])
def test_series_appender_merges_segments_of_ragged_appends(cv, oracle, tname):
    """Hundreds of ragged append calls that each complete sections for a few series only: the handle merges sixteen
    segments of one level into one of the next, so it holds a handful of segments instead of one per call, and a
    long run of null sections in the middle stays a count (what lies in front of it is merged into one segment that
    is never touched again).  Readers at several points and the finished vectors against the oracle's appenders."""
    rng = np.random.default_rng({"u32": 31, "f32": 32}[tname])
    n = 150
    et = {"u32": oracle.U32, "f32": oracle.F32}[tname]
    app = cv.SeriesAppender(tname, n)
    oracles = [oracle.Appender(et, 64) for _ in range(n)]
    most = 0
    for step in range(620):
        if step == 300:
            runs = rng.integers(0, 5_000_000, size=n)        # a long run of nulls: pinned, not written out
            app.append_nulls(runs)
            for v, o in enumerate(oracles):
                if runs[v]:
                    o.append_nulls(int(runs[v]))
        elif step % 37 == 5:
            runs = rng.integers(0, 700, size=n) * (rng.random(n) < 0.2)   # short runs for a fifth of the series
            app.append_nulls(runs)
            for v, o in enumerate(oracles):
                if runs[v]:
                    o.append_nulls(int(runs[v]))
        else:
            counts = np.where(rng.random(n) < 0.15, rng.integers(1, 200, size=n), rng.integers(0, 3, size=n))
            off = np.zeros(n + 1, dtype=np.int64)
            off[1:] = np.cumsum(counts)
            vals = np.concatenate([_values(rng, tname, int(c)) for c in counts]) if off[-1] else np.zeros(0, NPD[tname])
            app.append(vals, off)
            for v, o in enumerate(oracles):
                o.append_many(vals[off[v]:off[v + 1]])
        most = max(most, app.stats()["segments"])
        if step in (40, 299, 301, 560):
            _same_vectors(app.reader().download(), oracles, reader=True)
    assert most <= 3 * 15 + 4, most                           # 15 per level at most, the pinned ones, not one per call
    total = max(o.num_elements() for o in oracles) + 1000
    _same_vectors(app.finish(total).download(), oracles, total=total)
    assert app.stats()["segments"] == 0


def test_series_appender_host_values_equal_device_values(cv, oracle):
    """cvec_series_appender_append_host: the same calls with the values in host memory."""
    rng = np.random.default_rng(31)
    n = 23
    dev, host = cv.SeriesAppender("u32", n), cv.SeriesAppender("u32", n)
    oracles = [oracle.Appender(oracle.U32, 64) for _ in range(n)]
    for step in range(6):
        if step % 2 == 0:
            block = rng.integers(0, 1 << 20, size=(n, int(rng.integers(1, 400)))).astype(np.uint32)
            dev.append(block)
            host.append_host(block)
            for v, o in enumerate(oracles):
                o.append_many(block[v])
        else:
            counts = rng.integers(0, 500, size=n)
            off = np.zeros(n + 1, dtype=np.uint64)
            off[1:] = np.cumsum(counts)
            vals = rng.integers(0, 1 << 31, size=int(off[-1])).astype(np.uint32)
            dev.append(vals, off.astype(np.int64))
            host.append_host(vals, off)
            for v, o in enumerate(oracles):
                o.append_many(vals[int(off[v]):int(off[v + 1])])
    with pytest.raises(cv.CodingError):
        host.append_host(np.zeros(3, np.uint32), np.array([0, 2, 1] + [3] * (n - 2), dtype=np.uint64))   # offsets go down
    total = max(o.num_elements() for o in oracles)
    a, b = dev.finish(total).download(), host.finish(total).download()
    _same_vectors(a, oracles, total=total)
    assert all(a.vector(v) == b.vector(v) for v in range(n))
