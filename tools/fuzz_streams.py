"""Randomised soak of the stream codecs (pack_u64, pack_u64_delta, pack_f64_xor and their unpackers,
src/nibblepacking.rs:15-45, :239-400) and of the multi-column filter against the CPU oracle.

Streams: ragged batches of adversarial values; packed bytes and unpacked values must equal the oracle's;
then the packed bytes are damaged and cut, and every stream must fail exactly when the oracle fails and
give the oracle's values when it does not.  Columns: two to four columns of ragged u32 vectors with
section-aligned and unaligned length differences; totals, per-row counts and masks against the oracle.
Histograms: random series through DeltaDiffPackSink, some packed histograms damaged or cut.
Usage: python tools/fuzz_streams.py [--seconds 60] [--seed0 0]
"""
import argparse
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

cv = importlib.import_module("compressed-vec_amd")
import oracle  # noqa: E402
from fuzz_encode import int_piece  # noqa: E402

STATS = {"gated_rounds": 0}
PACK = {"u64": oracle.pack_u64, "u64_delta": oracle.pack_u64_delta, "f64_xor": oracle.pack_f64_xor}
UNPACK = {"u64": oracle.unpack_u64, "u64_delta": oracle.unpack_u64_delta, "f64_xor": oracle.unpack_f64_xor}


def stream_values(rng, kind, n):
    if kind == "f64_xor":
        mode = int(rng.integers(0, 5))
        if mode == 0:
            x = np.cumsum(rng.normal(size=n)) + 100.0
        elif mode == 1:
            x = np.full(n, rng.normal())
        elif mode == 2:
            x = rng.integers(0, 1 << 63, size=n, dtype=np.uint64).view(np.float64)       # raw patterns
        elif mode == 3:
            x = rng.choice(np.array([0.0, -0.0, np.nan, np.inf, -np.inf, 1e-310, 1.0]), size=n)
        else:
            x = np.round(rng.normal(size=n) * 100) / 4
        return np.ascontiguousarray(x, dtype=np.float64)
    parts, left = [], n
    while left > 0:
        k = int(min(left, rng.integers(1, 300)))
        parts.append(int_piece(rng, k, 64))
        left -= k
    v = np.concatenate(parts) if parts else np.zeros(0, dtype=np.uint64)
    if kind == "u64_delta" and rng.random() < 0.7:
        v = np.cumsum(v >> np.uint64(int(rng.integers(8, 60))), dtype=np.uint64)        # mostly increasing
    return v.astype(np.uint64)


def streams_round(seed, kind):
    rng = np.random.default_rng(seed)
    ns = int(rng.integers(1, 25))
    lens = [int(rng.integers(1 if kind == "f64_xor" else 0, int(rng.integers(2, 1500)))) for _ in range(ns)]
    streams = [stream_values(rng, kind, n) for n in lens]
    eo = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    vals = np.concatenate(streams)
    errs = []
    blob, bo, st = cv.stream.pack_batch(kind, vals, eo)
    if np.any(st):
        return [("pack status", list(st))]
    want = [PACK[kind](s) for s in streams]
    for i in range(ns):
        if blob[int(bo[i]):int(bo[i + 1])].tobytes() != bytes(want[i]):
            errs.append(("packed bytes", i, lens[i]))
            return errs
    got, st = cv.stream.unpack_batch(kind, blob, bo, eo)
    if np.any(st) or got.view(np.uint64).tobytes() != np.ascontiguousarray(_expected_roundtrip(kind, streams)).view(np.uint64).tobytes():
        errs.append(("round trip",))
    # damage: flip bytes, cut tails
    pieces = []
    for i in range(ns):
        b = bytearray(want[i])
        how = int(rng.integers(0, 4))
        if how == 1 and len(b):
            b[int(rng.integers(0, len(b)))] ^= 1 << int(rng.integers(0, 8))
        elif how == 2 and len(b):
            b = b[:int(rng.integers(0, len(b)))]
        elif how == 3 and len(b):
            for _ in range(3):
                b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
        pieces.append(bytes(b))
    dbo = np.concatenate([[0], np.cumsum([len(p) for p in pieces])]).astype(np.uint64)
    dblob = np.frombuffer(b"".join(pieces), dtype=np.uint8).copy()
    if dblob.size == 0:
        return errs
    got, st = cv.stream.unpack_batch(kind, dblob, dbo, eo)
    for i in range(ns):
        a, b = int(eo[i]), int(eo[i + 1])
        try:
            w = UNPACK[kind](pieces[i], lens[i])[0][:lens[i]] if lens[i] else np.zeros(0, dtype=np.uint64)
            wrc = 0
        except oracle.OracleError as e:
            wrc = e.code
        if (st[i] != 0) != (wrc != 0):
            errs.append(("damaged status", i, int(st[i]), wrc, lens[i], len(pieces[i])))
        elif wrc == 0 and got[a:b].view(np.uint64).tobytes() != np.ascontiguousarray(w).view(np.uint64).tobytes():
            errs.append(("damaged values", i, lens[i]))
        elif wrc != 0 and int(st[i]) != int(wrc):
            errs.append(("damaged status code", i, int(st[i]), wrc))
    return errs


def _expected_roundtrip(kind, streams):
    if kind != "u64_delta":
        return np.concatenate(streams)
    out = []
    for s in streams:                                      # decreases are clipped to a zero delta (:18)
        if s.size == 0:
            out.append(s)
            continue
        o = np.empty_like(s)
        last = 0
        acc = 0
        for i, x in enumerate(s.tolist()):
            d = x - last if x >= last else 0
            acc = (acc + d) & ((1 << 64) - 1)
            o[i] = acc
            last = x
        out.append(o)
    return np.concatenate(out)


def columns_round(seed):
    rng = np.random.default_rng(seed)
    n = int(rng.integers(1, 30))
    ncol = int(rng.integers(2, 5))
    lens = rng.integers(1, 1800, size=n)
    ragged = rng.random() < 0.5          # otherwise every column has the same layout and the gated path runs
    blobs = []
    for c in range(ncol):
        vecs = []
        for v in range(n):
            ln = int(lens[v])
            if ragged and c and rng.random() < 0.2:
                ln = max(1, ln - int(rng.integers(1, 600)))                    # a shorter vector in this column
            x = rng.integers(0, int(rng.integers(2, 5)), size=ln).astype(np.uint32)
            for a in range(0, ln, 256):
                if rng.random() < 0.25:
                    x[a:a + 256] = int(rng.integers(0, 3)) if rng.random() < 0.5 else 0   # Const / Null sections
            vecs.append(oracle.encode_all(oracle.U32, x))
        blobs.append(vecs)
    preds = [int(rng.integers(0, 3)) for _ in range(ncol)]
    devs = [cv.EncodedBatch.from_vectors("u32", b).to_device() for b in blobs]
    errs = []
    cols = [d.filter_masks(p) for d, p in zip(devs, preds)]
    total, per_row, masks, rows = cv.multi_filter_count(cols, per_row=True, return_masks=True)
    sb0 = devs[0].d_sec_base.cpu().numpy()
    want_total = 0
    for v in range(n):
        cnt, wm = oracle.multi_filter_count(oracle.U32, [blobs[c][v] for c in range(ncol)], preds, return_masks=True)
        want_total += cnt
        ns = wm.size // 32
        if per_row[v] != cnt or rows[v] != ns:
            errs.append(("row", v, int(per_row[v]), cnt, int(rows[v]), ns))
        elif not np.array_equal(masks[32 * int(sb0[v]):32 * (int(sb0[v]) + ns)], wm):
            errs.append(("row masks", v))
    if total != want_total:
        errs.append(("total", int(total), want_total))
    same_layout = all(np.array_equal(d.d_sec_base.cpu().numpy(), sb0) for d in devs)
    if same_layout:                                                            # the gated path needs equal layouts
        t2, per2, m2, r2 = cv.multi_filter(devs, preds, per_row=True, return_masks=True)
        if t2 != total or not np.array_equal(per2, per_row) or not np.array_equal(m2, masks) or not np.array_equal(r2, rows):
            errs.append(("gated differs",))
        for d in devs:                                                         # and over resident section indexes
            d.build_index()
        t3, per3, m3, r3 = cv.multi_filter(devs, preds, per_row=True, return_masks=True)
        if t3 != total or not np.array_equal(per3, per_row) or not np.array_equal(m3, masks) or not np.array_equal(r3, rows):
            errs.append(("gated over index differs",))
        STATS["gated_rounds"] += 1
    return errs


def hist_round(seed):
    """DeltaDiffPackSink over random series of histograms (src/histogram.rs:92-171), some of them damaged or
    cut: outputs, value_dropped and which histograms fail (a failure takes the rest of its series along)."""
    rng = np.random.default_rng(seed)
    hist = cv.histogram
    nb = int(rng.integers(1, 600))
    lens = [int(rng.integers(0, 7)) for _ in range(int(rng.integers(1, 6)))]
    series_off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    packed = []
    for ln in lens:
        counts = np.zeros(nb, dtype=np.uint64)
        for _ in range(ln):
            counts = counts + rng.integers(0, 1 << int(rng.integers(1, 50)), nb).astype(np.uint64) * (rng.random(nb) < 0.6)
            if rng.random() < 0.25:
                counts[rng.integers(0, nb)] = 0
            vals = counts if rng.random() < 0.3 else np.diff(np.concatenate([[0], np.sort(counts)])).astype(np.uint64)
            b = bytearray(oracle.pack_u64(vals))
            how = rng.random()
            if how < 0.08 and len(b) > 1:
                b = b[:int(rng.integers(0, len(b)))]
            elif how < 0.16:
                b[int(rng.integers(0, len(b)))] ^= 1 << int(rng.integers(0, 8))
            packed.append(bytes(b))
    if not packed or sum(len(p) for p in packed) == 0:
        return []
    want_out, want_drop, want_fail = [], [], []
    h = 0
    for ln in lens:
        osink = oracle.DeltaDiffPackSink(nb)
        failed = False
        for _ in range(ln):
            if not failed:
                try:
                    o, dr, _ = osink.feed(packed[h])
                except oracle.OracleError:
                    failed = True
            want_fail.append(failed)
            want_out.append(b"" if failed else o)
            want_drop.append(False if failed else dr)
            h += 1
    off = np.concatenate([[0], np.cumsum([len(p) for p in packed])]).astype(np.uint64)
    blob = np.frombuffer(b"".join(packed), dtype=np.uint8).copy()
    out, oo, dropped, st = hist.delta_diff_batch(blob, off, series_off, nb)
    errs = []
    for i in range(len(packed)):
        if (st[i] != 0) != want_fail[i]:
            errs.append(("hist status", i, int(st[i]), want_fail[i], nb))
        elif not want_fail[i]:
            if out[int(oo[i]):int(oo[i + 1])].tobytes() != want_out[i]:
                errs.append(("hist bytes", i, nb))
            elif bool(dropped[i]) != want_drop[i]:
                errs.append(("hist dropped", i, nb))
    STATS["hist_failed"] = STATS.get("hist_failed", 0) + sum(want_fail)
    return errs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=60)
    ap.add_argument("--seed0", type=int, default=0)
    args = ap.parse_args()
    t0 = last = time.time()
    seed, bad, rounds = args.seed0, 0, 0
    while time.time() - t0 < args.seconds and bad <= 20:
        for kind in ("u64", "u64_delta", "f64_xor"):
            for e in streams_round(seed, kind):
                bad += 1
                print("MISMATCH seed", seed, kind, e, flush=True)
        for e in columns_round(seed):
            bad += 1
            print("MISMATCH seed", seed, "columns", e, flush=True)
        for e in hist_round(seed):
            bad += 1
            print("MISMATCH seed", seed, "histograms", e, flush=True)
        rounds += 5
        seed += 1
        if time.time() - last > 20:
            last = time.time()
            print("progress: rounds", rounds, "mismatches", bad, flush=True)
    print("stream/column fuzz done: rounds", rounds, "mismatches", bad, STATS, flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
