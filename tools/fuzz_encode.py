"""Randomised soak of the device encoder against the CPU oracle on adversarial values: extremes of the
integer range, ranges whose max - min needs every bit, values with chosen leading/trailing zero nibbles,
NaN payloads, signed zeros, infinities, denormals and raw random bit patterns.

For every seed and element type the device-encoded bytes and offsets must equal the oracle's, and the
device decode of them must equal the oracle's decode bit for bit.  Each seed also drives the appender
mirror and the oracle's appender through the same random append / append_nulls / reader / finish calls.
Usage: python tools/fuzz_encode.py [--seconds 60] [--seed0 0]
"""
import argparse
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

cv = importlib.import_module("compressed-vec_amd")
import oracle  # noqa: E402

ET = {"u32": 0x11, "u64": 0x10, "f32": 0x12}
NPD = {"u32": np.uint32, "u64": np.uint64, "f32": np.float32}


def int_piece(rng, k, bits):
    top = (1 << bits) - 1
    mode = int(rng.integers(0, 10))
    if mode == 0:
        p = np.zeros(k, dtype=np.uint64)
    elif mode == 1:
        p = np.full(k, top, dtype=np.uint64)
    elif mode == 2:                                   # max - min needs every bit
        p = rng.choice(np.array([0, 1, top - 1, top], dtype=np.uint64), size=k)
    elif mode == 3:                                   # narrow range just under the top
        p = np.uint64(top) - rng.integers(0, 1 << int(rng.integers(1, 12)), size=k, dtype=np.uint64)
    elif mode == 4:                                   # chosen trailing zero nibbles
        tz = 4 * int(rng.integers(0, bits // 4))
        p = (rng.integers(0, 1 << int(rng.integers(1, bits - tz + 1)), size=k, dtype=np.uint64, endpoint=False)
             << np.uint64(tz)) & np.uint64(top)
    elif mode == 5:                                   # one value in a constant stretch
        p = np.full(k, int(rng.integers(0, top, dtype=np.uint64)), dtype=np.uint64)
        p[int(rng.integers(0, k))] = int(rng.integers(0, top, dtype=np.uint64))
    elif mode == 6:                                   # a different width in every octet
        p = np.zeros(k, dtype=np.uint64)
        for a in range(0, k, 8):
            nb = int(rng.integers(1, bits + 1))
            p[a:a + 8] = rng.integers(0, 1 << nb, size=min(8, k - a), dtype=np.uint64, endpoint=False) \
                if nb < 64 else rng.integers(0, top, size=min(8, k - a), dtype=np.uint64, endpoint=True)
    elif mode == 7:                                   # mostly zero octets
        p = rng.integers(0, top, size=k, dtype=np.uint64, endpoint=True)
        p[rng.random(k) < 0.9] = 0
    elif mode == 8:                                   # base + small, base near a power of two
        base = (1 << int(rng.integers(1, bits))) - int(rng.integers(0, 3))
        p = (np.uint64(base) + rng.integers(0, 1 << int(rng.integers(1, 9)), size=k, dtype=np.uint64)) & np.uint64(top)
    else:
        p = rng.integers(0, top, size=k, dtype=np.uint64, endpoint=True)
    return p


def f32_piece(rng, k):
    mode = int(rng.integers(0, 9))
    if mode == 0:
        bits = rng.integers(0, 1 << 32, size=k, dtype=np.uint64).astype(np.uint32)     # raw patterns: NaNs, denormals
    elif mode == 1:
        bits = rng.choice(np.array([0, 0x80000000], dtype=np.uint32), size=k)           # signed zeros
    elif mode == 2:
        bits = rng.choice(np.array([0x7fc00000, 0x7fc00001, 0xffc00000, 0x7f800001, 0x3f800000], dtype=np.uint32), size=k)
    elif mode == 3:
        bits = rng.choice(np.array([0x7f800000, 0xff800000, 0x00000001, 0x807fffff, 0x7f7fffff], dtype=np.uint32), size=k)
    elif mode == 4:
        v = np.full(k, np.float32(rng.normal()), dtype=np.float32)
        if rng.random() < 0.5:
            v[int(rng.integers(0, k))] = np.nan
        bits = v.view(np.uint32)
    elif mode == 5:
        v = np.full(k, np.float32(rng.normal()), dtype=np.float32)
        v[int(rng.integers(0, k))] = -v[0]
        bits = v.view(np.uint32)
    elif mode == 6:
        bits = (rng.integers(0, 16, size=k) / np.float32(1.4)).astype(np.float32).view(np.uint32)
    elif mode == 7:
        bits = np.zeros(k, dtype=np.uint32)
    else:
        bits = rng.normal(size=k).astype(np.float32).view(np.uint32)
    return bits.astype(np.uint32)


def make(rng, tname):
    nv = int(rng.integers(1, 30))
    parts, off = [], [0]
    for _ in range(nv):
        n = int(rng.integers(0, int(rng.integers(2, 2500))))
        left, mine = n, []
        while left > 0:
            # pieces aligned to sections now and then, so whole sections are of one kind
            k = int(min(left, 256 if rng.random() < 0.4 else rng.integers(1, 600)))
            mine.append(f32_piece(rng, k) if tname == "f32" else
                        int_piece(rng, k, 32 if tname == "u32" else 64).astype(NPD[tname]))
            left -= k
        v = np.concatenate(mine) if mine else np.zeros(0, dtype=np.uint32 if tname == "f32" else NPD[tname])
        parts.append(v)
        off.append(off[-1] + v.size)
    flat = np.concatenate(parts) if parts else np.zeros(0, dtype=np.uint32)
    if tname == "f32":
        flat = flat.astype(np.uint32).view(np.float32)
    return np.ascontiguousarray(flat.astype(NPD[tname]) if tname != "f32" else flat), np.array(off, dtype=np.uint64)


def bits_of(a):
    a = np.ascontiguousarray(a)
    return a.view(np.uint32 if a.dtype.itemsize == 4 else np.uint64)


def one_round(seed, tname):
    rng = np.random.default_rng(seed)
    vals, eo = make(rng, tname)
    et = ET[tname]
    want_blob, want_off = oracle.batch_encode(et, vals, eo)
    got = cv.encode_batch(tname, vals, eo)
    errs = []
    if not np.array_equal(np.asarray(got.byte_off), np.asarray(want_off)):
        errs.append(("byte offsets",))
    elif not np.array_equal(np.asarray(got.blob), np.asarray(want_blob)):
        d = np.flatnonzero(np.asarray(got.blob) != np.asarray(want_blob))
        errs.append(("bytes", int(d[0]), int(d.size)))
    else:
        dev = got.to_device()
        dec, doff, st = dev.decode(return_status=True)
        wdec, wst, _ = oracle.batch_decode(et, want_blob, want_off, np.asarray(doff))
        if np.any(np.asarray(st) != 0) or np.any(np.asarray(wst) != 0):
            errs.append(("status",))
        elif not np.array_equal(bits_of(np.asarray(dec)), bits_of(wdec)):
            errs.append(("decoded values",))
        elif vals.size:
            # filter on a value that occurs, against the oracle
            p = vals[int(rng.integers(0, vals.size))]
            total, per = dev.filter_count(p, per_vector=True)
            wtotal, wper, _, _ = oracle.batch_filter_count(et, want_blob, want_off, p)
            if total != wtotal or not np.array_equal(per, wper):
                errs.append(("filter", p))
    return errs, int(vals.size)


APPENDER = {"u32": "VectorU32Appender", "u64": "VectorU64Appender", "f32": "VectorF32XorAppender"}


def appender_round(seed, tname):
    """A random sequence of append / append_many / append_nulls on the appender mirror and on the oracle's
    appender (src/vector.rs:207-379), a look through reader() on the way, then finish() with as many rows as
    appended or more: same bytes, null-section count included."""
    rng = np.random.default_rng(seed)
    a = getattr(cv, APPENDER[tname]).try_new(int(rng.integers(8, 4096)))
    o = oracle.Appender(ET[tname], 4096)
    errs = []
    for _ in range(int(rng.integers(0, 12))):
        op = int(rng.integers(0, 4))
        if op == 0:
            k = int(rng.integers(1, 700))
            v = f32_piece(rng, k).view(np.float32) if tname == "f32" else \
                int_piece(rng, k, 32 if tname == "u32" else 64).astype(NPD[tname])
            a.append_many(v)
            o.append_many(v)
        elif op == 1:
            v = NPD[tname](rng.integers(0, 1000))
            a.append(v)
            o.append(v)
        elif op == 2:
            n = int(rng.integers(0, 900))
            a.append_nulls(n)
            o.append_nulls(n)
        else:
            r = a.reader()
            w = cv.VectorReader.try_new(o.reader_bytes(), tname)
            if r.num_elements() != w.num_elements() or r.get_stats() != w.get_stats():
                errs.append(("reader", r.get_stats(), w.get_stats()))
            elif not np.array_equal(bits_of(r.iterate()), bits_of(w.iterate())):
                errs.append(("reader values",))
        if a.num_elements() != o.num_elements():
            errs.append(("num_elements", a.num_elements(), o.num_elements()))
    total = a.num_elements() + (int(rng.integers(0, 1500)) if rng.random() < 0.4 else 0)
    if a.finish(total) != o.finish(total):
        errs.append(("finish bytes", total))
    return errs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=60)
    ap.add_argument("--seed0", type=int, default=0)
    args = ap.parse_args()
    t0 = last = time.time()
    seed, bad, rounds, elems = args.seed0, 0, 0, 0
    while time.time() - t0 < args.seconds and bad <= 20:
        for tname in ("u32", "u64", "f32"):
            errs, n = one_round(seed, tname)
            rounds += 1
            elems += n
            for e in errs:
                bad += 1
                print("MISMATCH seed", seed, tname, e, flush=True)
            for e in appender_round(seed, tname):
                bad += 1
                print("MISMATCH seed", seed, tname, "appender", e, flush=True)
        seed += 1
        if time.time() - last > 20:
            last = time.time()
            print("progress: rounds", rounds, "elements", elems, "mismatches", bad, flush=True)
    print("encode fuzz done: rounds", rounds, "elements", elems, "mismatches", bad, flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
