#!/usr/bin/env python3
"""Soak of the series appender against oracle appenders: random numbers of series, random sequences of ragged and
uniform appends, ragged and uniform null runs (short, section-sized, huge), readers and finishes with and without
padding, the handle reused; every reader / finish compared byte for byte per series.

    python3 tools/fuzz_series.py --seconds 60 [--seed 1]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import compressed_vec_amd as cv  # noqa: E402
import oracle  # noqa: E402

NPD = {"u32": np.uint32, "u64": np.uint64, "f32": np.float32}
OET = {"u32": oracle.U32, "u64": oracle.U64, "f32": oracle.F32}


def values(rng, tname, n):
    mode = int(rng.integers(0, 6))
    if tname == "f32":
        if mode == 0:
            return np.zeros(n, np.float32)
        if mode == 1:
            return np.full(n, -7.25, np.float32)
        if mode == 2:
            x = rng.standard_normal(n).astype(np.float32)
            x[rng.random(n) < 0.05] = np.nan
            return x
        return (rng.standard_normal(n) * 10.0 ** int(rng.integers(-4, 8))).astype(np.float32)
    dt = NPD[tname]
    if mode == 0:
        return np.zeros(n, dt)
    if mode == 1:
        return np.full(n, 123456, dt)
    if mode == 2:
        return (rng.integers(0, 14, size=n) + (1 << 24)).astype(dt)
    if mode == 3:
        x = np.zeros(n, dt)
        k = rng.random(n) < 0.03
        x[k] = rng.integers(1, 1000, size=int(k.sum()))
        return x
    return rng.integers(0, 2 ** (32 if tname == "u32" else 64), size=n, dtype=np.uint64).astype(dt)


def same(host, oracles, total=None):
    for v, o in enumerate(oracles):
        want = o.reader_bytes() if total is None else o.finish(total)
        if host.vector(v) != bytes(want):
            return v
    return -1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=30)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    assert cv.lib().cvec_init(0) == 0
    oracle.build()
    rng = np.random.default_rng(args.seed)
    t_end = time.time() + args.seconds
    rounds = checks = series = 0
    while time.time() < t_end:
        tname = ["u32", "u64", "f32"][int(rng.integers(0, 3))]
        n = int(rng.integers(1, 60))
        app = cv.SeriesAppender(tname, n)
        for _use in range(int(rng.integers(1, 3))):
            oracles = [oracle.Appender(OET[tname], 64) for _ in range(n)]
            for _step in range(int(rng.integers(1, 12))):
                op = int(rng.integers(0, 7))
                if op <= 1:                                   # ragged append
                    counts = rng.integers(0, [3, 300, 900][int(rng.integers(0, 3))], size=n)
                    counts[rng.random(n) < 0.2] = 0
                    off = np.zeros(n + 1, np.int64)
                    off[1:] = np.cumsum(counts)
                    vals = np.concatenate([values(rng, tname, int(c)) for c in counts] + [np.zeros(0, NPD[tname])])
                    app.append(vals, off)
                    for v, o in enumerate(oracles):
                        if counts[v]:
                            o.append_many(vals[off[v]:off[v + 1]])
                elif op == 2:                                 # a scrape: k values for every series
                    k = int(rng.integers(1, 600)) if rng.random() < 0.5 else int(rng.integers(1, 9))
                    block = np.stack([values(rng, tname, k) for _ in range(n)])
                    app.append(block)
                    for v, o in enumerate(oracles):
                        o.append_many(block[v])
                elif op == 3:                                 # ragged nulls, some of them long
                    counts = rng.integers(0, [5, 256, 1000, 3_000_000][int(rng.integers(0, 4))], size=n)
                    counts[rng.random(n) < 0.3] = 0
                    app.append_nulls(counts)
                    for v, o in enumerate(oracles):
                        if counts[v]:
                            o.append_nulls(int(counts[v]))
                elif op == 4:                                 # uniform nulls
                    k = int([1, 3, 255, 256, 257, 512, 100_000][int(rng.integers(0, 7))])
                    app.append_nulls(k)
                    for o in oracles:
                        o.append_nulls(k)
                else:                                         # reader in between
                    bad = same(app.reader().download(), oracles)
                    checks += 1
                    if bad >= 0:
                        print("MISMATCH reader: seed %d round %d type %s series %d" % (args.seed, rounds, tname, bad), flush=True)
                        sys.exit(1)
            longest = max(o.num_elements() for o in oracles)
            total = longest + int([0, 1, 255, 256, 5000, 10_000_000][int(rng.integers(0, 6))])
            bad = same(app.finish(total).download(), oracles, total)
            checks += 1
            series += n
            if bad >= 0:
                print("MISMATCH finish: seed %d round %d type %s series %d" % (args.seed, rounds, tname, bad), flush=True)
                sys.exit(1)
        rounds += 1
        if rounds % 20 == 0:
            print("rounds %d, readers and finishes compared %d, series finished %d" % (rounds, checks, series), flush=True)
    print("ok: %d rounds, %d readers / finishes compared (%d series finished), no difference" % (rounds, checks, series), flush=True)


if __name__ == "__main__":
    main()
