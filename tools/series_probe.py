#!/usr/bin/env python3
"""Scrape-style ingest at the headline shape: n series receive k values per call until each holds `length`; then
finish().  Prints the time of the append calls (those that complete sections and those that do not), of finish(),
the device memory held before finish, and whether the finished batch equals the one-shot encode of the same matrix.

    python3 tools/series_probe.py [n_series] [length] [k]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import compressed_vec_amd as cv  # noqa: E402
from compressed_vec_amd import synth  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
    length = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
    k = int(sys.argv[3]) if len(sys.argv) > 3 else 64
    assert cv.lib().cvec_init(0) == 0
    chunks = [synth.batch_values_torch("f32_xor", min(4096, n - c0), length, v_first=c0)[0] for c0 in range(0, n, 4096)]
    mat = torch.cat(chunks).reshape(n, length)
    eo = torch.arange(n + 1, device="cuda", dtype=torch.int64) * length
    one_shot = cv.encode_device("f32", mat.reshape(-1), eo)
    app = cv.SeriesAppender("f32", n)
    t_plain, t_flush = [], []
    for rep in range(2):                                  # the first pass warms the handle's buffers
        t_plain, t_flush = [], []
        for c0 in range(0, length, k):
            block = mat[:, c0:c0 + k].contiguous()
            torch.cuda.synchronize()
            before = app.stats()["segments"]
            t0 = time.perf_counter()
            app.append(block)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            (t_flush if app.stats()["segments"] > before else t_plain).append(dt)
        held = app.stats()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        batch = app.finish(length)
        torch.cuda.synchronize()
        t_finish = time.perf_counter() - t0
    same = batch.n_bytes == one_shot.n_bytes and bool(torch.equal(batch.d_bytes[:batch.n_bytes], one_shot.d_bytes[:one_shot.n_bytes])) \
        and bool(torch.equal(batch.d_byte_off, one_shot.d_byte_off)) and bool(torch.equal(batch.d_sec_base, one_shot.d_sec_base))
    raw = n * length * 4
    ms = lambda xs: 1e3 * sum(xs) / max(len(xs), 1)
    total = sum(t_plain) + sum(t_flush) + t_finish
    print("series %d x %d f32, %d values per call: %d calls without a complete section %.3f ms each, %d calls completing one "
          "%.3f ms each (%.0f GB/s of the 256 values encoded), finish %.2f ms; ingest + finish %.1f ms = %.1f GB/s of raw values; "
          "held before finish: %d segments, %.1f MiB encoded + %.1f MiB open sections and offsets (raw %.1f MiB); "
          "equal to the one-shot encode: %s"
          % (n, length, k, len(t_plain), ms(t_plain), len(t_flush), ms(t_flush), n * 256 * 4 / (ms(t_flush) * 1e6) if t_flush else 0,
             t_finish * 1e3, total * 1e3, raw / total / 1e9, held["segments"], held["encoded_bytes"] / 2**20,
             held["state_bytes"] / 2**20, raw / 2**20, same), flush=True)
    sys.exit(0 if same else 1)


if __name__ == "__main__":
    main()
