#!/usr/bin/env python3
"""Unpack time of the legacy stream codecs over octets of different sizes: 13107 streams x 10000 u64 values whose
deltas need `bits` bits (octets of 1 + bits-dependent bytes), device-resident, best of 5 calls.

    python3 tools/stream_probe.py [bits ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import compressed_vec_amd as cv  # noqa: E402
from compressed_vec_amd import stream  # noqa: E402


def main():
    bits_list = [int(x) for x in sys.argv[1:]] or [0, 2, 4, 10, 20, 40]
    assert cv.lib().cvec_init(0) == 0
    n, length = 13107, 10000
    rng = np.random.default_rng(5)
    seo = torch.arange(n + 1, device="cuda", dtype=torch.int64) * length
    for bits in bits_list:
        inc = rng.integers(0, 1 << bits, size=(n, length), dtype=np.int64) if bits else np.zeros((n, length), np.int64)
        if bits:
            inc[rng.random((n, length)) < 0.0] = 0
        vals = torch.from_numpy(np.cumsum(inc, axis=1).reshape(-1)).cuda()
        d_bytes, d_bo, _, d_res = stream.pack_device("u64_delta", vals, seo, n * length)
        nb = int(d_res.cpu()[0])
        out = torch.zeros(n * length, dtype=torch.int64, device="cuda")
        ws = torch.empty(cv.lib().cvec_stream_unpack_workspace_bytes(n, n * length), dtype=torch.uint8, device="cuda")
        best = 1e9
        for _ in range(6):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            stream.unpack_device("u64_delta", d_bytes, d_bo, nb, seo, n * length, d_out=out, ws=ws)
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        ok = bool(torch.equal(out, vals))
        print("deltas of %2d bits: %.1f bytes per octet, unpack %.3f ms = %.0f GB/s of values, exact: %s"
              % (bits, nb / (n * length / 8), best * 1e3, n * length * 8 / best / 1e9, ok), flush=True)
        del vals, d_bytes, out, ws


if __name__ == "__main__":
    main()
