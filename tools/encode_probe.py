#!/usr/bin/env python3
"""Time of encode_device (all kernels of one call, device-resident values) on the bench configurations:

    python3 tools/encode_probe.py [cfg ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import compressed_vec_amd as cv  # noqa: E402
import profile_op  # noqa: E402


def main():
    names = sys.argv[1:] or ["cfg1", "cfg2", "cfg3", "cfg4", "cfg5sin", "cfg5mix"]
    assert cv.lib().cvec_init(0) == 0
    for name in names:
        tname, vals, eo, _, _ = profile_op.build(name)
        ref = cv.encode_device(tname, vals, eo)
        for _ in range(3):
            b = cv.encode_device(tname, vals, eo)
        same = b.n_bytes == ref.n_bytes and bool(torch.equal(b.d_bytes[:b.n_bytes], ref.d_bytes[:ref.n_bytes]))
        torch.cuda.synchronize()
        ts = []
        for _ in range(10):
            t0 = time.perf_counter()
            b = cv.encode_device(tname, vals, eo)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        raw = vals.numel() * vals.element_size()
        best = min(ts)
        # the same through preallocated buffers, ten calls enqueued back to back
        nvec = eo.numel() - 1
        bufs = cv.EncodeBuffers(tname, nvec, vals.numel())
        for _ in range(3):
            cv.encode_device_async(bufs, vals, eo)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            cv.encode_device_async(bufs, vals, eo)
        torch.cuda.synchronize()
        t_async = (time.perf_counter() - t0) / 10
        same = same and bool(torch.equal(bufs.d_bytes[:ref.n_bytes], ref.d_bytes[:ref.n_bytes]))
        print("%-8s encode enqueued back to back %.3f ms per call = %.0f GB/s of values" % (name, t_async * 1e3, raw / t_async / 1e9), flush=True)
        del bufs
        print("%-8s encode %.3f ms (best of 10 calls, allocations of the call included) = %.0f GB/s of values; %d bytes; "
              "repeatable: %s" % (name, best * 1e3, raw / best / 1e9, b.n_bytes, same), flush=True)
        del b, ref, vals
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
