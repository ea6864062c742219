#!/usr/bin/env python3
"""The operations that the per-configuration profiles do not reach, each a fixed number of times, for
rocprofv3 --kernel-trace --stats:  OneOf predicates (4 and 16 values) on dense u32, a two-column MultiVectorFilter
whose second column is gated by the first (fresh index and resident index), match positions, the legacy stream
codecs (pack / unpack of u64 delta and f64 XOR) and a histogram delta-diff batch.

    python3 tools/profile_misc.py [iters]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import compressed_vec_amd as cv  # noqa: E402
from compressed_vec_amd import stream  # noqa: E402
import profile_op  # noqa: E402


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    assert cv.lib().cvec_init(0) == 0
    # OneOf on dense u32 (values 1..15)
    tname, vals, eo, _, _ = profile_op.build("cfg1")
    dense = cv.encode_device(tname, vals, eo)
    res = dense.result_buffer()
    for pred in ([3, 5, 7, 11], list(range(1, 17))):
        p = cv.make_pred(dense.etype, pred)
        for _ in range(iters):
            dense.filter_count_async(p, result=res)
    # two columns, the second behind the gate of the first; then match positions of the ANDed masks
    half = vals.numel() // 2
    n2 = (eo.numel() - 1) // 2
    col_a = cv.encode_device(tname, vals[:n2 * 10000], eo[:n2 + 1])
    col_b = cv.encode_device(tname, vals[n2 * 10000:2 * n2 * 10000], eo[:n2 + 1])
    for _ in range(iters):
        cv.multi_filter([col_a, col_b], [3, 7])
    col_b.build_index()
    for _ in range(iters):
        cv.multi_filter([col_a, col_b], [3, 7])
    fc = col_a.filter_masks(3)
    for _ in range(iters):
        fc.match_positions()
    del dense, col_a, col_b, vals, fc
    torch.cuda.empty_cache()
    # legacy stream codecs: 13107 streams x 10000 values
    n, length = 13107, 10000
    rng = np.random.default_rng(5)
    inc = torch.from_numpy(np.cumsum(rng.integers(0, 1000, size=(n, length)), axis=1).astype(np.int64).reshape(-1)).cuda()
    flt = torch.from_numpy((rng.standard_normal(n * length) * 100).astype(np.float64).view(np.int64)).cuda()
    seo = torch.arange(n + 1, device="cuda", dtype=torch.int64) * length
    for kind, d_vals in (("u64_delta", inc), ("f64_xor", flt)):
        for _ in range(iters):
            d_bytes, d_bo, _, d_res = stream.pack_device(kind, d_vals, seo, n * length)
        nb = int(d_res.cpu()[0])
        for _ in range(iters):
            stream.unpack_device(kind, d_bytes, d_bo, nb, seo, n * length)
    torch.cuda.synchronize()
    print("done", flush=True)


if __name__ == "__main__":
    main()
