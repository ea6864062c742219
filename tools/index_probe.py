#!/usr/bin/env python3
"""The index pass alone, back to back (its header lines may then still sit in the last-level cache from the
pass before), against the same pass inside whole steps:

    python3 tools/index_probe.py [cfg5sin]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import compressed_vec_amd as cv  # noqa: E402
from compressed_vec_amd import _native as N  # noqa: E402
from compressed_vec_amd.batch import _ptr, _stream_ptr  # noqa: E402
import profile_op  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "cfg5sin"
    assert cv.lib().cvec_init(0) == 0
    tname, vals, eo, pred, add_const = profile_op.build(name)
    batch = cv.encode_device(tname, vals, eo)
    ws = batch.workspace()
    res = batch.result_buffer()
    p = cv.make_pred(batch.etype, pred)

    def index():
        N.check(N.lib().cvec_index_dev(C.byref(batch.view), _ptr(ws), ws.numel(), _stream_ptr()))

    def timed(fn, n=200):
        for _ in range(50):
            fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        a.record()
        for _ in range(n):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / n

    t_idx = timed(index)
    t_step = timed(lambda: batch.filter_count_async(p, result=res))
    junk = torch.empty(1 << 28, dtype=torch.float32, device="cuda")
    t_fill = timed(lambda: junk.fill_(1.0))
    t_both = timed(lambda: (junk.fill_(1.0), index()))
    print("%s: index pass alone %.4f ms; whole step %.4f ms; index pass behind a 1 GiB fill %.4f ms (fill alone %.4f)" % (
        name, t_idx, t_step, t_both - t_fill, t_fill))


if __name__ == "__main__":
    main()
