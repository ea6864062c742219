#!/usr/bin/env python3
"""Section kernel time of the count-only pass against the pass with per-section masks, warm, per configuration
(HIP events of the library around its kernels):

    python3 tools/mask_probe.py [cfg ...]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import compressed_vec_amd as cv  # noqa: E402
from compressed_vec_amd import _native as N  # noqa: E402
import profile_op  # noqa: E402


def kernel_ms(lib, fn, iters=30):
    for _ in range(60):
        fn()
    torch.cuda.synchronize()
    N.check(lib.cvec_timing_begin(iters))
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    ms = [C.c_float(0) for _ in range(3)]
    n = C.c_int32(0)
    N.check(lib.cvec_timing_end(C.byref(ms[0]), C.byref(ms[1]), C.byref(ms[2]), C.byref(n)))
    return ms[1].value / n.value


def main():
    names = sys.argv[1:] or ["cfg1", "cfg3", "cfg4", "cfg5sin", "cfg5mix"]
    lib = cv.lib()
    assert lib.cvec_init(0) == 0
    for name in names:
        tname, vals, eo, pred, _ = profile_op.build(name)
        occ = vals[:1 << 22]
        occ = occ[occ != 0][7].item()
        batch = cv.encode_device(tname, vals, eo)
        masks = torch.empty(batch.n_sections * 32, dtype=torch.uint8, device="cuda")
        res = batch.result_buffer()
        for what, p in (("reference predicate", pred), ("occurring value", occ)):
            pp = cv.make_pred(batch.etype, p)
            t_c = kernel_ms(lib, lambda: batch.filter_count_async(pp, result=res))
            t_m = kernel_ms(lib, lambda: batch.filter_count_async(pp, result=res, masks=masks))
            t_r = kernel_ms(lib, lambda: batch.filter_count_async(pp, result=res, tail_rule="reference"))
            print("%-8s %-20s count %.4f ms   masks %.4f ms (+%.1f %%)   count under the reference tail rule %.4f ms (+%.1f %%)" %
                  (name, what, t_c, t_m, 100 * (t_m / t_c - 1), t_r, 100 * (t_r / t_c - 1)), flush=True)
        del batch, masks, vals
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
