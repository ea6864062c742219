#!/usr/bin/env python3
"""Decode kernel times per configuration (HIP events of the library around its three kernels) with a check of the
decoded values, for comparing builds:

    CVEC_LIB_PATH=compressed-vec_amd/lib/<variant>.so [PROBE_READ_PATH=slots|vstream] python3 tools/decode_probe.py [cfg ...]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import compressed_vec_amd as cv  # noqa: E402
from compressed_vec_amd import _native as N  # noqa: E402
import profile_op  # noqa: E402


RP = os.environ.get("PROBE_READ_PATH") or None


def main():
    names = sys.argv[1:] or ["cfg1", "cfg3", "cfg4", "cfg5sin", "cfg5mix"]
    lib = cv.lib()
    assert lib.cvec_init(0) == 0
    for name in names:
        tname, vals, eo, pred, add_const = profile_op.build(name)
        nvec, length = eo.numel() - 1, int(eo[1].item())
        batch = cv.encode_device(tname, vals, eo)
        out = torch.empty(batch.n_sections * 256, dtype=vals.dtype, device="cuda")
        for _ in range(30):
            batch.decode_async(out, add_const=add_const, read_path=RP)
        torch.cuda.synchronize()
        iters = 20
        N.check(lib.cvec_timing_begin(iters))
        for _ in range(iters):
            batch.decode_async(out, add_const=add_const, read_path=RP)
        torch.cuda.synchronize()
        ms = [C.c_float(0), C.c_float(0), C.c_float(0)]
        nrec = C.c_int32(0)
        N.check(lib.cvec_timing_end(C.byref(ms[0]), C.byref(ms[1]), C.byref(ms[2]), C.byref(nrec)))
        ok = torch.equal(out.view(nvec, -1)[:, :length].contiguous().view(-1).view(torch.uint8), (vals + add_const).view(torch.uint8))
        k = ms[1].value / nrec.value
        traffic = vals.numel() * vals.element_size() + batch.n_bytes
        print("%-8s sections kernel %.4f ms  (%.0f GB/s of traffic, %.3f of 8 TB/s)  index %.4f  exact %s" %
              (name, k, traffic / k / 1e6, traffic / k / 1e6 / 8000, ms[0].value / nrec.value, ok), flush=True)
        del batch, out, vals
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
