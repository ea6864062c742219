#!/usr/bin/env python3
"""When every wave of the section kernel starts and ends (probe build only:
CVEC_LIB_PATH=compressed-vec_amd/lib/libcvec_hip_probes.so; the kernel stamps the 100 MHz wall clock):

    python3 tools/wave_times.py [cfg5sin]

Prints the distribution of the waves' end times and their means by XCD, by wave of the workgroup and by
block-index octile (the second half of the blocks are the workgroups that reach a CU second)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import compressed_vec_amd as cv  # noqa: E402
from compressed_vec_amd import _native as N  # noqa: E402
import profile_op  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "cfg5sin"
    assert cv.lib().cvec_init(0) == 0
    tname, vals, eo, pred, add_const = profile_op.build(name)
    batch = cv.encode_device(tname, vals, eo)
    res = batch.result_buffer()
    p = cv.make_pred(batch.etype, pred)
    assert cv.lib().cvec_set_read_path(N.PATH_SLOTS) == 0
    for _ in range(200):
        batch.filter_count_async(p, result=res)
    torch.cuda.synchronize()
    buf = np.zeros(8192 * 2, dtype=np.uint64)
    fn = cv.lib().cvec_debug_sections_times
    fn.restype = C.c_int32
    fn.argtypes = [C.c_void_p]
    assert fn(buf.ctypes.data) == 0
    t = buf.reshape(8192, 2).astype(np.int64)
    nw = int((t[:, 0] != 0).sum())
    t = t[:nw]
    gw = np.arange(nw)
    e = (t[:, 1] - t[:, 0].min()) / 100.0
    blk = gw // 4
    nb = blk.max() + 1
    print("end by XCD (block % 8):", " ".join("%.0f" % e[blk % 8 == x].mean() for x in range(8)))
    print("end by wave in workgroup:", " ".join("%.0f" % e[gw % 4 == x].mean() for x in range(4)))
    print("end by block index octile:", " ".join("%.0f" % e[(blk * 8 // nb) == x].mean() for x in range(8)))
    wgend = e[:nw // 4 * 4].reshape(-1, 4)
    print("spread inside a workgroup (max-min) mean %.1f; std of workgroup means %.1f" % ((wgend.max(1) - wgend.min(1)).mean(), wgend.mean(1).std()))
    s0 = (t[:, 0] - t[:, 0].min()) / 100.0
    print("k_sections %s: %d waves; start max %.2f; end min %.2f  p10 %.2f  median %.2f  p90 %.2f  max %.2f us" % (
        name, nw, s0.max(), e.min(), np.percentile(e, 10), np.median(e), np.percentile(e, 90), e.max()))


if __name__ == "__main__":
    main()
