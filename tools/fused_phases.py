#!/usr/bin/env python3
"""Phase times of k_fused per wave (probe build: CVEC_LIB_PATH=compressed-vec_amd/lib/libcvec_hip_probes.so):
start -> vectors found -> first chunk's descriptors -> chunk loop done -> per-vector fold done.

    CVEC_LIB_PATH=... python3 tools/fused_phases.py [cfg5sin]"""
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
    assert cv.lib().cvec_set_read_path(N.PATH_FUSED) == 0
    for _ in range(200):
        batch.filter_count_async(p, result=res)
    torch.cuda.synchronize()
    buf = np.zeros(2048 * 8, dtype=np.uint64)
    fn = cv.lib().cvec_debug_fused_times
    fn.restype = C.c_int32
    fn.argtypes = [C.c_void_p]
    assert fn(buf.ctypes.data) == 0
    t = buf.reshape(2048, 8)[:, :5].astype(np.int64)
    t0 = t[:, 0].min()
    us = (t - t0) / 100.0
    names = ["start", "found vectors", "first descriptors", "loop done", "fold done"]
    for i, nme in enumerate(names):
        print("%-18s min %7.2f  median %7.2f  max %7.2f us (since the first wave's start)" % (nme, us[:, i].min(), np.median(us[:, i]), us[:, i].max()))
    d = np.diff(us, axis=1)
    for i in range(4):
        print("phase %-28s median %7.2f  max %7.2f us" % (names[i] + " -> " + names[i + 1], np.median(d[:, i]), d[:, i].max()))


def slots(name):
    """start / end of every wave of k_sections on the same batch"""
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
    print("end by XCD (block % 8):", " ".join("%.0f" % e[blk % 8 == x].mean() for x in range(8)))
    print("end by wave in workgroup:", " ".join("%.0f" % e[gw % 4 == x].mean() for x in range(4)))
    nb = blk.max() + 1
    print("end by block index octile:", " ".join("%.0f" % e[(blk * 8 // nb) == x].mean() for x in range(8)))
    print("end by (block // 8) % 4:", " ".join("%.0f" % e[(blk // 8) % 4 == x].mean() for x in range(4)))
    print("first 64 waves' end:", " ".join("%.0f" % x for x in e[:64]))
    wgend = e.reshape(-1, 4) if nw % 4 == 0 else e[:nw // 4 * 4].reshape(-1, 4)
    print("spread inside a workgroup (max-min) mean %.1f; std of workgroup means %.1f" % ((wgend.max(1) - wgend.min(1)).mean(), wgend.mean(1).std()))
    t0 = t[:, 0].min()
    us = (t - t0) / 100.0
    print("k_sections: %d waves; start max %.2f; end min %.2f  p10 %.2f  median %.2f  p90 %.2f  max %.2f us" % (
        len(us), us[:, 0].max(), us[:, 1].min(), np.percentile(us[:, 1], 10), np.median(us[:, 1]), np.percentile(us[:, 1], 90), us[:, 1].max()))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[2] == "slots":
        assert cv.lib().cvec_init(0) == 0
        slots(sys.argv[1])
        sys.exit(0)
    main()
