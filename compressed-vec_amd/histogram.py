"""Histogram helpers of the reference (src/histogram.rs) for batches: compress_geom_nonincreasing and
the DeltaDiffPackSink re-packing ("2D delta") of series of histograms."""
import ctypes as C
import struct

import numpy as np
import torch

from . import _native as N
from . import stream
from .batch import _ptr, _stream_ptr

FORMAT_EMPTY, FORMAT_GEOMETRIC_DELTA, FORMAT_GEOMETRIC1_DELTA = 0, 1, 2   # BinHistogramFormat (src/histogram.rs:8-12)


def compress_geom_nonincreasing(num_buckets, initial_bucket, multiplier, format_code, bucket_values):
    """src/histogram.rs:55-76: BinHistogramHeader + PackedGeometricBuckets + pack_u64 of the values
    (packed on the GPU)."""
    header = struct.pack("<BHHdd", format_code, 16 + 2, num_buckets, initial_bucket, multiplier)
    return header + stream.pack_u64(bucket_values)


def delta_diff_device(d_bytes, d_byte_off, n_bytes, d_series_off, num_buckets):
    """DeltaDiffPackSink over device-resident histograms.  d_bytes: uint8 (16-byte aligned, padded
    allocation), d_byte_off int64[n_hist+1], d_series_off int64[n_series+1].  Returns (d_out, d_out_off,
    d_dropped, d_status, d_result)."""
    n_hist, n_series, device = d_byte_off.numel() - 1, d_series_off.numel() - 1, d_byte_off.device
    lib = N.lib()
    cap = lib.cvec_hist_max_packed_bytes(n_hist, num_buckets)
    d_out = torch.zeros((cap + 255) // 256 * 256, dtype=torch.uint8, device=device)
    d_oo = torch.zeros(n_hist + 1, dtype=torch.int64, device=device)
    d_dropped = torch.zeros(max(n_hist, 1), dtype=torch.int32, device=device)
    d_st = torch.zeros(max(n_hist, 1), dtype=torch.int32, device=device)
    d_res = torch.zeros(N.RESULT_WORDS, dtype=torch.int64, device=device)
    ws = torch.empty(lib.cvec_hist_workspace_bytes(n_hist, num_buckets), dtype=torch.uint8, device=device)
    with torch.cuda.device(device):
        N.check(lib.cvec_hist_delta_diff_dev(_ptr(d_bytes), _ptr(d_byte_off), int(n_bytes), _ptr(d_series_off), n_series,
                                             n_hist, num_buckets, _ptr(d_out), d_out.numel(), _ptr(d_oo), _ptr(d_dropped),
                                             _ptr(d_st), _ptr(d_res), _ptr(ws), ws.numel(), _stream_ptr()))
    return d_out, d_oo, d_dropped, d_st, d_res


def delta_diff_batch(blob, byte_off, series_off, num_buckets, device="cuda"):
    """Host arrays in and out: (out blob uint8, out_off uint64[n_hist+1], dropped bool[n_hist], status
    int32[n_hist]).  Histogram h's re-packed stream is blob[out_off[h]:out_off[h+1]]."""
    bo = np.ascontiguousarray(byte_off, dtype=np.uint64)
    so = np.ascontiguousarray(series_off, dtype=np.uint64)
    n_hist, nbytes = bo.size - 1, int(bo[-1])
    if so[0] != 0 or so[-1] != n_hist or np.any(so[1:] < so[:-1]):
        raise N.CodingError(N.ERR_BAD_ARGUMENT, "series_off must run from 0 to the number of histograms")
    d_bytes = torch.zeros((nbytes + 16 + 255) // 256 * 256, dtype=torch.uint8, device=device)
    d_bytes[:nbytes].copy_(torch.from_numpy(np.ascontiguousarray(blob, dtype=np.uint8)[:nbytes].copy()))
    d_bo = torch.from_numpy(bo.view(np.int64).copy()).to(device)
    d_so = torch.from_numpy(so.view(np.int64).copy()).to(device)
    d_out, d_oo, d_dropped, d_st, d_res = delta_diff_device(d_bytes, d_bo, nbytes, d_so, num_buckets)
    total = int(d_res[0].item())
    return (d_out[:total].cpu().numpy(), d_oo.cpu().numpy().view(np.uint64), d_dropped.cpu().numpy()[:n_hist] != 0,
            d_st.cpu().numpy()[:n_hist])


class DeltaDiffPackSink:
    """The reference's sink for one series, fed one packed histogram at a time (unpack + finish):
    keeps the previous histogram and re-packs the pair on the GPU.  Bulk work should hand whole
    series to delta_diff_batch instead."""

    def __init__(self, num_buckets):
        self.num_buckets = num_buckets
        self._prev = None
        self.value_dropped = False

    def reset(self):
        self._prev = None
        self.value_dropped = False

    def feed(self, packed):
        cur = bytes(packed)
        hists = [cur] if self._prev is None else [self._prev, cur]
        off = np.zeros(len(hists) + 1, dtype=np.uint64)
        off[1:] = np.cumsum([len(h) for h in hists])
        blob = np.frombuffer(b"".join(hists), dtype=np.uint8)
        out, oo, dropped, st = delta_diff_batch(blob, off, [0, len(hists)], self.num_buckets)
        if st[-1]:
            raise N.CodingError(int(st[-1]))
        self._prev = cur
        self.value_dropped = bool(dropped[-1])
        return out[int(oo[-2]):int(oo[-1])].tobytes()
