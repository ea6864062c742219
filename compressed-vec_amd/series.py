"""Many appenders side by side, resident in HBM: the host mirror of cvec_series_appender_* (include/cvec.h).

One `SeriesAppender` stands for n VectorAppender<T, W> of the reference (src/vector.rs:207-379) that receive their
calls together, as the series of a time-series store do on every scrape.  Open sections and encoded sections stay
on the device; `finish` / `reader` return a DeviceBatch that the filter and decode calls take as it is."""
import ctypes as C

import numpy as np
import torch

from . import _native as N
from .batch import DeviceBatch, TORCH_DTYPE, _ptr, _stream_ptr, etype_of


class SeriesAppender:
    def __init__(self, etype, n_series, device="cuda"):
        self.etype = etype_of(etype)
        self.n_series = int(n_series)
        self.device = torch.device(device)
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            N.check(N.lib().cvec_series_appender_new(self.etype, self.n_series, C.byref(h)))
        self._h = h

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            N.lib().cvec_series_appender_free(h)

    def _values(self, values):
        want = TORCH_DTYPE[self.etype]
        if not torch.is_tensor(values):
            arr = np.ascontiguousarray(values)
            if arr.dtype.kind == "u":                     # u32 / u64 bits travel as int32 / int64
                arr = arr.view(np.dtype("i%d" % arr.dtype.itemsize))
            values = torch.from_numpy(arr)
        if values.dtype != want:
            raise N.CodingError(N.ERR_BAD_ARGUMENT, "values must have the appender's element type")
        return values.to(self.device).contiguous()

    def append(self, values, elem_off=None):
        """append (src/vector.rs:300-307) for every series: `values` is n_series x k (every series receives k
        values), or 1-D together with elem_off (int64[n_series + 1]: series v receives values[elem_off[v]:elem_off[v+1]])."""
        t = self._values(values)
        with torch.cuda.device(self.device):
            if elem_off is None:
                if t.dim() != 2 or t.shape[0] != self.n_series:
                    raise N.CodingError(N.ERR_BAD_ARGUMENT, "values must be n_series x k")
                if t.shape[1]:
                    N.check(N.lib().cvec_series_appender_append_dev(self._h, _ptr(t), None, int(t.shape[1]), _stream_ptr()))
            else:
                eo = self._offsets(elem_off, t.numel())
                keep = t if t.numel() else torch.zeros(4, dtype=t.dtype, device=self.device)
                N.check(N.lib().cvec_series_appender_append_dev(self._h, _ptr(keep), _ptr(eo), 0, _stream_ptr()))

    def append_host(self, values, elem_off=None):
        """append for values in host memory (numpy): n_series x k, or 1-D with elem_off (uint64[n_series + 1], from 0).
        The library copies them to the device itself."""
        arr = np.ascontiguousarray(values)
        if arr.dtype.itemsize != (8 if self.etype == N.U64 else 4) or (arr.dtype.kind == "f") != (self.etype == N.F32):
            raise N.CodingError(N.ERR_BAD_ARGUMENT, "values must have the appender's element type")
        with torch.cuda.device(self.device):
            if elem_off is None:
                if arr.ndim != 2 or arr.shape[0] != self.n_series:
                    raise N.CodingError(N.ERR_BAD_ARGUMENT, "values must be n_series x k")
                N.check(N.lib().cvec_series_appender_append_host(self._h, arr.ctypes.data_as(C.c_void_p), None, int(arr.shape[1]),
                                                                 _stream_ptr()))
            else:
                eo = np.ascontiguousarray(elem_off, dtype=np.uint64)
                if eo.size != self.n_series + 1 or int(eo[-1]) > arr.size:
                    raise N.CodingError(N.ERR_BAD_ARGUMENT, "elem_off must hold n_series + 1 offsets within the values")
                N.check(N.lib().cvec_series_appender_append_host(self._h, arr.ctypes.data_as(C.c_void_p),
                                                                 eo.ctypes.data_as(C.c_void_p), 0, _stream_ptr()))

    def _offsets(self, elem_off, limit=None):
        eo = torch.as_tensor(elem_off, dtype=torch.int64)
        if eo.numel() != self.n_series + 1 or bool((eo[1:] < eo[:-1]).any()) or int(eo[0]) < 0:
            raise N.CodingError(N.ERR_BAD_ARGUMENT, "elem_off must hold n_series + 1 non-decreasing offsets")
        if limit is not None and int(eo[-1]) > limit:
            raise N.CodingError(N.ERR_BAD_ARGUMENT, "elem_off reaches past the values")
        return eo.to(self.device).contiguous()

    def append_nulls(self, counts):
        """append_nulls (src/vector.rs:311-335): an int (every series) or one count per series."""
        with torch.cuda.device(self.device):
            if isinstance(counts, int):
                if counts:
                    N.check(N.lib().cvec_series_appender_append_nulls_dev(self._h, None, counts, _stream_ptr()))
                return
            c = torch.as_tensor(counts, dtype=torch.int64)
            eo = torch.zeros(self.n_series + 1, dtype=torch.int64)
            eo[1:] = torch.cumsum(c.cpu(), 0)
            N.check(N.lib().cvec_series_appender_append_nulls_dev(self._h, _ptr(self._offsets(eo)), 0, _stream_ptr()))

    def _sealed_batch(self, reader, total_num_rows):
        L = N.lib()
        nb, ns = C.c_uint64(0), C.c_uint64(0)
        detail = (C.c_uint64 * 2)()
        with torch.cuda.device(self.device):
            st = L.cvec_series_appender_seal(self._h, 1 if reader else 0, int(total_num_rows), C.byref(nb), C.byref(ns), detail,
                                             _stream_ptr())
            if st == N.INVALID_NUM_ROWS:
                raise N.CodingError(st, int(detail[0]), int(detail[1]))
            N.check(st)
            n = self.n_series
            # the read kernels stage the blob with 16-byte loads: pad the allocation so the last chunk is readable
            d_bytes = torch.zeros(((nb.value + 16 + 255) // 256) * 256, dtype=torch.uint8, device=self.device)
            d_off = torch.empty(n + 1, dtype=torch.int64, device=self.device)
            d_sec = torch.empty(n + 1, dtype=torch.int64, device=self.device)
            st = L.cvec_series_appender_emit_dev(self._h, _ptr(d_bytes), d_bytes.numel(), _ptr(d_off), _ptr(d_sec), _stream_ptr())
            if st != N.OK:
                L.cvec_series_appender_unseal(self._h)
            N.check(st)
        return DeviceBatch(self.etype, d_bytes, d_off, d_sec, n, ns.value, nb.value)

    def finish(self, total_num_rows):
        """finish (src/vector.rs:344-370) for every series; the appender is empty afterwards."""
        return self._sealed_batch(False, total_num_rows)

    def reader(self):
        """reader (src/vector.rs:372-378): the complete sections of every series, as a batch."""
        return self._sealed_batch(True, 0)

    def reset(self):
        N.check(N.lib().cvec_series_appender_reset(self._h))

    def stats(self):
        a, b, c = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        N.check(N.lib().cvec_series_appender_stats(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return {"segments": a.value, "encoded_bytes": b.value, "state_bytes": c.value}
