"""Batches of encoded vectors, on the host and resident in MI355X HBM.

A batch is n reference-format vectors stored back to back in one byte blob plus byte_off[n+1].
Device memory, streams and (for multi-GPU) torch.distributed come from PyTorch; all work on the
bytes is done by the HIP library through the C ABI (include/cvec.h)."""
import ctypes as C

import numpy as np
import torch

from . import _native as N

NP_DTYPE = {N.U64: np.uint64, N.U32: np.uint32, N.F32: np.float32}
TORCH_DTYPE = {N.U64: torch.int64, N.U32: torch.int32, N.F32: torch.float32}
WIDTH = {N.U64: 8, N.U32: 4, N.F32: 4}
FIXED_LEN = 256


def etype_of(dtype):
    """Maps 'u32' / 'u64' / 'f32' (or numpy dtypes, or the header's minor-type byte) to the element type."""
    if isinstance(dtype, int) and dtype in NP_DTYPE:
        return dtype
    if isinstance(dtype, str):
        key = dtype.lower()
        if key in ("u32", "uint32"):
            return N.U32
        if key in ("u64", "uint64"):
            return N.U64
        if key in ("f32", "float32"):
            return N.F32
    d = np.dtype(dtype)
    if d == np.uint32:
        return N.U32
    if d == np.uint64:
        return N.U64
    if d == np.float32:
        return N.F32
    raise ValueError("unsupported element type %r" % (dtype,))


def _bits(etype, value):
    """Raw little-endian bits of a scalar of the element type, as an int."""
    a = np.array([value], dtype=NP_DTYPE[etype])
    return int(a.view(np.uint64 if etype == N.U64 else np.uint32)[0])


def make_pred(etype, pred):
    """scalar → EqualsSink predicate, sequence → OneOfSink predicate (reference src/filter.rs:129-177)."""
    p = N.Pred()
    if isinstance(pred, (list, tuple, np.ndarray)):
        vals = list(pred)
        if not 1 <= len(vals) <= N.MAX_PRED_VALUES:
            raise N.CodingError(N.ERR_BAD_ARGUMENT, "OneOf takes 1..%d values" % N.MAX_PRED_VALUES)
        p.kind, p.count = N.PRED_ONEOF, len(vals)
    else:
        vals = [pred]
        p.kind, p.count = N.PRED_EQ, 1
    for i, v in enumerate(vals):
        p.values[i] = _bits(etype, v)
    return p


def make_chain(etype, add_const=None):
    c = N.SinkChain()
    if add_const is not None:
        c.add_const = 1
        c.add_const_bits = _bits(etype, add_const)
    return c


class EncodedBatch:
    """Host-side batch: `blob` (uint8 array) and `byte_off` (uint64, n+1)."""

    def __init__(self, etype, blob, byte_off):
        self.etype = etype_of(etype)
        self.blob = np.ascontiguousarray(blob, dtype=np.uint8)
        self.byte_off = np.ascontiguousarray(byte_off, dtype=np.uint64)
        assert self.byte_off.ndim == 1 and self.byte_off.size >= 1

    @classmethod
    def from_vectors(cls, etype, vectors):
        """Concatenates individually encoded vectors (bytes objects)."""
        lens = np.array([len(v) for v in vectors], dtype=np.uint64)
        off = np.zeros(len(vectors) + 1, dtype=np.uint64)
        np.cumsum(lens, out=off[1:])
        blob = np.frombuffer(b"".join(bytes(v) for v in vectors), dtype=np.uint8).copy() if len(vectors) else \
            np.zeros(0, dtype=np.uint8)
        return cls(etype, blob, off)

    @property
    def n_vectors(self):
        return self.byte_off.size - 1

    def vector(self, i):
        return self.blob[int(self.byte_off[i]):int(self.byte_off[i + 1])].tobytes()

    def to_device(self, device="cuda", stream=None):
        return DeviceBatch.upload(self, device=device)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class DeviceBatch:
    """A batch resident in HBM: blob, byte_off and sec_base as torch tensors + the C view of them."""

    def __init__(self, etype, d_bytes, d_byte_off, d_sec_base, n_vectors, n_sections, n_bytes):
        self.etype = etype
        self.d_bytes, self.d_byte_off, self.d_sec_base = d_bytes, d_byte_off, d_sec_base
        self.n_vectors, self.n_sections, self.n_bytes = int(n_vectors), int(n_sections), int(n_bytes)
        self.device = d_bytes.device
        v = N.DBatchView()
        v.etype = etype
        v.n_vectors, v.n_sections, v.n_bytes = self.n_vectors, self.n_sections, self.n_bytes
        v.d_bytes, v.d_byte_off, v.d_sec_base = d_bytes.data_ptr(), d_byte_off.data_ptr(), d_sec_base.data_ptr()
        self.view = v
        self._ws = None
        self._result = None
        self._layout_sig = None
        self._keep_index = False    # build_index() was called: ungated filters reuse the section index
        self._index_valid = False   # the workspace holds the ungated index of this batch

    @classmethod
    def upload(cls, host, device="cuda", pinned=False):
        """H2D copy of a host batch, then sec_base from the headers on the device."""
        dev = torch.device(device)
        n_bytes = int(host.byte_off[-1])
        # the kernels stage the blob with 16-byte loads: pad the allocation so the last chunk is readable
        d_bytes = torch.zeros(((n_bytes + 16 + 255) // 256) * 256, dtype=torch.uint8, device=dev)
        hb = torch.from_numpy(host.blob[:n_bytes])
        ho = torch.from_numpy(host.byte_off.view(np.int64))
        if pinned:
            hb, ho = hb.pin_memory(), ho.pin_memory()
        d_bytes[:n_bytes].copy_(hb, non_blocking=pinned)
        d_byte_off = ho.to(dev, non_blocking=pinned)
        return cls.from_device_tensors(host.etype, d_bytes, d_byte_off, n_bytes)

    @classmethod
    def from_device_tensors(cls, etype, d_bytes, d_byte_off, n_bytes):
        n = d_byte_off.numel() - 1
        d_sec_base = torch.empty(n + 1, dtype=torch.int64, device=d_bytes.device)
        nsec = C.c_uint64(0)
        with torch.cuda.device(d_bytes.device):
            N.check(N.lib().cvec_sec_base_dev(_ptr(d_bytes), _ptr(d_byte_off), n, _ptr(d_sec_base), C.byref(nsec),
                                              _stream_ptr()))
        return cls(etype, d_bytes, d_byte_off, d_sec_base, n, nsec.value, n_bytes)

    @classmethod
    def concat(cls, batches):
        """One resident batch out of several (same element type, same device): the vectors of the inputs
        in order.  Vectors are independent, so the blob of the result is the inputs' blobs back to back;
        only the offsets are rebased (device-side copies, no host round trip)."""
        batches = list(batches)
        first = batches[0]
        if any(b.etype != first.etype or b.device != first.device for b in batches):
            raise N.CodingError(N.ERR_BAD_ARGUMENT, "batches differ in element type or device")
        n_bytes = sum(b.n_bytes for b in batches)
        d_bytes = torch.zeros(((n_bytes + 16 + 255) // 256) * 256, dtype=torch.uint8, device=first.device)
        offs, secs = [], []
        at_bytes = at_secs = 0
        for b in batches:
            d_bytes[at_bytes:at_bytes + b.n_bytes].copy_(b.d_bytes[:b.n_bytes])
            offs.append(b.d_byte_off[:-1] + at_bytes)
            secs.append(b.d_sec_base[:-1] + at_secs)
            at_bytes += b.n_bytes
            at_secs += b.n_sections
        end = torch.tensor([at_bytes], dtype=torch.int64, device=first.device)
        send = torch.tensor([at_secs], dtype=torch.int64, device=first.device)
        return cls(first.etype, d_bytes, torch.cat(offs + [end]), torch.cat(secs + [send]),
                   sum(b.n_vectors for b in batches), at_secs, at_bytes)

    def append_values(self, d_values, d_elem_off):
        """The batch extended by freshly encoded vectors (Vector*Appender::encode_all on the device for the
        new values): returns the new DeviceBatch; this one stays valid."""
        return DeviceBatch.concat([self, encode_device(self.etype, d_values, d_elem_off)])

    def download(self):
        blob = self.d_bytes[:self.n_bytes].cpu().numpy()
        off = self.d_byte_off.cpu().numpy().view(np.uint64)
        return EncodedBatch(self.etype, blob, off)

    # -- scratch owned by the batch so that repeated queries allocate nothing ------------------------
    def workspace(self, min_bytes=0):
        nbytes = max(N.lib().cvec_workspace_bytes(self.n_vectors, self.n_sections), min_bytes)
        if self._ws is None or self._ws.numel() < nbytes:
            self._ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
            self._index_valid = False
        return self._ws

    def layout_signature(self):
        """A number that two batches share iff their vectors have the same section counts (sec_base equal):
        computed on the device once per batch, so that comparing layouts costs no synchronisation later."""
        if self._layout_sig is None:
            sb = self.d_sec_base
            idx = torch.arange(1, sb.numel() + 1, device=sb.device, dtype=torch.int64)
            # position-weighted wrapping sums with two odd multipliers: equal arrays give equal sums, and two
            # different monotone arrays of the same length and total do not collide in practice
            a = int((sb * (idx * 0x9E3779B1 + 1)).sum().item())
            b = int(((sb ^ (sb >> 7)) * (idx * 0x85EBCA77 + 3)).sum().item())
            self._layout_sig = (self.n_vectors, self.n_sections, a, b)
        return self._layout_sig

    def result_buffer(self):
        if self._result is None:
            self._result = torch.zeros(N.RESULT_WORDS, dtype=torch.int64, device=self.device)
        return self._result

    def valid_sections(self):
        """After a filter or decode pass: (leading sections of every vector that produced a mask / data,
        flags of the vectors whose iteration ended at a header that does not parse), as numpy arrays."""
        ws = self.workspace()
        valid = torch.empty(max(self.n_vectors, 1), dtype=torch.int32, device=self.device)
        pf = torch.zeros(max(self.n_vectors, 1), dtype=torch.uint8, device=self.device)
        with torch.cuda.device(self.device):
            N.check(N.lib().cvec_valid_sections_dev(C.byref(self.view), _ptr(ws), ws.numel(), _ptr(valid), _ptr(pf),
                                                    _stream_ptr()))
        return valid.cpu().numpy()[:self.n_vectors].view(np.uint32), pf.cpu().numpy()[:self.n_vectors].astype(bool)

    def error_details(self, status):
        """After a filter or decode pass that returned per-vector statuses: an (n_vectors, 2) uint64 array,
        [section index at which the vector's iteration stopped (2**64-1: its header), payload of the reference's
        CodingError (BadOffset offset, InvalidSectionType byte, WrongVectorType byte; otherwise the byte offset of
        that section in the vector)]; zeros for vectors that did not fail."""
        ws = self.workspace()
        n = max(self.n_vectors, 1)
        st = torch.as_tensor(np.ascontiguousarray(status, dtype=np.int32), device=self.device) \
            if not isinstance(status, torch.Tensor) else status
        valid = torch.empty(n, dtype=torch.int32, device=self.device)
        pf = torch.zeros(n, dtype=torch.uint8, device=self.device)
        detail = torch.zeros(2 * n, dtype=torch.int64, device=self.device)
        with torch.cuda.device(self.device):
            N.check(N.lib().cvec_valid_sections_dev(C.byref(self.view), _ptr(ws), ws.numel(), _ptr(valid), _ptr(pf),
                                                    _stream_ptr()))
            N.check(N.lib().cvec_error_detail_dev(C.byref(self.view), _ptr(st), _ptr(valid), _ptr(pf), _ptr(detail),
                                                  _stream_ptr()))
        return detail.cpu().numpy().view(np.uint64).reshape(-1, 2)[:self.n_vectors]

    # -- operations -------------------------------------------------------------------------------------
    def build_index(self):
        """Walks the section headers of every vector once (what each read operation otherwise does as
        its first kernel) and keeps the result in the batch's workspace: from here on ungated filter
        and decode passes skip that walk.  The batch is immutable, so the index never goes stale (gated
        passes over an index read the gate in the section kernel and leave the index as it is).  All
        operations of a batch share its workspace: enqueue them on one stream (or synchronise between
        streams)."""
        ws = self.workspace()
        with torch.cuda.device(self.device):
            N.check(N.lib().cvec_index_dev(C.byref(self.view), _ptr(ws), ws.numel(), _stream_ptr()))
        self._keep_index = self._index_valid = True
        return self

    def filter_count_async(self, pred, per_vector=None, status=None, masks=None, result=None, gate=None,
                           tail_rule=None, read_path=None):
        """Enqueues decode+filter+count on the current stream; returns the device result tensor
        (int64[4]: total, failed vectors, max status, reserved).  `gate`: masks of an earlier pass over
        a batch with the same section layout; sections whose gate mask is empty are not decoded.
        `tail_rule` ("format" | "reference") and `read_path` ("auto" | "slots" | "vstream") hold for this call
        alone (cvec_read_options); None = the process-wide default."""
        res = result if result is not None else self.result_buffer()
        if isinstance(pred, (list, tuple, np.ndarray)) and len(pred) > N.MAX_PRED_VALUES:
            # OneOf over a large set (the reference's set is a Vec, src/filter.rs:143-160): one pass per 16 values
            if gate is not None:
                raise N.CodingError(N.ERR_BAD_ARGUMENT, "a gated pass takes at most %d values" % N.MAX_PRED_VALUES)
            vals = (C.c_uint64 * len(pred))(*[_bits(self.etype, v) for v in pred])
            ws = self.workspace(N.lib().cvec_filter_set_workspace_bytes(self.n_vectors, self.n_sections))
            with torch.cuda.device(self.device):
                N.check(N.lib().cvec_filter_count_set_dev(C.byref(self.view), vals, len(pred), _ptr(res), _ptr(per_vector),
                                                          _ptr(status), _ptr(masks), _ptr(ws), ws.numel(), _stream_ptr()))
            self._index_valid = False
            return res
        p = pred if isinstance(pred, N.Pred) else make_pred(self.etype, pred)
        ws = self.workspace()
        if self._keep_index and not self._index_valid:
            self.build_index()                  # an unindexed pass used the workspace since: walk the headers again
        if tail_rule is not None or read_path is not None:
            indexed = bool(self._keep_index and self._index_valid)
            o = N.read_options(read_path, tail_rule, indexed, gate.data_ptr() if gate is not None else None)
            with torch.cuda.device(self.device):
                N.check(N.lib().cvec_filter_count_opt_dev(C.byref(self.view), C.byref(p), C.byref(o), _ptr(res),
                                                          _ptr(per_vector), _ptr(status), _ptr(masks), _ptr(ws),
                                                          ws.numel(), _stream_ptr()))
            self._index_valid = indexed
            return res
        with torch.cuda.device(self.device):
            if self._keep_index and self._index_valid:
                N.check(N.lib().cvec_filter_count_indexed_dev(C.byref(self.view), C.byref(p), _ptr(gate), _ptr(res),
                                                              _ptr(per_vector), _ptr(status), _ptr(masks), _ptr(ws),
                                                              ws.numel(), _stream_ptr()))
            else:
                N.check(N.lib().cvec_filter_count_gated_dev(C.byref(self.view), C.byref(p), _ptr(gate), _ptr(res),
                                                            _ptr(per_vector), _ptr(status), _ptr(masks), _ptr(ws),
                                                            ws.numel(), _stream_ptr()))
                self._index_valid = False       # only cvec_index_dev and the indexed calls keep an index
        return res

    def filter_count(self, pred, per_vector=False, return_status=False, return_masks=False, tail_rule=None,
                     read_path=None):
        pv = torch.empty(self.n_vectors, dtype=torch.int64, device=self.device) if per_vector else None
        st = torch.empty(self.n_vectors, dtype=torch.int32, device=self.device) if return_status else None
        mk = torch.zeros(self.n_sections * 32, dtype=torch.uint8, device=self.device) if return_masks else None
        res = self.filter_count_async(pred, pv, st, mk, tail_rule=tail_rule, read_path=read_path).cpu().numpy()
        if res[1] and not return_status:
            raise N.CodingError(int(res[2]), "%d vector(s) failed" % int(res[1]))
        out = [int(res[0])]
        if per_vector:
            out.append(pv.cpu().numpy().view(np.uint64))
        if return_status:
            out.append(st.cpu().numpy())
        if return_masks:
            out.append(mk.cpu().numpy())
        return out[0] if len(out) == 1 else tuple(out)

    def filter_masks(self, pred, gate=None):
        """Runs the filter pass and keeps its per-section masks on the device: returns a FilteredColumn
        (masks, number of sections every vector yielded) for multi_filter_count / match_positions.
        `gate`: a FilteredColumn over a batch with the same section layout (the previous column of a
        MultiVectorFilter); sections it left without a match are skipped (src/filter.rs:277-282)."""
        masks = torch.zeros(self.n_sections * 32, dtype=torch.uint8, device=self.device)
        g = None
        if gate is not None and gate.batch.n_sections == self.n_sections and \
                gate.batch.n_vectors == self.n_vectors and gate.batch.layout_signature() == self.layout_signature():
            g = gate.masks
        res = self.filter_count_async(pred, masks=masks, gate=g)
        valid = torch.empty(max(self.n_vectors, 1), dtype=torch.int32, device=self.device)
        ws = self.workspace()
        with torch.cuda.device(self.device):
            N.check(N.lib().cvec_valid_sections_dev(C.byref(self.view), _ptr(ws), ws.numel(), _ptr(valid), None,
                                                    _stream_ptr()))
        return FilteredColumn(self, masks, valid, res.clone())

    def decode_async(self, out, add_const=None, status=None, result=None, tail_rule=None, read_path=None):
        """Enqueues decode through [AddConstSink →] VecSink into `out` (256 * n_sections elements).  `tail_rule`,
        `read_path`: as for filter_count_async."""
        assert out.numel() >= self.n_sections * FIXED_LEN and out.element_size() == WIDTH[self.etype]
        chain = make_chain(self.etype, add_const)
        ws = self.workspace()
        res = result if result is not None else self.result_buffer()
        if tail_rule is not None or read_path is not None:
            indexed = bool(self._keep_index and self._index_valid)
            o = N.read_options(read_path, tail_rule, indexed)
            with torch.cuda.device(self.device):
                N.check(N.lib().cvec_decode_opt_dev(C.byref(self.view), C.byref(chain), C.byref(o), _ptr(out), _ptr(res),
                                                    _ptr(status), _ptr(ws), ws.numel(), _stream_ptr()))
            self._index_valid = indexed
            return res
        with torch.cuda.device(self.device):
            indexed = self._keep_index and self._index_valid
            fn = N.lib().cvec_decode_indexed_dev if indexed else N.lib().cvec_decode_dev
            N.check(fn(C.byref(self.view), C.byref(chain), _ptr(out), _ptr(res), _ptr(status), _ptr(ws), ws.numel(),
                       _stream_ptr()))
        self._index_valid = indexed
        return res

    def vector_items(self, v, start=0, stop=None):
        """VectorItemIter over vector v of the resident batch (src/vector.rs:512-556): its values at positions
        [start, stop) as a device tensor.  Only that vector is decoded (a one-vector view of the batch)."""
        if not 0 <= v < self.n_vectors:
            raise IndexError(v)
        bo = self.d_byte_off[v:v + 2]
        sb = self.d_sec_base[v:v + 2]
        a, b = (int(x) for x in bo.cpu())
        s0, s1 = (int(x) for x in sb.cpu())
        # the blob pointer must stay 16-byte aligned: keep the batch's base and its absolute offsets
        one = DeviceBatch(self.etype, self.d_bytes, bo.contiguous(), (sb - s0).contiguous(), 1, s1 - s0, self.n_bytes)
        out = torch.zeros(max(s1 - s0, 1) * FIXED_LEN, dtype=TORCH_DTYPE[self.etype], device=self.device)
        st = torch.zeros(1, dtype=torch.int32, device=self.device)
        one.decode_async(out, status=st)
        n = int.from_bytes(bytes(self.d_bytes[a + 8:a + 12].cpu().numpy()), "little") if b - a >= 16 else 0
        if int(st.item()) != N.OK:
            raise N.CodingError(int(st.item()))
        stop = n if stop is None else min(stop, n)
        return out[start:stop]

    def decode(self, add_const=None, return_status=False, tail_rule=None, read_path=None):
        """Returns (values, out_off): every vector padded to whole sections, vectors back to back."""
        out = torch.zeros(self.n_sections * FIXED_LEN, dtype=TORCH_DTYPE[self.etype], device=self.device)
        st = torch.empty(self.n_vectors, dtype=torch.int32, device=self.device) if return_status else None
        res = self.decode_async(out, add_const, st, tail_rule=tail_rule, read_path=read_path).cpu().numpy()
        if res[1] and not return_status:
            raise N.CodingError(int(res[2]), "%d vector(s) failed" % int(res[1]))
        vals = out.cpu().numpy().view(NP_DTYPE[self.etype])
        off = self.d_sec_base.cpu().numpy().view(np.uint64) * np.uint64(FIXED_LEN)
        return (vals, off, st.cpu().numpy()) if return_status else (vals, off)


class FilteredColumn:
    """The masks of one filter pass over a DeviceBatch (what the reference's VectorFilter yields, for
    every vector of the batch), resident on the device."""

    def __init__(self, batch, masks, valid, result):
        self.batch, self.masks, self.valid, self.result = batch, masks, valid, result
        c = N.MaskColumn()
        c.d_masks, c.d_sec_base, c.d_valid = masks.data_ptr(), batch.d_sec_base.data_ptr(), valid.data_ptr()
        c.n_sections = batch.n_sections
        self.column = c

    def match_positions_async(self, pos=None, pos_off=None):
        """filter::match_positions for every vector, left on the device and not waited for: (positions int32 tensor
        with room for every element, pos_off int64[n+1] tensor; vector v's positions are pos[pos_off[v]:pos_off[v+1]]).
        `pos` / `pos_off`: buffers of an earlier call to write into."""
        b = self.batch
        lib = N.lib()
        need = lib.cvec_positions_workspace_bytes(b.n_sections)
        if getattr(self, "_pos_ws", None) is None or self._pos_ws.numel() < need:
            self._pos_ws = torch.empty(need, dtype=torch.uint8, device=b.device)
        cap = b.n_sections * FIXED_LEN
        if pos_off is None:
            pos_off = torch.empty(b.n_vectors + 1, dtype=torch.int64, device=b.device)
        if pos is None:
            pos = torch.empty(max(cap, 1), dtype=torch.int32, device=b.device)
        assert pos.numel() >= max(cap, 1) and pos_off.numel() >= b.n_vectors + 1
        with torch.cuda.device(b.device):
            N.check(lib.cvec_match_positions_dev(C.byref(self.column), b.n_vectors, _ptr(pos_off), _ptr(pos), cap,
                                                 _ptr(self._pos_ws), self._pos_ws.numel(), _stream_ptr()))
        return pos, pos_off

    def match_positions(self):
        """filter::match_positions for every vector: (positions uint32, pos_off uint64[n+1]) on the host."""
        pos, pos_off = self.match_positions_async()
        off = pos_off.cpu().numpy().view(np.uint64)
        return pos[:int(off[-1])].cpu().numpy().view(np.uint32), off


def multi_filter(batches, preds, per_row=False, return_masks=False):
    """MultiVectorFilter over whole batches: column c is batches[c] filtered by preds[c]; every column
    after the first is gated by the masks of the one before it, so sections that an earlier column
    already ruled out are not decoded (put the most selective column first, src/filter.rs:236-238)."""
    cols = []
    for b, p in zip(batches, preds):
        cols.append(b.filter_masks(p, gate=cols[-1] if cols else None))
    return multi_filter_count(cols, per_row=per_row, return_masks=return_masks)


def multi_filter_count(columns, per_row=False, return_masks=False):
    """MultiVectorFilter + count_hits (reference src/filter.rs:239-299) over FilteredColumns with the same
    number of vectors: row r ANDs the masks of vector r of every column.  Returns the total
    (and per-row counts / (ANDed masks, masks yielded per row) on request)."""
    k = len(columns)
    if not 1 <= k <= N.MAX_COLUMNS:
        raise N.CodingError(N.ERR_BAD_ARGUMENT, "1..%d columns" % N.MAX_COLUMNS)
    b0 = columns[0].batch
    n = b0.n_vectors
    if any(c.batch.n_vectors != n for c in columns):
        raise N.CodingError(N.ERR_BAD_ARGUMENT, "columns differ in the number of vectors")
    arr = (N.MaskColumn * k)(*[c.column for c in columns])
    lib = N.lib()
    ws = torch.empty(lib.cvec_multi_workspace_bytes(b0.n_sections), dtype=torch.uint8, device=b0.device)
    res = torch.zeros(N.RESULT_WORDS, dtype=torch.int64, device=b0.device)
    pr = torch.empty(n, dtype=torch.int64, device=b0.device) if per_row else None
    rs = torch.empty(max(n, 1), dtype=torch.int32, device=b0.device) if return_masks else None
    mk = torch.zeros(b0.n_sections * 32, dtype=torch.uint8, device=b0.device) if return_masks else None
    with torch.cuda.device(b0.device):
        N.check(lib.cvec_multi_filter_count_dev(arr, k, n, _ptr(res), _ptr(pr), _ptr(rs), _ptr(mk), _ptr(ws),
                                                ws.numel(), _stream_ptr()))
    out = [int(res[0].item())]
    if per_row:
        out.append(pr.cpu().numpy().view(np.uint64))
    if return_masks:
        out.append(mk.cpu().numpy())
        out.append(rs.cpu().numpy()[:n])
    return out[0] if len(out) == 1 else tuple(out)


def encode_device(etype, d_values, d_elem_off, n_elements_total=None):
    """Vector*Appender::encode_all for a batch already in HBM.  d_values: 1-D tensor of the element
    type (int32/int64 carry u32/u64 bits), d_elem_off: int64[n+1].  Returns a DeviceBatch."""
    etype = etype_of(etype)
    n = d_elem_off.numel() - 1
    n_el = int(d_values.numel()) if n_elements_total is None else int(n_elements_total)
    L = N.lib()
    cap = L.cvec_max_encoded_bytes(etype, n, n_el)
    wsb = L.cvec_encode_workspace_bytes(n, n_el)
    dev = d_values.device
    d_bytes = torch.empty(((cap + 255) // 256) * 256, dtype=torch.uint8, device=dev)
    d_byte_off = torch.empty(n + 1, dtype=torch.int64, device=dev)
    d_sec_base = torch.empty(n + 1, dtype=torch.int64, device=dev)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    res = torch.zeros(N.RESULT_WORDS, dtype=torch.int64, device=dev)
    with torch.cuda.device(dev):
        N.check(L.cvec_encode_dev(etype, _ptr(d_values), _ptr(d_elem_off), n, n_el, _ptr(d_bytes), d_bytes.numel(),
                                  _ptr(d_byte_off), _ptr(d_sec_base), _ptr(res), _ptr(ws), ws.numel(), _stream_ptr()))
    r = res.cpu().numpy()
    return DeviceBatch(etype, d_bytes, d_byte_off, d_sec_base, n, int(r[3]), int(r[0]))


class EncodeBuffers:
    """Preallocated outputs and workspace of encode_device_async for batches of up to n vectors / n_el elements:
    a caller that encodes batch after batch (an ingest loop) allocates once and never synchronises."""

    def __init__(self, etype, n, n_el, device="cuda"):
        self.etype = etype_of(etype)
        L = N.lib()
        cap = L.cvec_max_encoded_bytes(self.etype, n, n_el)
        self.n, self.n_el = int(n), int(n_el)
        self.d_bytes = torch.empty(((cap + 255) // 256) * 256, dtype=torch.uint8, device=device)
        self.d_byte_off = torch.empty(n + 1, dtype=torch.int64, device=device)
        self.d_sec_base = torch.empty(n + 1, dtype=torch.int64, device=device)
        self.ws = torch.empty(L.cvec_encode_workspace_bytes(n, n_el), dtype=torch.uint8, device=device)
        self.result = torch.zeros(N.RESULT_WORDS, dtype=torch.int64, device=device)


def encode_device_async(bufs, d_values, d_elem_off, n_elements_total=None):
    """Vector*Appender::encode_all for a batch in HBM, enqueued on the current stream without any host
    synchronisation or allocation: blob, byte_off, sec_base go to `bufs` (EncodeBuffers), bufs.result = [bytes, 0, 0,
    sections] once the stream has run.  Returns bufs."""
    n = d_elem_off.numel() - 1
    n_el = int(d_values.numel()) if n_elements_total is None else int(n_elements_total)
    if n > bufs.n or n_el > bufs.n_el:
        raise N.CodingError(N.ERR_CAPACITY, "the buffers hold %d vectors / %d elements" % (bufs.n, bufs.n_el))
    with torch.cuda.device(d_values.device):
        N.check(N.lib().cvec_encode_dev(bufs.etype, _ptr(d_values), _ptr(d_elem_off), n, n_el, _ptr(bufs.d_bytes),
                                        bufs.d_bytes.numel(), _ptr(bufs.d_byte_off), _ptr(bufs.d_sec_base),
                                        _ptr(bufs.result), _ptr(bufs.ws), bufs.ws.numel(), _stream_ptr()))
    return bufs


def encode_batch(etype, values, elem_off):
    """Host arrays in, host EncodedBatch out (one H2D, encode on the GPU, one D2H)."""
    etype = etype_of(etype)
    vals = np.ascontiguousarray(values, dtype=NP_DTYPE[etype])
    eo = np.ascontiguousarray(elem_off, dtype=np.uint64)
    if np.any(eo[1:] < eo[:-1]):
        raise N.CodingError(N.ERR_BAD_ARGUMENT, "elem_off must be non-decreasing")
    if eo.size > 1 and int((eo[1:] - eo[:-1]).max()) > 0xffffffff:
        raise N.CodingError(N.INVALID_NUM_ROWS, int((eo[1:] - eo[:-1]).max()), 0xffffffff)
    n_el = int(eo[-1] - eo[0])
    signed = vals.view(np.int64 if etype == N.U64 else (np.int32 if etype == N.U32 else np.float32))
    d_values = torch.from_numpy(signed[int(eo[0]):int(eo[-1])]).cuda() if n_el else \
        torch.zeros(4, dtype=TORCH_DTYPE[etype], device="cuda")
    d_eo = torch.from_numpy((eo - eo[0]).view(np.int64)).cuda()
    return encode_device(etype, d_values, d_eo, n_el).download()
