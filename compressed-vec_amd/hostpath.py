"""Host-buffer operations: numpy arrays in, numpy arrays out, through the *_batch entry points of the
C ABI.  The library cuts the batch into chunks of whole vectors and rotates them through a few
device slots, so the copy-in of one chunk, the kernels of another and the copy-out of a third run at
the same time (include/cvec.h).  Page-locked arrays (pin()) get the full overlap."""
import ctypes as C

import numpy as np

from . import _native as N
from .batch import FIXED_LEN, NP_DTYPE, EncodedBatch, etype_of, make_chain, make_pred


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def pin(array):
    """A page-locked copy of `array` (backed by a pinned torch tensor that the result keeps alive)."""
    import torch
    a = np.ascontiguousarray(array)
    t = torch.empty(a.nbytes, dtype=torch.uint8).pin_memory()
    out = t.numpy().view(a.dtype)[:a.size].reshape(a.shape)
    out[...] = a
    return out


def encode_host(etype, values, elem_off, out=None):
    """Vector*Appender::encode_all for every vector of a host batch.  Returns an EncodedBatch; `out`
    may bring a (pinned) uint8 array of at least cvec_max_encoded_bytes to encode into."""
    etype = etype_of(etype)
    vals = np.ascontiguousarray(values, dtype=NP_DTYPE[etype])
    eo = np.ascontiguousarray(elem_off, dtype=np.uint64)
    n = eo.size - 1
    L = N.lib()
    cap = L.cvec_max_encoded_bytes(etype, n, int(eo[-1] - eo[0]))
    blob = out if out is not None else np.empty(cap, dtype=np.uint8)
    if blob.size < cap:
        raise N.CodingError(N.ERR_CAPACITY)
    bo = np.zeros(n + 1, dtype=np.uint64)
    total = C.c_uint64(0)
    N.check(L.cvec_encode_batch(etype, _p(vals), _p(eo), n, _p(blob), blob.size, _p(bo), C.byref(total)))
    return EncodedBatch(etype, blob[:total.value], bo)


def filter_count_host(batch, pred, per_vector=False, return_status=False, tail_rule=None, read_path=None):
    """reader.filter_iter(pred) + count_hits for every vector of a host EncodedBatch.  `tail_rule` ("format" |
    "reference") and `read_path` hold for this call alone (None = the process-wide default)."""
    n = batch.n_vectors
    L = N.lib()
    p = pred if isinstance(pred, N.Pred) else make_pred(batch.etype, pred)
    total = C.c_uint64(0)
    per = np.zeros(n, dtype=np.uint64) if per_vector else None
    st = np.zeros(n, dtype=np.int32) if return_status else None
    o = N.read_options(read_path, tail_rule)
    rc = L.cvec_filter_count_batch_opt(batch.etype, _p(batch.blob), _p(batch.byte_off), n, C.byref(p), C.byref(o),
                                       C.byref(total), _p(per) if per is not None else None,
                                       _p(st) if st is not None else None)
    # 1..8 and ERR_SECTION_COUNT are "the largest per-vector status" (the statuses say which vectors);
    # anything above is a failure of the call as a whole
    if rc != N.OK and not (return_status and rc <= N.ERR_SECTION_COUNT and st.any()):
        N.check(rc)
    out = [total.value]
    if per_vector:
        out.append(per)
    if return_status:
        out.append(st)
    return out[0] if len(out) == 1 else tuple(out)


def decode_host(batch, add_const=None, return_status=False, out=None, tail_rule=None, read_path=None):
    """VectorReader::decode_to_sink through [AddConstSink ->] VecSink for every vector of a host
    EncodedBatch: (values padded to whole sections, out_off[n+1]) [+ status]."""
    n = batch.n_vectors
    L = N.lib()
    # sections per vector from the headers (u32 num_elements at +8)
    starts = batch.byte_off[:-1].astype(np.int64)
    has_header = (batch.byte_off[1:] - batch.byte_off[:-1]) >= 16
    at = starts[has_header]
    ne = np.zeros(at.size, dtype=np.int64)
    for k in range(4):
        ne |= batch.blob[at + 8 + k].astype(np.int64) << (8 * k)
    # a claim beyond what the bytes could hold (a section takes at least one byte) is cut to one slot
    # more than that bound, as the library does
    bound = (batch.byte_off[1:] - batch.byte_off[:-1]).astype(np.int64)[has_header] - 16
    claimed = (ne + FIXED_LEN - 1) // FIXED_LEN
    nsec = int(np.where(claimed > bound, bound + 1, claimed).sum())
    vals = out if out is not None else np.zeros(nsec * FIXED_LEN, dtype=NP_DTYPE[batch.etype])
    out_off = np.zeros(n + 1, dtype=np.uint64)
    st = np.zeros(n, dtype=np.int32) if return_status else None
    chain = make_chain(batch.etype, add_const)
    o = N.read_options(read_path, tail_rule)
    rc = L.cvec_decode_batch_opt(batch.etype, _p(batch.blob), _p(batch.byte_off), n, C.byref(chain), C.byref(o), _p(vals),
                                 vals.size, _p(out_off), _p(st) if st is not None else None)
    if rc != N.OK and not (return_status and rc <= N.ERR_SECTION_COUNT and st.any()):
        N.check(rc)
    return (vals, out_off, st) if return_status else (vals, out_off)
