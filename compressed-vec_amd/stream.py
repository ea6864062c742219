"""The legacy NibblePack stream codecs (reference src/nibblepacking.rs:15-45, :239-325, :364-400) for
batches of streams: delta-coded increasing u64 streams and XOR-coded f64 streams.  A batch is n
streams back to back: values + elem_off[n+1] on the input side, bytes + byte_off[n+1] packed."""
import ctypes as C

import numpy as np
import torch

from . import _native as N
from .batch import _ptr, _stream_ptr

KINDS = {"u64_delta": N.STREAM_U64_DELTA, "f64_xor": N.STREAM_F64_XOR, "u64": N.STREAM_U64_PLAIN}


def _as_bits(values, kind, device):
    v = np.ascontiguousarray(values, dtype=np.float64 if kind == N.STREAM_F64_XOR else np.uint64)
    return torch.from_numpy(v.view(np.int64).copy()).to(device)


def pack_device(kind, d_vals, d_eo, n_values_total):
    """Device-resident pack: d_vals int64 bit patterns, d_eo int64[n+1].  Returns (d_bytes, d_byte_off,
    d_status, d_result); d_result[0] = bytes written."""
    kind = KINDS[kind] if isinstance(kind, str) else kind
    n, total, device = d_eo.numel() - 1, int(n_values_total), d_eo.device
    lib = N.lib()
    cap = lib.cvec_stream_max_packed_bytes(kind, n, total)
    d_bytes = torch.zeros((cap + 255) // 256 * 256, dtype=torch.uint8, device=device)
    d_bo = torch.empty(n + 1, dtype=torch.int64, device=device)
    d_st = torch.zeros(max(n, 1), dtype=torch.int32, device=device)
    d_res = torch.zeros(N.RESULT_WORDS, dtype=torch.int64, device=device)
    ws = torch.empty(lib.cvec_stream_pack_workspace_bytes(n, total), dtype=torch.uint8, device=device)
    with torch.cuda.device(device):
        N.check(lib.cvec_stream_pack_dev(kind, _ptr(d_vals), _ptr(d_eo), n, total, _ptr(d_bytes), d_bytes.numel(),
                                         _ptr(d_bo), _ptr(d_st), _ptr(d_res), _ptr(ws), ws.numel(), _stream_ptr()))
    return d_bytes, d_bo, d_st, d_res


def unpack_device(kind, d_bytes, d_bo, n_bytes, d_eo, n_values_total, d_out=None, ws=None):
    """Device-resident unpack into d_out (int64 bit patterns, n_values_total).  Returns (d_out, d_status,
    d_result)."""
    kind = KINDS[kind] if isinstance(kind, str) else kind
    n, total, device = d_eo.numel() - 1, int(n_values_total), d_eo.device
    lib = N.lib()
    if d_out is None:
        d_out = torch.zeros(max(total, 1), dtype=torch.int64, device=device)
    d_st = torch.zeros(max(n, 1), dtype=torch.int32, device=device)
    d_res = torch.zeros(N.RESULT_WORDS, dtype=torch.int64, device=device)
    if ws is None:
        ws = torch.empty(lib.cvec_stream_unpack_workspace_bytes(n, total), dtype=torch.uint8, device=device)
    with torch.cuda.device(device):
        N.check(lib.cvec_stream_unpack_dev(kind, _ptr(d_bytes), _ptr(d_bo), int(n_bytes), _ptr(d_eo), n, total,
                                           _ptr(d_out), _ptr(d_st), _ptr(d_res), _ptr(ws), ws.numel(), _stream_ptr()))
    return d_out, d_st, d_res


def pack_batch(kind, values, elem_off, device="cuda"):
    """pack_u64_delta / pack_f64_xor for every stream of the batch.  Returns (blob uint8, byte_off uint64,
    status int32)."""
    kind = KINDS[kind] if isinstance(kind, str) else kind
    eo = np.ascontiguousarray(elem_off, dtype=np.uint64)
    n = eo.size - 1
    d_vals = _as_bits(values, kind, device)
    d_eo = torch.from_numpy(eo.view(np.int64).copy()).to(device)
    d_bytes, d_bo, d_st, d_res = pack_device(kind, d_vals, d_eo, int(eo[-1]))
    nbytes = int(d_res[0].item())
    return (d_bytes[:nbytes].cpu().numpy(), d_bo.cpu().numpy().view(np.uint64), d_st.cpu().numpy()[:n])


def unpack_batch(kind, blob, byte_off, elem_off, device="cuda"):
    """unpack into a DeltaSink / unpack_f64_xor into a DoubleXorSink for every stream: returns (values,
    status); stream s's values are values[elem_off[s]:elem_off[s+1]] (undefined where status != 0)."""
    kind = KINDS[kind] if isinstance(kind, str) else kind
    bo = np.ascontiguousarray(byte_off, dtype=np.uint64)
    eo = np.ascontiguousarray(elem_off, dtype=np.uint64)
    n, total, nbytes = eo.size - 1, int(eo[-1]), int(bo[-1])
    d_bytes = torch.zeros((nbytes + 16 + 255) // 256 * 256, dtype=torch.uint8, device=device)
    d_bytes[:nbytes].copy_(torch.from_numpy(np.ascontiguousarray(blob, dtype=np.uint8)[:nbytes].copy()))
    d_bo = torch.from_numpy(bo.view(np.int64).copy()).to(device)
    d_eo = torch.from_numpy(eo.view(np.int64).copy()).to(device)
    d_out, d_st, _ = unpack_device(kind, d_bytes, d_bo, nbytes, d_eo, total)
    out = d_out[:total].cpu().numpy()
    vals = out.view(np.float64) if kind == N.STREAM_F64_XOR else out.view(np.uint64)
    return vals, d_st.cpu().numpy()[:n]


def pack_u64_delta(values):
    """nibblepacking::pack_u64_delta (src/nibblepacking.rs:15-23) of one stream."""
    v = np.ascontiguousarray(values, dtype=np.uint64)
    blob, _, _ = pack_batch(N.STREAM_U64_DELTA, v, [0, v.size])
    return blob.tobytes()


def unpack_u64_delta(buf, num_values):
    """nibblepacking::unpack into a DeltaSink (src/nibblepacking.rs:239-279, :364-377)."""
    b = np.frombuffer(bytes(buf), dtype=np.uint8)
    vals, st = unpack_batch(N.STREAM_U64_DELTA, b, [0, b.size], [0, num_values])
    if st[0]:
        raise N.CodingError(int(st[0]))
    return vals


def pack_f64_xor(values):
    """nibblepacking::pack_f64_xor (src/nibblepacking.rs:29-45) of one stream."""
    v = np.ascontiguousarray(values, dtype=np.float64)
    blob, _, st = pack_batch(N.STREAM_F64_XOR, v, [0, v.size])
    if st[0]:
        raise N.CodingError(int(st[0]))
    return blob.tobytes()


def unpack_f64_xor(buf, num_values):
    """nibblepacking::unpack_f64_xor into a DoubleXorSink (src/nibblepacking.rs:284-325, :392-400)."""
    b = np.frombuffer(bytes(buf), dtype=np.uint8)
    vals, st = unpack_batch(N.STREAM_F64_XOR, b, [0, b.size], [0, num_values])
    if st[0]:
        raise N.CodingError(int(st[0]))
    return vals


def pack_u64(values):
    """nibblepacking::pack_u64 (src/nibblepacking.rs:68-93) of one stream: the values as they are."""
    v = np.ascontiguousarray(values, dtype=np.uint64)
    blob, _, _ = pack_batch(N.STREAM_U64_PLAIN, v, [0, v.size])
    return blob.tobytes()


def unpack_u64(buf, num_values):
    """nibblepacking::unpack (src/nibblepacking.rs:364-377) into a plain value sink."""
    b = np.frombuffer(bytes(buf), dtype=np.uint8)
    vals, st = unpack_batch(N.STREAM_U64_PLAIN, b, [0, b.size], [0, num_values])
    if st[0]:
        raise N.CodingError(int(st[0]))
    return vals
