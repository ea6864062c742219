"""ctypes front-end of the CPU oracle (oracle/cvec_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The shipped GPU path never imports this package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# CVEC_ORACLE_LIB: another build of the same source (the sanitizer build, `make -C oracle sanitize`) instead
_LIB_OVERRIDE = os.environ.get("CVEC_ORACLE_LIB")
_LIB_PATH = _LIB_OVERRIDE or os.path.join(_HERE, "libcvec_oracle.so")

OK, NOT_ENOUGH_SPACE, INPUT_TOO_SHORT, BAD_OFFSET, INVALID_SECTION_TYPE, INVALID_FORMAT, \
    INVALID_NUM_ROWS, WRONG_VECTOR_TYPE, SCROLL_ERR = range(9)
U64, U32, F32 = 0x10, 0x11, 0x12
SECT_NULL, SECT_NP, SECT_DELTA, SECT_CONST, SECT_XOR = 0, 1, 3, 5, 6
PRED_EQ, PRED_ONEOF = 0, 1

NP_DTYPE = {U64: np.uint64, U32: np.uint32, F32: np.float32}


def build(force=False):
    """Compile the oracle with gcc (a few hundred ms)."""
    src = os.path.join(_HERE, "cvec_oracle.c")
    if _LIB_OVERRIDE:
        return _LIB_PATH
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libcvec_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


class OracleError(Exception):
    def __init__(self, code):
        super().__init__("oracle error code %d" % code)
        self.code = code


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        u8p, u64p, vp, szp = C.POINTER(C.c_uint8), C.POINTER(C.c_uint64), C.c_void_p, C.POINTER(C.c_size_t)
        sigs = {
            "ocv_nibble_pack8": (C.c_int64, [vp, vp, C.c_size_t, C.c_size_t]),
            "ocv_pack_u64": (C.c_int64, [vp, C.c_size_t, vp, C.c_size_t, C.c_size_t]),
            "ocv_nibble_unpack8": (C.c_int, [vp, C.c_size_t, vp, szp]),
            "ocv_unpack8_u32": (C.c_int, [vp, C.c_size_t, vp, C.POINTER(C.c_int), szp]),
            "ocv_unpack_u64": (C.c_int, [vp, C.c_size_t, vp, C.c_size_t, szp]),
            "ocv_pack_u64_delta": (C.c_int64, [vp, C.c_size_t, vp, C.c_size_t]),
            "ocv_unpack_u64_delta": (C.c_int, [vp, C.c_size_t, vp, C.c_size_t, szp]),
            "ocv_pack_f64_xor": (C.c_int64, [vp, C.c_size_t, vp, C.c_size_t]),
            "ocv_unpack_f64_xor": (C.c_int, [vp, C.c_size_t, vp, C.c_size_t, szp]),
            "ocv_compress_geom_nonincreasing": (C.c_int64, [C.c_uint16, C.c_double, C.c_double, C.c_uint8, vp,
                                                            C.c_size_t, vp, C.c_size_t]),
            "ocv_delta_diff_pack": (C.c_int, [vp, C.c_size_t, C.c_size_t, vp, vp, C.c_size_t, szp, C.POINTER(C.c_int),
                                              szp]),
            "ocv_set_ref_tail_quirk": (None, [C.c_int]),
            "ocv_section_write": (C.c_int64, [C.c_int, vp, vp, C.c_size_t, C.c_size_t]),
            "ocv_section_write_kind": (C.c_int64, [C.c_int, C.c_int, vp, vp, C.c_size_t, C.c_size_t]),
            "ocv_section_parse": (C.c_int, [C.c_int, vp, C.c_size_t, C.POINTER(C.c_int), szp]),
            "ocv_section_decode": (C.c_int, [C.c_int, vp, C.c_size_t, vp]),
            "ocv_section_delta_info": (C.c_int, [C.c_int, vp, C.c_size_t, u64p, C.POINTER(C.c_int)]),
            "ocv_appender_new": (vp, [C.c_int, C.c_size_t]),
            "ocv_appender_free": (None, [vp]),
            "ocv_appender_append": (C.c_int, [vp, vp]),
            "ocv_appender_append_many": (C.c_int, [vp, vp, C.c_size_t]),
            "ocv_appender_append_nulls": (C.c_int, [vp, C.c_size_t]),
            "ocv_appender_num_elements": (C.c_size_t, [vp]),
            "ocv_appender_bytes": (vp, [vp, szp]),
            "ocv_appender_finish": (C.c_int64, [vp, C.c_size_t, vp, C.c_size_t]),
            "ocv_encode_all": (C.c_int64, [C.c_int, vp, C.c_size_t, vp, C.c_size_t]),
            "ocv_max_encoded_size": (C.c_size_t, [C.c_int, C.c_size_t]),
            "ocv_reader_check": (C.c_int, [C.c_int, vp, C.c_size_t, u64p]),
            "ocv_error_detail": (C.c_int, [C.c_int, vp, C.c_size_t, u64p]),
            "ocv_reader_num_elements": (C.c_int, [vp, C.c_size_t, C.POINTER(C.c_uint32), C.POINTER(C.c_uint16)]),
            "ocv_reader_sections": (C.c_int, [C.c_int, vp, C.c_size_t, szp, vp, C.c_size_t]),
            "ocv_decode_to_vec": (C.c_int, [C.c_int, vp, C.c_size_t, C.c_int, vp, vp, C.c_size_t, szp]),
            "ocv_iterate": (C.c_int, [C.c_int, vp, C.c_size_t, vp, C.c_size_t, szp]),
            "ocv_filter_count": (C.c_int, [C.c_int, vp, C.c_size_t, C.c_int, vp, C.c_size_t, u64p, vp,
                                           C.c_size_t, szp]),
            "ocv_match_positions": (C.c_int, [vp, C.c_size_t, C.c_int, vp, C.c_size_t, szp]),
            "ocv_multi_filter_count": (C.c_int, [C.c_int, vp, vp, C.c_size_t, vp, u64p, vp, C.c_size_t, szp]),
            "ocv_batch_filter_count": (C.c_int, [C.c_int, vp, vp, C.c_size_t, C.c_int, vp, C.c_size_t, u64p,
                                                 vp, vp]),
            "ocv_batch_decode": (C.c_int, [C.c_int, vp, vp, C.c_size_t, C.c_int, vp, vp, vp, vp]),
            "ocv_batch_encode": (C.c_int64, [C.c_int, vp, vp, C.c_size_t, vp, C.c_size_t, vp]),
        }
        for name, (res, args) in sigs.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _bytes_arr(b):
    return np.frombuffer(bytes(b), dtype=np.uint8).copy() if not isinstance(b, np.ndarray) else \
        np.ascontiguousarray(b, dtype=np.uint8)


def _check(rc):
    if rc != 0:
        raise OracleError(rc)


# ---- L1 ---------------------------------------------------------------------------------

def nibble_pack8(vals, cap=512, off=0):
    v = np.ascontiguousarray(vals, dtype=np.uint64)
    assert v.size == 8
    out = np.zeros(cap, dtype=np.uint8)
    r = lib().ocv_nibble_pack8(_ptr(v), _ptr(out), cap, off)
    if r < 0:
        raise OracleError(-r)
    return out[:r].tobytes()


def pack_u64(vals, cap=None, off=0):
    v = np.ascontiguousarray(vals, dtype=np.uint64)
    cap = cap if cap is not None else off + 66 * ((v.size + 7) // 8) + 16
    out = np.zeros(cap, dtype=np.uint8)
    r = lib().ocv_pack_u64(_ptr(v), v.size, _ptr(out), cap, off)
    if r < 0:
        raise OracleError(-r)
    return out[:r].tobytes()


def nibble_unpack8(buf):
    b = _bytes_arr(buf)
    out = np.zeros(8, dtype=np.uint64)
    c = C.c_size_t(0)
    _check(lib().ocv_nibble_unpack8(_ptr(b), b.size, _ptr(out), C.byref(c)))
    return out, c.value


def unpack8_u32(buf):
    b = _bytes_arr(buf)
    out = np.zeros(8, dtype=np.uint32)
    c = C.c_size_t(0)
    z = C.c_int(0)
    _check(lib().ocv_unpack8_u32(_ptr(b), b.size, _ptr(out), C.byref(z), C.byref(c)))
    return out, c.value


def unpack_u64(buf, nvalues):
    b = _bytes_arr(buf)
    out = np.zeros((nvalues + 7) // 8 * 8, dtype=np.uint64)
    c = C.c_size_t(0)
    _check(lib().ocv_unpack_u64(_ptr(b), b.size, _ptr(out), nvalues, C.byref(c)))
    return out, c.value


def pack_u64_delta(vals):
    v = np.ascontiguousarray(vals, dtype=np.uint64)
    cap = 66 * ((v.size + 7) // 8) + 16
    out = np.zeros(cap, dtype=np.uint8)
    r = lib().ocv_pack_u64_delta(_ptr(v), v.size, _ptr(out), cap)
    if r < 0:
        raise OracleError(-r)
    return out[:r].tobytes()


def unpack_u64_delta(buf, nvalues):
    b = _bytes_arr(buf)
    out = np.zeros((nvalues + 7) // 8 * 8, dtype=np.uint64)
    c = C.c_size_t(0)
    _check(lib().ocv_unpack_u64_delta(_ptr(b), b.size, _ptr(out), nvalues, C.byref(c)))
    return out, c.value


def compress_geom_nonincreasing(num_buckets, initial_bucket, multiplier, format_code, values):
    v = np.ascontiguousarray(values, dtype=np.uint64)
    cap = 21 + 66 * ((v.size + 7) // 8) + 16
    out = np.zeros(cap, dtype=np.uint8)
    r = lib().ocv_compress_geom_nonincreasing(num_buckets, initial_bucket, multiplier, format_code, _ptr(v), v.size,
                                              _ptr(out), cap)
    if r < 0:
        raise OracleError(-r)
    return out[:r].tobytes()


class DeltaDiffPackSink:
    """histogram.rs:92-171 as the reference drives it: feed(packed) = unpack(packed, sink, num_buckets) then
    finish(); returns (bytes appended by this histogram, value_dropped, bytes consumed)."""

    def __init__(self, num_buckets):
        self.num_buckets = num_buckets
        self.last = np.zeros(max(num_buckets, 1), dtype=np.uint64)

    def reset(self):
        self.last[:] = 0

    def feed(self, packed):
        b = _bytes_arr(packed)
        cap = 66 * ((self.num_buckets + 7) // 8 + 1) + 16
        out = np.zeros(cap, dtype=np.uint8)
        off, c, dropped = C.c_size_t(0), C.c_size_t(0), C.c_int(0)
        _check(lib().ocv_delta_diff_pack(_ptr(b), b.size, self.num_buckets, _ptr(self.last), _ptr(out), cap,
                                         C.byref(off), C.byref(dropped), C.byref(c)))
        return out[:off.value].tobytes(), bool(dropped.value), c.value


def pack_f64_xor(vals):
    v = np.ascontiguousarray(vals, dtype=np.float64)
    cap = 8 + 66 * ((v.size + 7) // 8) + 16
    out = np.zeros(cap, dtype=np.uint8)
    r = lib().ocv_pack_f64_xor(_ptr(v), v.size, _ptr(out), cap)
    if r < 0:
        raise OracleError(-r)
    return out[:r].tobytes()


def unpack_f64_xor(buf, nvalues):
    b = _bytes_arr(buf)
    out = np.zeros(1 + (nvalues + 6) // 8 * 8, dtype=np.float64)
    c = C.c_size_t(0)
    _check(lib().ocv_unpack_f64_xor(_ptr(b), b.size, _ptr(out), nvalues, C.byref(c)))
    return out, c.value


# ---- L3 ---------------------------------------------------------------------------------

def section_write(etype, vals, kind=None, cap=4096, off=0):
    v = np.ascontiguousarray(vals, dtype=NP_DTYPE[etype])
    assert v.size == 256
    out = np.zeros(cap, dtype=np.uint8)
    if kind is None:
        r = lib().ocv_section_write(etype, _ptr(v), _ptr(out), cap, off)
    else:
        r = lib().ocv_section_write_kind(etype, kind, _ptr(v), _ptr(out), cap, off)
    if r < 0:
        raise OracleError(-r)
    return out[:r].tobytes()


def section_parse(etype, buf):
    b = _bytes_arr(buf)
    kind = C.c_int(0)
    nb = C.c_size_t(0)
    _check(lib().ocv_section_parse(etype, _ptr(b), b.size, C.byref(kind), C.byref(nb)))
    return kind.value, nb.value


def section_decode(etype, buf):
    b = _bytes_arr(buf)
    out = np.zeros(256, dtype=NP_DTYPE[etype])
    _check(lib().ocv_section_decode(etype, _ptr(b), b.size, _ptr(out)))
    return out


def section_delta_info(etype, buf):
    b = _bytes_arr(buf)
    base = C.c_uint64(0)
    nbits = C.c_int(0)
    _check(lib().ocv_section_delta_info(etype, _ptr(b), b.size, C.byref(base), C.byref(nbits)))
    return base.value, nbits.value


# ---- L4 ---------------------------------------------------------------------------------

class Appender:
    """Mirror of VectorAppender<T, W> (vector.rs:207-379) on the CPU oracle."""

    def __init__(self, etype, initial_capacity=4096):
        self.etype = etype
        self._h = lib().ocv_appender_new(etype, initial_capacity)
        if not self._h:
            raise OracleError(NOT_ENOUGH_SPACE)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().ocv_appender_free(self._h)
            self._h = None

    def append(self, value):
        v = np.array([value], dtype=NP_DTYPE[self.etype])
        _check(lib().ocv_appender_append(self._h, _ptr(v)))

    def append_many(self, values):
        v = np.ascontiguousarray(values, dtype=NP_DTYPE[self.etype])
        _check(lib().ocv_appender_append_many(self._h, _ptr(v), v.size))

    def append_nulls(self, n):
        _check(lib().ocv_appender_append_nulls(self._h, n))

    def num_elements(self):
        return lib().ocv_appender_num_elements(self._h)

    def reader_bytes(self):
        n = C.c_size_t(0)
        p = lib().ocv_appender_bytes(self._h, C.byref(n))
        return C.string_at(p, n.value)

    def finish(self, total_rows):
        cap = lib().ocv_max_encoded_size(self.etype, max(total_rows, self.num_elements())) + 4096
        out = np.zeros(cap, dtype=np.uint8)
        r = lib().ocv_appender_finish(self._h, total_rows, _ptr(out), cap)
        if r < 0:
            raise OracleError(-r)
        return out[:r].tobytes()

    def encode_all(self, values):
        v = np.ascontiguousarray(values, dtype=NP_DTYPE[self.etype])
        self.append_many(v)
        return self.finish(v.size)


def encode_all(etype, values):
    v = np.ascontiguousarray(values, dtype=NP_DTYPE[etype])
    cap = lib().ocv_max_encoded_size(etype, v.size)
    out = np.zeros(cap, dtype=np.uint8)
    r = lib().ocv_encode_all(etype, _ptr(v), v.size, _ptr(out), cap)
    if r < 0:
        raise OracleError(-r)
    return out[:r].tobytes()


def reader_check(etype, buf):
    b = _bytes_arr(buf)
    d = C.c_uint64(0)
    return lib().ocv_reader_check(etype, _ptr(b), b.size, C.byref(d)), d.value


def error_detail(etype, buf):
    """(status, section index at which the iteration stops or 2**64-1 for the vector header, payload)."""
    b = _bytes_arr(buf)
    d = (C.c_uint64 * 2)(0, 0)
    rc = lib().ocv_error_detail(etype, _ptr(b), b.size, d)
    return rc, int(d[0]), int(d[1])


def reader_num_elements(buf):
    b = _bytes_arr(buf)
    n = C.c_uint32(0)
    ns = C.c_uint16(0)
    _check(lib().ocv_reader_num_elements(_ptr(b), b.size, C.byref(n), C.byref(ns)))
    return n.value, ns.value


def reader_sections(etype, buf):
    b = _bytes_arr(buf)
    cnt = C.c_size_t(0)
    kinds = np.zeros(max(16, b.size), dtype=np.uint8)
    _check(lib().ocv_reader_sections(etype, _ptr(b), b.size, C.byref(cnt), _ptr(kinds), kinds.size))
    return kinds[:cnt.value].copy()


def decode_to_vec(etype, buf, add_base=None):
    b = _bytes_arr(buf)
    cap = 256 * max(1, b.size)  # every section is >= 1 byte
    cap = min(cap, 256 * (len(reader_sections(etype, b)) + 1))
    out = np.zeros(cap, dtype=NP_DTYPE[etype])
    n = C.c_size_t(0)
    base = np.array([add_base if add_base is not None else 0], dtype=NP_DTYPE[etype])
    _check(lib().ocv_decode_to_vec(etype, _ptr(b), b.size, int(add_base is not None), _ptr(base), _ptr(out),
                                   cap, C.byref(n)))
    return out[:n.value].copy()


def iterate(etype, buf):
    b = _bytes_arr(buf)
    nel, _ = reader_num_elements(b)
    out = np.zeros(max(nel, 1), dtype=NP_DTYPE[etype])
    n = C.c_size_t(0)
    _check(lib().ocv_iterate(etype, _ptr(b), b.size, _ptr(out), out.size, C.byref(n)))
    return out[:n.value].copy()


def filter_count(etype, buf, pred, return_masks=False):
    """pred: scalar → EqualsSink; list/tuple → OneOfSink."""
    b = _bytes_arr(buf)
    if isinstance(pred, (list, tuple, np.ndarray)):
        kind, pv = PRED_ONEOF, np.ascontiguousarray(pred, dtype=NP_DTYPE[etype])
    else:
        kind, pv = PRED_EQ, np.array([pred], dtype=NP_DTYPE[etype])
    cnt = C.c_uint64(0)
    ns = C.c_size_t(0)
    maxs = b.size
    masks = np.zeros(32 * max(maxs, 1), dtype=np.uint8) if return_masks else None
    _check(lib().ocv_filter_count(etype, _ptr(b), b.size, kind, _ptr(pv), pv.size, C.byref(cnt),
                                  _ptr(masks), maxs if return_masks else 0, C.byref(ns)))
    if return_masks:
        return cnt.value, masks[:32 * ns.value].copy()
    return cnt.value


def match_positions(masks, ref_quirk=False):
    m = np.ascontiguousarray(masks, dtype=np.uint8)
    ns = m.size // 32
    pos = np.zeros(ns * 256 + 1, dtype=np.uint64)
    n = C.c_size_t(0)
    _check(lib().ocv_match_positions(_ptr(m), ns, int(ref_quirk), _ptr(pos), pos.size, C.byref(n)))
    return pos[:n.value].copy()


def multi_filter_count(etype, bufs, preds, return_masks=False):
    arrs = [_bytes_arr(b) for b in bufs]
    k = len(arrs)
    ptrs = (C.c_void_p * k)(*[a.ctypes.data for a in arrs])
    lens = (C.c_size_t * k)(*[a.size for a in arrs])
    pv = np.ascontiguousarray(preds, dtype=NP_DTYPE[etype])
    cnt = C.c_uint64(0)
    ns = C.c_size_t(0)
    maxs = max(a.size for a in arrs)
    masks = np.zeros(32 * maxs, dtype=np.uint8)
    _check(lib().ocv_multi_filter_count(etype, ptrs, lens, k, _ptr(pv), C.byref(cnt), _ptr(masks), maxs,
                                        C.byref(ns)))
    if return_masks:
        return cnt.value, masks[:32 * ns.value].copy()
    return cnt.value


# ---- batches ----------------------------------------------------------------------------

def batch_encode(etype, values, elem_off):
    v = np.ascontiguousarray(values, dtype=NP_DTYPE[etype])
    eo = np.ascontiguousarray(elem_off, dtype=np.uint64)
    n = eo.size - 1
    cap = int(sum(lib().ocv_max_encoded_size(etype, int(eo[i + 1] - eo[i])) for i in range(n))) if n < 4096 \
        else int(n * lib().ocv_max_encoded_size(etype, int((eo[1:] - eo[:-1]).max())))
    out = np.zeros(cap + 64, dtype=np.uint8)
    bo = np.zeros(n + 1, dtype=np.uint64)
    r = lib().ocv_batch_encode(etype, _ptr(v), _ptr(eo), n, _ptr(out), cap, _ptr(bo))
    if r < 0:
        raise OracleError(-r)
    return out[:r].copy(), bo


def batch_filter_count(etype, blob, byte_off, pred):
    b = np.ascontiguousarray(blob, dtype=np.uint8)
    bo = np.ascontiguousarray(byte_off, dtype=np.uint64)
    n = bo.size - 1
    if isinstance(pred, (list, tuple, np.ndarray)):
        kind, pv = PRED_ONEOF, np.ascontiguousarray(pred, dtype=NP_DTYPE[etype])
    else:
        kind, pv = PRED_EQ, np.array([pred], dtype=NP_DTYPE[etype])
    total = C.c_uint64(0)
    per = np.zeros(n, dtype=np.uint64)
    st = np.zeros(n, dtype=np.int32)
    rc = lib().ocv_batch_filter_count(etype, _ptr(b), _ptr(bo), n, kind, _ptr(pv), pv.size, C.byref(total),
                                      _ptr(per), _ptr(st))
    return total.value, per, st, rc


def batch_decode(etype, blob, byte_off, out_off, add_base=None):
    b = np.ascontiguousarray(blob, dtype=np.uint8)
    bo = np.ascontiguousarray(byte_off, dtype=np.uint64)
    oo = np.ascontiguousarray(out_off, dtype=np.uint64)
    n = bo.size - 1
    out = np.zeros(int(oo[-1]), dtype=NP_DTYPE[etype])
    st = np.zeros(n, dtype=np.int32)
    base = np.array([add_base if add_base is not None else 0], dtype=NP_DTYPE[etype])
    rc = lib().ocv_batch_decode(etype, _ptr(b), _ptr(bo), n, int(add_base is not None), _ptr(base), _ptr(out),
                                _ptr(oo), _ptr(st))
    return out, st, rc


class ref_tail_quirk:
    """with oracle.ref_tail_quirk(): ...  decodes the way the reference's tail fallback does
    (nibblepack_simd.rs:433-450, :546-554: zeros for a 5..8-nibble octet with fewer than 34 bytes left
    in the vector slice).  The GPU library's CVEC_TAIL_REFERENCE rule is compared with this mode."""

    def __enter__(self):
        lib().ocv_set_ref_tail_quirk(1)
        return self

    def __exit__(self, *exc):
        lib().ocv_set_ref_tail_quirk(0)
        return False


def tail_quirk_census(etype, blob, byte_off, pred=None):
    """How far the reference crate's decoder (tail fallback bug included) would differ from the wire
    format's plain bitstream rule on this batch: (vectors whose decoded values differ, decoded
    elements that differ, filter-count difference for `pred` or None)."""
    bo = np.ascontiguousarray(byte_off, dtype=np.uint64)
    n = bo.size - 1
    nsec = np.zeros(n + 1, dtype=np.uint64)
    for v in range(n):
        ne = int(np.frombuffer(blob[int(bo[v]) + 8:int(bo[v]) + 12].tobytes(), dtype="<u4")[0])
        nsec[v + 1] = nsec[v] + np.uint64((ne + 255) // 256 * 256)
    good, st, rc = batch_decode(etype, blob, bo, nsec)
    assert rc == 0
    with ref_tail_quirk():
        bad, st, rc = batch_decode(etype, blob, bo, nsec)
        assert rc == 0
        cq = batch_filter_count(etype, blob, bo, pred)[0] if pred is not None else None
    bits = np.uint64 if etype == U64 else np.uint32
    diff = good.view(bits) != bad.view(bits)
    per_vec = np.add.reduceat(diff.astype(np.int64), nsec[:-1].astype(np.int64)) if n else np.zeros(0, dtype=np.int64)
    dc = None
    if pred is not None:
        dc = int(cq) - int(batch_filter_count(etype, blob, bo, pred)[0])
    return int((per_vec > 0).sum()), int(diff.sum()), dc
