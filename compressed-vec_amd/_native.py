"""ctypes binding of the C ABI in include/cvec.h (lib/libcvec_hip.so).

There is no CPU fallback: if the HIP library is missing every entry point raises."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CVEC_LIB_PATH") or os.path.join(_HERE, "lib", "libcvec_hip.so")

U64, U32, F32 = 0x10, 0x11, 0x12
PRED_EQ, PRED_ONEOF = 0, 1
MAX_PRED_VALUES = 16
RESULT_WORDS = 4
STATIC_SCRATCH_BYTES = 2 * 1024 * 1024
PATH_AUTO, PATH_SLOTS, PATH_VSTREAM = 0, 1, 2
TAIL_FORMAT, TAIL_REFERENCE = 0, 1

OK = 0
NOT_ENOUGH_SPACE, INPUT_TOO_SHORT, BAD_OFFSET, INVALID_SECTION_TYPE, INVALID_FORMAT, INVALID_NUM_ROWS, \
    WRONG_VECTOR_TYPE, SCROLL_ERR = range(1, 9)
ERR_SECTION_COUNT, ERR_BAD_ARGUMENT, ERR_HIP, ERR_NO_DEVICE, ERR_WORKSPACE, ERR_CAPACITY = 100, 101, 102, 103, 104, 105

_NAMES = {0: "Ok", 1: "NotEnoughSpace", 2: "InputTooShort", 3: "BadOffset", 4: "InvalidSectionType",
          5: "InvalidFormat", 6: "InvalidNumRows", 7: "WrongVectorType", 8: "ScrollErr",
          100: "SectionCountMismatch", 101: "BadArgument", 102: "HipError", 103: "NoDevice",
          104: "WorkspaceTooSmall", 105: "OutputTooSmall"}


class CodingError(Exception):
    """Mirror of the reference's CodingError enum (src/error.rs:2-11) plus library errors (>= 100)."""

    def __init__(self, code, *detail):
        self.code = int(code)
        self.detail = detail
        super().__init__("%s%s" % (_NAMES.get(self.code, "Error%d" % self.code),
                                   "(%s)" % ", ".join(map(str, detail)) if detail else ""))

    @property
    def name(self):
        return _NAMES.get(self.code, "Unknown")


class Pred(C.Structure):
    _fields_ = [("kind", C.c_int32), ("count", C.c_uint32), ("values", C.c_uint64 * MAX_PRED_VALUES)]


class SinkChain(C.Structure):
    _fields_ = [("add_const", C.c_int32), ("_pad", C.c_int32), ("add_const_bits", C.c_uint64)]


class DBatchView(C.Structure):
    _fields_ = [("etype", C.c_int32), ("_pad", C.c_int32), ("n_vectors", C.c_uint64), ("n_sections", C.c_uint64),
                ("n_bytes", C.c_uint64), ("d_bytes", C.c_void_p), ("d_byte_off", C.c_void_p),
                ("d_sec_base", C.c_void_p)]


class ReadOptions(C.Structure):
    """cvec_read_options: what one read call runs with (-1 = the process-wide default)."""
    _fields_ = [("read_path", C.c_int32), ("tail_rule", C.c_int32), ("indexed", C.c_int32), ("_pad", C.c_int32),
                ("d_gate", C.c_void_p)]


OPT_DEFAULT = -1
_PATHS = {None: OPT_DEFAULT, "auto": PATH_AUTO, "slots": PATH_SLOTS, "vstream": PATH_VSTREAM}
_TAILS = {None: OPT_DEFAULT, "format": TAIL_FORMAT, "reference": TAIL_REFERENCE}


def read_options(read_path=None, tail_rule=None, indexed=False, gate_ptr=None):
    """A ReadOptions from names: read_path None | "auto" | "slots" | "vstream", tail_rule None | "format" |
    "reference" (None = the process-wide default)."""
    o = ReadOptions()
    o.read_path = _PATHS[read_path] if not isinstance(read_path, int) else read_path
    o.tail_rule = _TAILS[tail_rule] if not isinstance(tail_rule, int) else tail_rule
    o.indexed = 1 if indexed else 0
    o.d_gate = gate_ptr
    return o


class MaskColumn(C.Structure):
    _fields_ = [("d_masks", C.c_void_p), ("d_sec_base", C.c_void_p), ("d_valid", C.c_void_p),
                ("n_sections", C.c_uint64)]


MAX_COLUMNS = 16
STREAM_U64_DELTA, STREAM_F64_XOR, STREAM_U64_PLAIN = 1, 2, 3

# Every symbol include/cvec.h declares, with its signature.
SIGNATURES = {
    "cvec_version": (C.c_char_p, []),
    "cvec_init": (C.c_int32, [C.c_int]),
    "cvec_last_hip_error": (C.c_char_p, []),
    "cvec_status_name": (C.c_char_p, [C.c_int32]),
    "cvec_set_read_path": (C.c_int32, [C.c_int32]),
    "cvec_set_tail_rule": (C.c_int32, [C.c_int32]),
    "cvec_read_options_effective": (C.c_int32, [C.POINTER(ReadOptions), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "cvec_filter_count_opt_dev": (C.c_int32, [C.POINTER(DBatchView), C.POINTER(Pred), C.POINTER(ReadOptions), C.c_void_p,
                                              C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "cvec_decode_opt_dev": (C.c_int32, [C.POINTER(DBatchView), C.POINTER(SinkChain), C.POINTER(ReadOptions), C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "cvec_filter_count_batch_opt": (C.c_int32, [C.c_int32, C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(Pred),
                                                C.POINTER(ReadOptions), C.POINTER(C.c_uint64), C.c_void_p, C.c_void_p]),
    "cvec_decode_batch_opt": (C.c_int32, [C.c_int32, C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(SinkChain),
                                          C.POINTER(ReadOptions), C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]),
    "cvec_timing_begin": (C.c_int32, [C.c_int32]),
    "cvec_timing_end": (C.c_int32, [C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float),
                                    C.POINTER(C.c_int32)]),
    "cvec_workspace_bytes": (C.c_size_t, [C.c_uint64, C.c_uint64]),
    "cvec_sec_base_dev": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.POINTER(C.c_uint64),
                                      C.c_void_p]),
    "cvec_filter_count_dev": (C.c_int32, [C.POINTER(DBatchView), C.POINTER(Pred), C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "cvec_filter_set_workspace_bytes": (C.c_size_t, [C.c_uint64, C.c_uint64]),
    "cvec_filter_count_set_dev": (C.c_int32, [C.POINTER(DBatchView), C.POINTER(C.c_uint64), C.c_uint32, C.c_void_p,
                                              C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "cvec_filter_count_static_dev": (C.c_int32, [C.POINTER(DBatchView), C.POINTER(Pred), C.c_void_p, C.c_void_p,
                                                 C.c_void_p, C.c_void_p, C.c_void_p]),
    "cvec_filter_count_gated_dev": (C.c_int32, [C.POINTER(DBatchView), C.POINTER(Pred), C.c_void_p, C.c_void_p,
                                                C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                                C.c_void_p]),
    "cvec_index_dev": (C.c_int32, [C.POINTER(DBatchView), C.c_void_p, C.c_size_t, C.c_void_p]),
    "cvec_filter_count_indexed_dev": (C.c_int32, [C.POINTER(DBatchView), C.POINTER(Pred), C.c_void_p, C.c_void_p,
                                                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                                  C.c_void_p]),
    "cvec_decode_dev": (C.c_int32, [C.POINTER(DBatchView), C.POINTER(SinkChain), C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "cvec_decode_indexed_dev": (C.c_int32, [C.POINTER(DBatchView), C.POINTER(SinkChain), C.c_void_p, C.c_void_p,
                                            C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "cvec_valid_sections_dev": (C.c_int32, [C.POINTER(DBatchView), C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p,
                                            C.c_void_p]),
    "cvec_appender_new": (C.c_int32, [C.c_int32, C.c_size_t, C.POINTER(C.c_void_p)]),
    "cvec_appender_free": (None, [C.c_void_p]),
    "cvec_appender_reset": (C.c_int32, [C.c_void_p]),
    "cvec_appender_num_elements": (C.c_uint64, [C.c_void_p]),
    "cvec_appender_append": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "cvec_appender_append_nulls": (C.c_int32, [C.c_void_p, C.c_size_t]),
    "cvec_appender_finish": (C.c_int32, [C.c_void_p, C.c_uint64, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t),
                                         C.POINTER(C.c_uint64)]),
    "cvec_appender_reader": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "cvec_appender_stats": (C.c_int32, [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "cvec_series_appender_new": (C.c_int32, [C.c_int32, C.c_uint64, C.POINTER(C.c_void_p)]),
    "cvec_series_appender_free": (None, [C.c_void_p]),
    "cvec_series_appender_reset": (C.c_int32, [C.c_void_p]),
    "cvec_series_appender_append_dev": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]),
    "cvec_series_appender_append_host": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]),
    "cvec_series_appender_append_nulls_dev": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]),
    "cvec_series_appender_seal": (C.c_int32, [C.c_void_p, C.c_int32, C.c_uint64, C.POINTER(C.c_uint64),
                                              C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_void_p]),
    "cvec_series_appender_emit_dev": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cvec_series_appender_unseal": (C.c_int32, [C.c_void_p]),
    "cvec_series_appender_stats": (C.c_int32, [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64),
                                               C.POINTER(C.c_uint64)]),
    "cvec_error_detail_dev": (C.c_int32, [C.POINTER(DBatchView), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cvec_multi_workspace_bytes": (C.c_size_t, [C.c_uint64]),
    "cvec_multi_filter_count_dev": (C.c_int32, [C.POINTER(MaskColumn), C.c_uint32, C.c_uint64, C.c_void_p,
                                                C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                                C.c_void_p]),
    "cvec_positions_workspace_bytes": (C.c_size_t, [C.c_uint64]),
    "cvec_match_positions_dev": (C.c_int32, [C.POINTER(MaskColumn), C.c_uint64, C.c_void_p, C.c_void_p,
                                             C.c_uint64, C.c_void_p, C.c_size_t, C.c_void_p]),
    "cvec_stream_max_packed_bytes": (C.c_size_t, [C.c_int32, C.c_uint64, C.c_uint64]),
    "cvec_stream_pack_workspace_bytes": (C.c_size_t, [C.c_uint64, C.c_uint64]),
    "cvec_stream_pack_dev": (C.c_int32, [C.c_int32, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p,
                                         C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                         C.c_void_p]),
    "cvec_stream_unpack_workspace_bytes": (C.c_size_t, [C.c_uint64, C.c_uint64]),
    "cvec_stream_unpack_dev": (C.c_int32, [C.c_int32, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64,
                                           C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                           C.c_void_p]),
    "cvec_hist_max_packed_bytes": (C.c_size_t, [C.c_uint64, C.c_uint32]),
    "cvec_hist_workspace_bytes": (C.c_size_t, [C.c_uint64, C.c_uint32]),
    "cvec_hist_delta_diff_dev": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_uint64,
                                             C.c_uint32, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p,
                                             C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "cvec_max_encoded_bytes": (C.c_size_t, [C.c_int32, C.c_uint64, C.c_uint64]),
    "cvec_encode_workspace_bytes": (C.c_size_t, [C.c_uint64, C.c_uint64]),
    "cvec_encode_dev": (C.c_int32, [C.c_int32, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p,
                                    C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                    C.c_void_p]),
    "cvec_filter_count_batch": (C.c_int32, [C.c_int32, C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(Pred),
                                            C.POINTER(C.c_uint64), C.c_void_p, C.c_void_p]),
    "cvec_decode_batch": (C.c_int32, [C.c_int32, C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(SinkChain),
                                      C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]),
    "cvec_encode_batch": (C.c_int32, [C.c_int32, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_size_t,
                                      C.c_void_p, C.POINTER(C.c_uint64)]),
    "cvec_allreduce_counts": (C.c_int32, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]),
    "cvec_comm_unique_id": (C.c_int32, [C.c_void_p]),
    "cvec_comm_init_rank": (C.c_int32, [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]),
    "cvec_comm_destroy": (C.c_int32, [C.c_void_p]),
    "cvec_debug_fail_chunk": (C.c_int32, [C.c_int64]),
}

_lib = None


def set_tail_rule(rule):
    """"format" (default): every octet decodes by the bitstream rule of the wire format, i.e. to the values that
    were encoded.  "reference": the last octets of a vector decode the way the reference crate's tail fallback
    does (cvec_set_tail_rule, CVEC_TAIL_REFERENCE).  The process-wide default; the read operations also take
    tail_rule= per call."""
    check(lib().cvec_set_tail_rule({"format": TAIL_FORMAT, "reference": TAIL_REFERENCE}[rule]))


def lib():
    """Loads lib/libcvec_hip.so; raises if it has not been built (no fallback path exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("HIP library %s is missing: run __graft_entry__.build() (or make -C "
                               "compressed-vec_amd/csrc); there is no CPU fallback" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(status, *detail):
    if status != OK:
        if status == ERR_HIP:
            raise CodingError(status, lib().cvec_last_hip_error().decode())
        raise CodingError(status, *detail)
