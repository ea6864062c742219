"""Vector appenders and reader with the reference's API surface (src/vector.rs), backed by the HIP
library.  One vector is far below GPU launch granularity, so these classes are thin conveniences
over the batch operations in batch.py; bulk work should use EncodedBatch / DeviceBatch directly."""
import ctypes as C

import numpy as np

from . import _native as N
from .batch import (FIXED_LEN, NP_DTYPE, DeviceBatch, EncodedBatch, encode_batch, etype_of)
from .filter import VectorFilter
from .sink import AddConstSink, VecSink

HEADER_BYTES = 16


class VectorAppender:
    """VectorAppender<T, W> (src/vector.rs:207-379) over the library's appender handle
    (cvec_appender_new / _append / _append_nulls / _finish / _reader): the handle keeps the values, the
    fill of the open section and the explicit null-section count; finish() encodes on the GPU with the
    type's default section writer (AutoEncoder for u32/u64, the XOR writer for f32) and returns the
    reference byte layout, null-section count in the header included."""

    etype = None

    def __init__(self, initial_capacity):
        h = C.c_void_p()
        rc = N.lib().cvec_appender_new(self.etype, initial_capacity, C.byref(h))
        if rc:
            raise N.CodingError(rc)
        self._h = h
        self.initial_capacity = initial_capacity

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            N.lib().cvec_appender_free(h)

    @classmethod
    def try_new(cls, initial_capacity):
        return cls(initial_capacity)          # a capacity below 8 raises NotEnoughSpace (src/vector.rs:232, :128-131)

    def num_elements(self):
        return int(N.lib().cvec_appender_num_elements(self._h))

    def reset(self):
        N.check(N.lib().cvec_appender_reset(self._h))

    def append(self, value):
        v = np.array([value], dtype=NP_DTYPE[self.etype])
        N.check(N.lib().cvec_appender_append(self._h, v.ctypes.data, 1))

    def append_many(self, values):
        v = np.ascontiguousarray(values, dtype=NP_DTYPE[self.etype]).ravel()
        N.check(N.lib().cvec_appender_append(self._h, v.ctypes.data if v.size else None, v.size))

    def append_nulls(self, num_nulls):
        N.check(N.lib().cvec_appender_append_nulls(self._h, int(num_nulls)))

    def stats(self):
        """(bytes of the sections encoded so far, bytes of values the handle still holds raw)."""
        e, r = C.c_uint64(0), C.c_uint64(0)
        N.check(N.lib().cvec_appender_stats(self._h, C.byref(e), C.byref(r)))
        return e.value, r.value

    def _bytes_of(self, call):
        """Runs a finish / reader call twice: once to learn the size, once into a buffer of that size."""
        need = C.c_size_t(0)
        rc = call(None, 0, need)
        if rc not in (N.OK, N.ERR_CAPACITY):
            return rc, b""
        buf = (C.c_uint8 * max(need.value, 1))()
        rc = call(buf, need.value, need)
        return rc, bytes(buf[:need.value])

    def finish(self, total_num_rows):
        detail = (C.c_uint64 * 2)(0, 0)
        rc, out = self._bytes_of(lambda buf, cap, need: N.lib().cvec_appender_finish(
            self._h, int(total_num_rows), buf, cap, C.byref(need), detail))
        if rc == N.INVALID_NUM_ROWS:
            raise N.CodingError(rc, int(detail[0]), int(detail[1]))
        N.check(rc)
        return out

    def reader(self):
        """Reader over what has been written so far (src/vector.rs:372-378): the complete sections;
        values still in the 256-element write buffer are not visible yet."""
        rc, out = self._bytes_of(lambda buf, cap, need: N.lib().cvec_appender_reader(self._h, buf, cap, C.byref(need)))
        N.check(rc)
        return VectorReader.try_new(out, self.etype)

    def encode_all(self, collection):
        self.append_many(np.fromiter(collection, dtype=NP_DTYPE[self.etype]) if not isinstance(
            collection, np.ndarray) else collection)
        return self.finish(self.num_elements())


class VectorU32Appender(VectorAppender):
    etype = N.U32


class VectorU64Appender(VectorAppender):
    etype = N.U64


class VectorF32XorAppender(VectorAppender):
    etype = N.F32


class VectorReader:
    """VectorReader<T> (src/vector.rs:395-465) over the bytes of one finished vector."""

    def __init__(self, vect_bytes, etype):
        self.vect_bytes = bytes(vect_bytes)
        self.etype = etype
        self._dev = None
        self._vals = None

    @classmethod
    def try_new(cls, vect_bytes, dtype):
        etype = etype_of(dtype)
        b = bytes(vect_bytes)
        # same order of checks as src/vector.rs:404-414 (scroll's pread errors included)
        if len(b) == 0:
            raise N.CodingError(N.BAD_OFFSET, 0)
        if len(b) < 4:
            raise N.CodingError(N.NOT_ENOUGH_SPACE)
        from_header = int.from_bytes(b[0:4], "little")
        if len(b) <= 5:
            raise N.CodingError(N.BAD_OFFSET, 5)
        if len(b) < from_header + 4:
            raise N.CodingError(N.INPUT_TOO_SHORT)
        if b[5] != etype:
            raise N.CodingError(N.WRONG_VECTOR_TYPE, b[5])
        if len(b) < HEADER_BYTES:
            raise N.CodingError(N.INPUT_TOO_SHORT)
        return cls(b, etype)

    def num_elements(self):
        return int.from_bytes(self.vect_bytes[8:12], "little")

    def total_bytes(self):
        return len(self.vect_bytes)

    def get_stats(self):
        return {"num_elements": self.num_elements(),
                "num_null_sections": int.from_bytes(self.vect_bytes[12:14], "little")}

    def _device_batch(self):
        if self._dev is None:
            host = EncodedBatch.from_vectors(self.etype, [self.vect_bytes])
            self._dev = DeviceBatch.upload(host)
        return self._dev

    def decode_to_sink(self, output):
        """Decodes every section through the sink stack `output` (VecSink or AddConstSink(VecSink))."""
        if output.etype != self.etype:
            raise N.CodingError(N.WRONG_VECTOR_TYPE, output.etype)
        vals, _off, status = self._device_batch().decode(add_const=output._add_const(), return_status=True)
        if status[0] != N.OK:
            raise N.CodingError(int(status[0]))
        output._terminal()._accept(vals)

    def sections(self):
        """FixedSectIterator over the vector (src/section.rs:846-884): one SectionRef per section, with the
        section's type byte and size; SectionRef.decode(sink) pushes that section's 256 values through a sink
        stack (FixedSectReader::decode_to_sink, src/section.rs:274-290).  A header that does not parse raises
        where the reference's iterator yields the error."""
        b, w = self.vect_bytes, (8 if self.etype == N.U64 else 4)
        pos, k, out = HEADER_BYTES, 0, []
        while pos < len(b):
            rem, kind = len(b) - pos, b[pos]
            if kind == 0:
                nb = 1
            elif kind == 5:
                if rem < 1 + w:
                    raise N.CodingError(N.INPUT_TOO_SHORT)
                nb = 1 + w
            elif kind in (1, 3, 6):
                if rem <= 1:
                    raise N.CodingError(N.BAD_OFFSET, 1)
                if rem < 3:
                    raise N.CodingError(N.NOT_ENOUGH_SPACE)
                n16 = int.from_bytes(b[pos + 1:pos + 3], "little")
                nb = n16 + 3 if kind == 1 else (n16 + 12 if kind == 3 else n16)
                if nb > rem:
                    raise N.CodingError(N.SCROLL_ERR)
                if nb < 3:
                    raise N.CodingError(N.INVALID_FORMAT)
            else:
                raise N.CodingError(N.INVALID_SECTION_TYPE, kind)
            out.append(SectionRef(self, k, kind, pos, nb))
            pos += nb
            k += 1
        return out

    def _decoded(self):
        """The whole vector decoded once on the GPU (values, per-vector status), kept for section-wise and
        ranged reads."""
        if self._vals is None:
            vals, _off, status = self._device_batch().decode(return_status=True)
            self._vals, self._vals_status = vals, int(status[0])
            self._vals_valid = int(self._device_batch().valid_sections()[0][0])
        return self._vals

    def filter_iter(self, sink):
        return VectorFilter(self, sink)

    def iterate(self, start=0, stop=None):
        """VectorItemIter (src/vector.rs:452, :512-556): the vector's num_elements values, or those at
        positions [start, stop).  The iterator of the reference decodes section after section as it goes; here
        the vector is decoded once and the range is a slice of it.  A section that fails to decode raises, as
        the reference's expect() panics."""
        n = self.num_elements()
        stop = n if stop is None else min(stop, n)
        vals = self._decoded()
        if self._vals_status != N.OK and stop > FIXED_LEN * self._vals_valid:
            raise N.CodingError(self._vals_status)
        return vals[start:stop]


class SectionRef:
    """One section of a vector (FixedSectEnum, src/section.rs:226-250): type byte, offset and size in the
    vector's bytes, and decode()."""

    def __init__(self, reader, index, kind, offset, num_bytes):
        self.reader, self.index, self.kind, self.offset, self.num_bytes = reader, index, kind, offset, num_bytes

    def is_null(self):
        return self.kind == 0

    def decode(self, sink):
        """Pushes the section's 256 values through `sink` (Section256Sink, VecSink, or AddConstSink over
        one of them)."""
        r = self.reader
        if sink.etype != r.etype:
            raise N.CodingError(N.WRONG_VECTOR_TYPE, sink.etype)
        vals = r._decoded()
        if self.index >= r._vals_valid:
            raise N.CodingError(r._vals_status)
        v = vals[FIXED_LEN * self.index:FIXED_LEN * (self.index + 1)]
        base = sink._add_const()
        if base is not None:
            # AddConstSink (src/sink.rs:289-320): wrapping add for integers, IEEE add for f32; a Null section
            # forwards the base itself
            if self.kind == 0:
                v = np.full(FIXED_LEN, base, dtype=v.dtype)
            elif v.dtype.kind == "f":
                v = v + v.dtype.type(base)
            else:
                v = (v.astype(np.uint64) + np.uint64(int(base) & 0xffffffffffffffff)).astype(v.dtype)
        sink._terminal()._accept(v)
