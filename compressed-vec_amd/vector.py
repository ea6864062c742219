"""Vector appenders and reader with the reference's API surface (src/vector.rs), backed by the HIP
library.  One vector is far below GPU launch granularity, so these classes are thin conveniences
over the batch operations in batch.py; bulk work should use EncodedBatch / DeviceBatch directly."""
import numpy as np

from . import _native as N
from .batch import (FIXED_LEN, NP_DTYPE, DeviceBatch, EncodedBatch, encode_batch, etype_of)
from .filter import VectorFilter
from .sink import AddConstSink, VecSink

HEADER_BYTES = 16
GROW_BYTES = 4096


class VectorAppender:
    """VectorAppender<T, W> (src/vector.rs:207-379).  Values are buffered on the host; finish() encodes
    them on the GPU with the type's default section writer (AutoEncoder for u32/u64, the XOR writer
    for f32) and returns the reference byte layout."""

    etype = None

    def __init__(self, initial_capacity):
        self._chunks = []
        self._len = 0            # elements appended so far (incl. nulls)
        self._wbuf = 0           # elements sitting in the 256-element write buffer
        self._null_sections = 0  # explicit null sections (src/vector.rs:321-327)
        self.initial_capacity = initial_capacity

    @classmethod
    def try_new(cls, initial_capacity):
        if initial_capacity < 8:   # the header write needs 8 bytes (src/vector.rs:232, :128-131)
            raise N.CodingError(N.NOT_ENOUGH_SPACE)
        return cls(initial_capacity)

    def num_elements(self):
        return self._len

    def reset(self):
        self._chunks, self._len, self._wbuf, self._null_sections = [], 0, 0, 0

    def append(self, value):
        self._chunks.append(np.array([value], dtype=NP_DTYPE[self.etype]))
        self._len += 1
        self._wbuf = (self._wbuf + 1) % FIXED_LEN

    def append_many(self, values):
        v = np.ascontiguousarray(values, dtype=NP_DTYPE[self.etype]).ravel()
        self._chunks.append(v)
        self._len += v.size
        self._wbuf = (self._wbuf + v.size) % FIXED_LEN

    def append_nulls(self, num_nulls):
        # bookkeeping of src/vector.rs:311-335: whole null sections are only counted while the write
        # buffer is empty
        left = int(num_nulls)
        if left and self._wbuf:
            fill = min(left, FIXED_LEN - self._wbuf)
            self._wbuf = (self._wbuf + fill) % FIXED_LEN
            left -= fill
        self._null_sections += left // FIXED_LEN
        self._wbuf = left % FIXED_LEN if left else self._wbuf
        if num_nulls:
            self._chunks.append(np.zeros(int(num_nulls), dtype=NP_DTYPE[self.etype]))
            self._len += int(num_nulls)

    def finish(self, total_num_rows):
        if self._len > total_num_rows:
            raise N.CodingError(N.INVALID_NUM_ROWS, total_num_rows, self._len)
        if total_num_rows > 0xffffffff:
            raise N.CodingError(N.INVALID_NUM_ROWS, total_num_rows, 0xffffffff)
        padded = -(-self._len // FIXED_LEN) * FIXED_LEN
        nulls = self._null_sections
        if total_num_rows > padded:   # finish() appends explicit null sections (src/vector.rs:357-359)
            nulls += -(-(total_num_rows - padded) // FIXED_LEN)
        vals = np.concatenate(self._chunks) if self._chunks else np.zeros(0, dtype=NP_DTYPE[self.etype])
        if total_num_rows > vals.size:
            vals = np.concatenate([vals, np.zeros(total_num_rows - vals.size, dtype=vals.dtype)])
        batch = encode_batch(self.etype, vals, np.array([0, vals.size], dtype=np.uint64))
        out = bytearray(batch.vector(0))
        out[12:14] = int(nulls & 0xffff).to_bytes(2, "little")
        self.reset()
        return bytes(out)

    def reader(self):
        """Reader over what has been written so far (src/vector.rs:372-378): the complete sections;
        values still in the 256-element write buffer are not visible yet."""
        flushed = self._len - self._wbuf
        vals = np.concatenate(self._chunks)[:flushed] if self._chunks else np.zeros(0, dtype=NP_DTYPE[self.etype])
        batch = encode_batch(self.etype, vals, np.array([0, flushed], dtype=np.uint64))
        out = bytearray(batch.vector(0))
        out[12:14] = int(self._null_sections & 0xffff).to_bytes(2, "little")
        return VectorReader.try_new(bytes(out), self.etype)

    def encode_all(self, collection):
        self.append_many(np.fromiter(collection, dtype=NP_DTYPE[self.etype]) if not isinstance(
            collection, np.ndarray) else collection)
        return self.finish(self._len)


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
        term = output._terminal()
        term.vec = np.concatenate([term.vec, vals])

    def filter_iter(self, sink):
        return VectorFilter(self, sink)

    def iterate(self):
        """All num_elements values (src/vector.rs:452, :512-556)."""
        sink = VecSink(self.etype)
        self.decode_to_sink(sink)
        return sink.vec[:self.num_elements()]
