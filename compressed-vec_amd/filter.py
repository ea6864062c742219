"""Filter-and-count path (reference src/filter.rs).  A filter sink is a predicate descriptor; the
decode kernel compares every decoded octet against it in registers and never writes the decoded
vector.  `VectorFilter` is what `VectorReader.filter_iter` returns: iterating it yields the 256-bit
mask of each section (32 bytes: byte = octet, LSB = first element)."""
import numpy as np

from . import _native as N
from .batch import etype_of, make_pred


class _FilterSink:
    def __init__(self, dtype, pred):
        self.etype = etype_of(dtype)
        self.pred_value = pred
        self.pred = make_pred(self.etype, pred)


class EqualsSink(_FilterSink):
    """GenericFilterSink<T, EqualsIF> (src/filter.rs:129-139): element == value (IEEE == for f32)."""

    def __init__(self, dtype, value):
        super().__init__(dtype, value)


class OneOfSink(_FilterSink):
    """GenericFilterSink<T, OneOfIF> (src/filter.rs:143-177): element equals any member of the set."""

    def __init__(self, dtype, values):
        super().__init__(dtype, list(values))


class VectorFilter:
    """Result of VectorReader.filter_iter(sink) (src/filter.rs:189-231).  The GPU pass runs once, on
    first use; the masks it produced are then iterated."""

    def __init__(self, reader, sink):
        if sink.etype != reader.etype:
            raise N.CodingError(N.WRONG_VECTOR_TYPE, sink.etype)
        self._reader, self._sink = reader, sink
        self._masks = None
        self._count = None

    def _run(self):
        if self._masks is None:
            dev = self._reader._device_batch()
            total, status, masks = dev.filter_count(self._sink.pred, return_status=True, return_masks=True)
            m = masks.reshape(-1, 32)
            if status[0] != N.OK:
                # A section that fails to decode ends the iteration silently (src/filter.rs:226); a
                # section header that fails to parse is a panic in the reference (src/filter.rs:221).
                valid, parse_failed = dev.valid_sections()
                if parse_failed[0]:
                    raise N.CodingError(int(status[0]))
                m = m[:int(valid[0])]
            self._masks = m
            self._count = total
            self._status = int(status[0])

    def __iter__(self):
        self._run()
        return iter(self._masks)

    def count_hits(self):
        self._run()
        return self._count


class MultiVectorFilter:
    """MultiVectorFilter::new(vec![filters]) (src/filter.rs:239-286): iterating it yields the AND of the
    filters' masks section by section; ends with the first filter, or with a later one that is
    consulted (running mask non-zero) and has no section left."""

    def __init__(self, vect_filters):
        if not vect_filters:
            raise ValueError("Cannot pass in empty filters to MultiVectorFilter")  # panics in the reference
        self._filters = list(vect_filters)
        self._masks = None

    def _run(self):
        if self._masks is None:
            from .batch import multi_filter_count
            cols = [f._reader._device_batch().filter_masks(f._sink.pred) for f in self._filters]
            total, masks, row_sections = multi_filter_count(cols, return_masks=True)
            self._count = total
            self._masks = masks.reshape(-1, 32)[:int(row_sections[0])]

    def __iter__(self):
        self._run()
        return iter(self._masks)

    def count_hits(self):
        self._run()
        return self._count


def count_hits(filter_iter):
    """filter::count_hits (src/filter.rs:296-299)."""
    if isinstance(filter_iter, (VectorFilter, MultiVectorFilter)):
        return filter_iter.count_hits()
    return int(sum(int(np.unpackbits(np.asarray(m, dtype=np.uint8)).sum()) for m in filter_iter))


def match_positions(filter_iter):
    """Element positions of the matches (src/filter.rs:302-321), as true positions: the reference's
    loop does not advance over all-zero 32-bit mask words, which under-reports later positions."""
    pos = []
    base = 0
    for m in filter_iter:
        bits = np.unpackbits(np.asarray(m, dtype=np.uint8), bitorder="little")
        pos.extend((np.nonzero(bits)[0] + base).tolist())
        base += 256
    return pos
