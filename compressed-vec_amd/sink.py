"""Sinks of the decode pipeline (reference src/sink.rs).  On the GPU the sink stack is not a chain of
calls per octet: it is a descriptor handed to the decode kernel, which fuses it into its epilogue.
XorSink (src/sink.rs:241-284) never appears here: as in the reference (src/section.rs:707) it is
implied by an Xor section."""
import numpy as np

from .batch import NP_DTYPE, etype_of


class VecSink:
    """Terminal sink that collects every decoded value, whole sections at a time, so up to 255
    trailing zeros follow the last element (src/sink.rs:148-182)."""

    def __init__(self, dtype):
        self.etype = etype_of(dtype)
        self.vec = np.zeros(0, dtype=NP_DTYPE[self.etype])

    def reset(self):
        self.vec = np.zeros(0, dtype=NP_DTYPE[self.etype])

    def _terminal(self):
        return self

    def _add_const(self):
        return None

    def _accept(self, vals):
        self.vec = np.concatenate([self.vec, vals])


class AddConstSink:
    """Adds a constant to every element before handing it to the inner sink (src/sink.rs:289-320):
    wrapping add for u32/u64, IEEE add for f32; a Null section forwards the constant itself."""

    def __init__(self, base, inner_sink):
        if not hasattr(inner_sink, "_accept"):
            raise TypeError("AddConstSink wraps a terminal sink (VecSink or Section256Sink)")
        self.base = base
        self.inner_sink = inner_sink
        self.etype = inner_sink.etype

    def reset(self):
        pass

    def _terminal(self):
        return self.inner_sink

    def _add_const(self):
        return self.base


class Section256Sink:
    """Holds the 256 values of one section (src/sink.rs:195-232): takes values until it is full and
    ignores what follows; reset() empties it for the next section.  Filled by SectionRef.decode(sink)
    (one section of VectorReader.sections()), or by VectorReader.decode_to_sink, which then leaves the
    vector's first section in it."""

    def __init__(self, dtype):
        self.etype = etype_of(dtype)
        self.values = np.zeros(256, dtype=NP_DTYPE[self.etype])
        self.i = 0

    def reset(self):
        self.i = 0

    def _terminal(self):
        return self

    def _add_const(self):
        return None

    def _accept(self, vals):
        take = min(256 - self.i, vals.size)
        self.values[self.i:self.i + take] = vals[:take]
        self.i += take
