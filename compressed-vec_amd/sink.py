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


class AddConstSink:
    """Adds a constant to every element before handing it to the inner sink (src/sink.rs:289-320):
    wrapping add for u32/u64, IEEE add for f32; a Null section forwards the constant itself."""

    def __init__(self, base, inner_sink):
        if not isinstance(inner_sink, VecSink):
            raise TypeError("AddConstSink wraps a VecSink")
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
    """Holds the 256 values of one section (src/sink.rs:195-232); filled by VectorReader.sections()."""

    def __init__(self, dtype):
        self.etype = etype_of(dtype)
        self.values = np.zeros(256, dtype=NP_DTYPE[self.etype])
