"""compressed_vec on MI355X: NibblePack / XOR / delta encodings as hand-written gfx950 kernels behind
the C ABI of include/cvec.h, with the reference crate's API names on top (see src/lib.rs:100-102 of
the reference for the list this mirrors)."""
from ._native import (CodingError, F32, U32, U64, LIB_PATH, lib)
from ._native import set_tail_rule
from .batch import (DeviceBatch, EncodeBuffers, EncodedBatch, FilteredColumn, encode_batch, encode_device,
                    encode_device_async, etype_of, make_pred,
                    multi_filter, multi_filter_count)
from .filter import EqualsSink, MultiVectorFilter, OneOfSink, VectorFilter, count_hits, match_positions
from . import stream
from . import hostpath
from . import histogram
from .hostpath import decode_host, encode_host, filter_count_host
from .series import SeriesAppender
from .sink import AddConstSink, Section256Sink, VecSink
from .vector import (VectorAppender, VectorF32XorAppender, VectorReader, VectorU32Appender, VectorU64Appender)

__all__ = ["SeriesAppender", "hostpath", "stream", "histogram", "decode_host", "encode_host", "filter_count_host", "MultiVectorFilter", "FilteredColumn", "multi_filter", "multi_filter_count", "CodingError", "U32", "U64", "F32", "DeviceBatch", "EncodedBatch", "encode_batch", "encode_device",
           "EqualsSink", "OneOfSink", "VectorFilter", "count_hits", "match_positions", "AddConstSink",
           "Section256Sink", "VecSink", "VectorAppender", "VectorF32XorAppender", "VectorReader",
           "VectorU32Appender", "VectorU64Appender", "etype_of", "make_pred", "lib", "LIB_PATH"]
