"""Synthetic inputs of the reference's `cargo bench vector` workloads (benches/bench.rs:45-56,
:131-170), generated in float32 like the reference does.  Vector v of a batch is phase-shifted by
37*v so that the vectors of a batch differ."""
import math

import numpy as np

VECTOR_LENGTH = 10000          # benches/bench.rs:131
PHASE_STEP = 37


def sinewave_varnonzeros_u32(fract_nonzeroes, length, phase=0):
    """benches/bench.rs:45-56: a sinusoid of period 20 scaled so that roughly `fract_nonzeroes` of the
    values are positive, truncated to u32 (negatives become 0)."""
    amp = np.float32(7.0) / np.float32(fract_nonzeroes)
    dist = np.float32(15.0) - amp
    i = (np.arange(length, dtype=np.int64) + phase).astype(np.float32)
    s = np.sin(i * np.float32(math.pi) / np.float32(10.0), dtype=np.float32)
    v = s * amp + dist
    return np.where(v >= 0, v, 0).astype(np.uint32)


def batch_values(kind, n_vectors, length):
    """Values of a whole batch as one flat array + elem_off.  kind: 'u32_dense' (bench.rs:133-137),
    'f32_xor' (bench.rs:146-151: 60 % density, /1.4), 'u64_delta' (bench.rs:153-158: +100000)."""
    out_dtype = {"u32_dense": np.uint32, "f32_xor": np.float32, "u64_delta": np.uint64}[kind]
    vals = np.empty(n_vectors * length, dtype=out_dtype)
    for v in range(n_vectors):
        if kind == "u32_dense":
            x = sinewave_varnonzeros_u32(1.0, length, PHASE_STEP * v)
        elif kind == "f32_xor":
            x = sinewave_varnonzeros_u32(0.6, length, PHASE_STEP * v).astype(np.float32) / np.float32(1.4)
        else:
            x = sinewave_varnonzeros_u32(1.0, length, PHASE_STEP * v).astype(np.uint64) + np.uint64(100000)
        vals[v * length:(v + 1) * length] = x
    elem_off = (np.arange(n_vectors + 1, dtype=np.uint64) * np.uint64(length))
    return vals, elem_off


def batch_values_torch(kind, n_vectors, length, device="cuda", v_first=0):
    """Same generator evaluated on the device with torch (float32 math), for batches too large to
    build on the host quickly; vector i of the result has the phase of global vector v_first + i.
    Returns (values tensor, elem_off tensor int64)."""
    import torch
    f = 0.6 if kind == "f32_xor" else 1.0
    amp = np.float32(7.0) / np.float32(f)
    dist = np.float32(15.0) - amp
    idx = torch.arange(length, device=device, dtype=torch.int64)[None, :] + \
        PHASE_STEP * (torch.arange(n_vectors, device=device, dtype=torch.int64)[:, None] + v_first)
    s = torch.sin(idx.to(torch.float32) * float(np.float32(math.pi)) / 10.0)
    v = s * float(amp) + float(dist)
    u = torch.where(v >= 0, v, torch.zeros_like(v)).to(torch.int64)     # truncation toward zero
    if kind == "u32_dense":
        vals = u.to(torch.int32)
    elif kind == "f32_xor":
        vals = u.to(torch.float32) / float(np.float32(1.4))
    else:
        vals = u + 100000
    elem_off = torch.arange(n_vectors + 1, device=device, dtype=torch.int64) * length
    return vals.reshape(-1).contiguous(), elem_off


def splitmix_f32_torch(n_vectors, length, device="cuda", v_first=0, mantissa_bits=12):
    """Aperiodic f32 test data: element i of vector v is a float in [1, 2) whose top `mantissa_bits`
    mantissa bits come from SplitMix64 seeded with 0xC0FFEE + v (the other mantissa bits are zero), so
    neighbouring values share sign and exponent and differ in a fixed number of bits.
    Returns (values tensor, elem_off tensor int64)."""
    import torch

    def srl(x, k):  # logical shift right of int64
        return (x >> k) & ((1 << (64 - k)) - 1)

    def wrap(c):  # a 64-bit constant as a wrapping int64
        return c - (1 << 64) if c >= (1 << 63) else c

    seed = torch.arange(n_vectors, device=device, dtype=torch.int64)[:, None] + (0xC0FFEE + v_first)
    idx = torch.arange(1, length + 1, device=device, dtype=torch.int64)[None, :]
    z = seed + idx * wrap(0x9E3779B97F4A7C15)
    z = (z ^ srl(z, 30)) * wrap(0xBF58476D1CE4E5B9)
    z = (z ^ srl(z, 27)) * wrap(0x94D049BB133111EB)
    z = z ^ srl(z, 31)
    frac = (z & ((1 << mantissa_bits) - 1)) << (23 - mantissa_bits)
    bits = (frac | 0x3F800000).to(torch.int32)
    elem_off = torch.arange(n_vectors + 1, device=device, dtype=torch.int64) * length
    return bits.view(torch.float32).reshape(-1).contiguous(), elem_off


def splitmix_f32(n_vectors, length, v_first=0, mantissa_bits=12):
    """splitmix_f32_torch on the host with numpy (same bits): (values float32, elem_off uint64)."""
    with np.errstate(over="ignore"):
        seed = (np.arange(n_vectors, dtype=np.uint64)[:, None] + np.uint64(0xC0FFEE + v_first))
        idx = np.arange(1, length + 1, dtype=np.uint64)[None, :]
        z = seed + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    frac = (z & np.uint64((1 << mantissa_bits) - 1)) << np.uint64(23 - mantissa_bits)
    bits = (frac | np.uint64(0x3F800000)).astype(np.uint32)
    elem_off = np.arange(n_vectors + 1, dtype=np.uint64) * np.uint64(length)
    return bits.view(np.float32).reshape(-1).copy(), elem_off


def sparse_u32_vector_parts():
    """The pieces of the reference's very sparse u32 vector (benches/bench.rs:160-170): explicit null
    runs and two runs of 50 sinusoid values; finish(10000).  Returns the list of ('nulls', n) /
    ('values', array) steps an appender is driven with, and the row count for finish()."""
    nonzeroes = sinewave_varnonzeros_u32(1.0, 50)
    return [("nulls", 2475), ("values", nonzeroes), ("nulls", 4950), ("values", nonzeroes)], VECTOR_LENGTH
