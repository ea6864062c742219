"""Decode kernel time of k_sections on f32 batches of growing section size (SplitMix data with 4 to 23 mantissa
bits, 131072 vectors of 2048), for comparing builds of the decode loop (narrow / wide stagings, 8- or 16-byte stores,
store policy):  CVEC_LIB_PATH=compressed-vec_amd/lib/<variant>.so python3 tools/decode_shape_probe.py"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import compressed_vec_amd as cv  # noqa: E402
from compressed_vec_amd import synth  # noqa: E402

lib = cv.lib()


def kernel_ms(fn, iters=20):
    for _ in range(25):
        fn()
    torch.cuda.synchronize()
    lib.cvec_timing_begin(iters)
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    ms = [C.c_float(0) for _ in range(3)]
    n = C.c_int32(0)
    lib.cvec_timing_end(C.byref(ms[0]), C.byref(ms[1]), C.byref(ms[2]), C.byref(n))
    return ms[1].value / n.value


MODE = sys.argv[1] if len(sys.argv) > 1 else "f32"
for bits in (((4, 6, 8, 10, 12, 16, 20, 24, 28, 32) if MODE == "u32" else (4, 8, 12, 16, 24, 32, 40, 48, 56, 63)) if MODE != "f32" else ()):
    # dense integers of `bits` random bits: 26214 vectors of 10000 (u32) or 13107 (u64)
    nvec = 26214 if MODE == "u32" else 13107
    dt = torch.int32 if MODE == "u32" else torch.int64
    g = torch.Generator(device="cuda").manual_seed(bits)
    vals = (torch.randint(0, 1 << min(bits, 31), (nvec * 10000,), generator=g, device="cuda", dtype=torch.int64) << (max(bits - 31, 0) if MODE == "u64" else 0)).to(dt)
    eo = torch.arange(nvec + 1, device="cuda", dtype=torch.int64) * 10000
    b = cv.encode_device(MODE, vals, eo)
    out = torch.empty(b.n_sections * 256, dtype=dt, device="cuda")
    t = kernel_ms(lambda: b.decode_async(out))
    ok = torch.equal(out.view(nvec, -1)[:, :10000].contiguous().view(-1), vals)
    print("%s of %2d bits: %4.0f B/section  decode kernel %.4f ms per %.2f GiB  exact %s" %
          (MODE, bits, b.n_bytes / b.n_sections, t, out.numel() * out.element_size() / 2**30, ok), flush=True)
    del vals, b, out
    torch.cuda.empty_cache()
for mb in ((4, 8, 12, 14, 16, 18, 20, 23) if MODE == "f32" else ()):
    vals = torch.cat([synth.splitmix_f32_torch(16384, 2048, v_first=c, mantissa_bits=mb)[0] for c in range(0, 131072, 16384)])
    eo = torch.arange(131072 + 1, device="cuda", dtype=torch.int64) * 2048
    b = cv.encode_device("f32", vals, eo)
    out = torch.empty(b.n_sections * 256, dtype=torch.float32, device="cuda")
    t = kernel_ms(lambda: b.decode_async(out))
    ok = torch.equal(out.view(torch.int32), vals.view(torch.int32))
    print("mantissa bits %2d: %4.0f B/section  decode kernel %.4f ms  exact %s" % (mb, b.n_bytes / b.n_sections, t, ok), flush=True)
    del vals, b, out
    torch.cuda.empty_cache()
