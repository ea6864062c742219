"""Filter rate of the benchmark batch for the predicate shapes that take different kernels."""
import ctypes as C
import sys
sys.path.insert(0, "/root/repo")
import torch
import compressed_vec_amd as cv
from compressed_vec_amd import synth

nvec, length = 131072, 2048
vals = torch.cat([synth.batch_values_torch("f32_xor", 16384, length, v_first=c)[0] for c in range(0, nvec, 16384)])
eo = torch.arange(nvec + 1, device="cuda", dtype=torch.int64) * length
b = cv.encode_device("f32", vals, eo)
raw = nvec * length * 4
lib = cv.lib()
masks = torch.zeros(b.n_sections * 32, dtype=torch.uint8, device="cuda")


def kernel_ms(fn, iters=20):
    fn()
    torch.cuda.synchronize()
    lib.cvec_timing_begin(iters)
    for _ in range(iters):
        fn()
    ms = [C.c_float(0) for _ in range(3)]
    n = C.c_int32(0)
    lib.cvec_timing_end(C.byref(ms[0]), C.byref(ms[1]), C.byref(ms[2]), C.byref(n))
    return ms[1].value / n.value


for name, fn in (("== 3.0 (bit-pattern count)", lambda: b.filter_count_async(3.0)),
                 ("== 0.0 (IEEE compare count)", lambda: b.filter_count_async(0.0)),
                 ("one of 4 values", lambda: b.filter_count_async([3.0, 1.0, 7.5, 0.0])),
                 ("one of 16 values", lambda: b.filter_count_async([float(i) for i in range(16)])),
                 ("== 3.0 with 256-bit masks", lambda: b.filter_count_async(3.0, masks=masks)),
                 ("one of 4 with masks", lambda: b.filter_count_async([3.0, 1.0, 7.5, 0.0], masks=masks)),
                 ("== 3.0 again", lambda: b.filter_count_async(3.0)),
                 ("== 0.0 again", lambda: b.filter_count_async(0.0))):
    t = kernel_ms(fn)
    print("%-30s k_sections %.3f ms  %.0f GB/s" % (name, t, raw / t / 1e6), flush=True)
