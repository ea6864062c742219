"""Kernel time of k_sections for every element type x mode on 1 GiB batches of 10000-element vectors."""
import ctypes as C
import sys
sys.path.insert(0, "/root/repo")
import torch
import compressed_vec_amd as cv
from compressed_vec_amd import synth

lib = cv.lib()


def kernel_ms(fn, iters=30):
    for _ in range(40):
        fn()
    torch.cuda.synchronize()
    lib.cvec_timing_begin(iters)
    for _ in range(iters):
        fn()
    ms = [C.c_float(0) for _ in range(3)]
    n = C.c_int32(0)
    lib.cvec_timing_end(C.byref(ms[0]), C.byref(ms[1]), C.byref(ms[2]), C.byref(n))
    return ms[1].value / n.value, ms[0].value / n.value


for name, tname, kind, nvec, p1, plist, addc in (("u32 dense", "u32", "u32_dense", 26214, 3, [3, 5, 7, 9], 7),
                                                 ("f32 xor", "f32", "f32_xor", 26214, 3.0, [3.0, 1.0, 7.5, 2.0], 2.5),
                                                 ("u64 delta", "u64", "u64_delta", 13107, 100003, [100003, 100005, 1, 2], 5)):
    vals = torch.cat([synth.batch_values_torch(kind, min(4096, nvec - c), 10000, v_first=c)[0] for c in range(0, nvec, 4096)])
    eo = torch.arange(nvec + 1, device="cuda", dtype=torch.int64) * 10000
    b = cv.encode_device(tname, vals, eo)
    raw = vals.numel() * vals.element_size()
    out = torch.empty(b.n_sections * 256, dtype=vals.dtype, device="cuda")
    masks = torch.zeros(b.n_sections * 32, dtype=torch.uint8, device="cuda")
    for mode, fn in (("count ==", lambda: b.filter_count_async(p1)),
                     ("count one-of-4", lambda: b.filter_count_async(plist)),
                     ("masks ==", lambda: b.filter_count_async(p1, masks=masks)),
                     ("decode", lambda: b.decode_async(out)),
                     ("decode + add", lambda: b.decode_async(out, add_const=addc))):
        t, ti = kernel_ms(fn)
        print("%-10s %-15s k_sections %.3f ms  %5.0f GB/s   (k_index %.3f ms)" % (name, mode, t, raw / t / 1e6, ti), flush=True)
    del vals, out, masks, b
