"""k_sections count time on f32 batches of growing section size (SplitMix data with more mantissa bits):
run once under CVEC_REGION_SHAPE=1 (18 KiB regions) and once under =2 (12 KiB) to see where the small
shape stops paying (it wins up to ~720 B per section and loses from ~840 B, where only 14 sections fit)."""
import sys, ctypes as C
sys.path.insert(0, "/root/repo")
import torch
import compressed_vec_amd as cv
from compressed_vec_amd import synth
lib = cv.lib()
def kernel_ms(fn, iters=30):
    for _ in range(40): fn()
    torch.cuda.synchronize()
    lib.cvec_timing_begin(iters)
    for _ in range(iters): fn()
    ms = [C.c_float(0) for _ in range(3)]; n = C.c_int32(0)
    lib.cvec_timing_end(C.byref(ms[0]), C.byref(ms[1]), C.byref(ms[2]), C.byref(n))
    return ms[1].value / n.value
for mb in (12, 16, 18, 20, 22):
    vals = torch.cat([synth.splitmix_f32_torch(16384, 2048, v_first=c, mantissa_bits=mb)[0] for c in range(0, 131072, 16384)])
    eo = torch.arange(131072 + 1, device="cuda", dtype=torch.int64) * 2048
    b = cv.encode_device("f32", vals, eo)
    p = float(vals[12345].item())
    t = kernel_ms(lambda: b.filter_count_async(p))
    print("mantissa bits %d: %.0f B/section  k_sections %.3f ms" % (mb, b.n_bytes / b.n_sections, t), flush=True)
    del vals, b
