"""Two f32 columns of the benchmark shape; the second column behind the gate of a first column that
matches in a given fraction of the sections."""
import sys
import time
sys.path.insert(0, "/root/repo")
import torch
import compressed_vec_amd as cv
from compressed_vec_amd import synth

nvec, length = 65536, 2048
vals = torch.cat([synth.batch_values_torch("f32_xor", 16384, length, v_first=c)[0] for c in range(0, nvec, 16384)])
eo = torch.arange(nvec + 1, device="cuda", dtype=torch.int64) * length
col = cv.encode_device("f32", vals, eo)
raw = nvec * length * 4
masks = torch.zeros(col.n_sections * 32, dtype=torch.uint8, device="cuda")


import ctypes as C
lib = cv.lib()


def timed(fn, iters=10):
    """Sum of the three kernels' event times per call (the host side of a call from Python costs more
    than the kernels once most sections are gated off)."""
    fn()
    torch.cuda.synchronize()
    assert lib.cvec_timing_begin(iters) == 0
    for _ in range(iters):
        fn()
    ms = [C.c_float(0), C.c_float(0), C.c_float(0)]
    nrec = C.c_int32(0)
    assert lib.cvec_timing_end(C.byref(ms[0]), C.byref(ms[1]), C.byref(ms[2]), C.byref(nrec)) == 0
    return sum(m.value for m in ms) / nrec.value * 1e-3


if len(sys.argv) > 1 and sys.argv[1] == "indexed":
    col.build_index()
    print("resident section index: the gate is read by the section kernel", flush=True)
t_plain = timed(lambda: col.filter_count_async(3.0, masks=masks))
print("masks pass, no gate: %.3f ms  %.0f GB/s" % (t_plain * 1e3, raw / t_plain / 1e9), flush=True)
g = torch.Generator(device="cuda").manual_seed(1)
for frac in (1.0, 0.5, 0.1, 0.01, 0.0):
    for runs in (1, 64):   # open sections scattered one by one / in runs of 64 consecutive sections
        nrun = col.n_sections // runs
        open_ = (torch.rand(nrun, device="cuda", generator=g) < frac).repeat_interleave(runs)
        gate = (open_[:, None].to(torch.uint8) * torch.ones(32, dtype=torch.uint8, device="cuda")).reshape(-1).contiguous()
        t = timed(lambda: col.filter_count_async(3.0, masks=masks, gate=gate))
        print("gate open %5.1f %% (runs of %2d): %.3f ms  %.0f GB/s" % (100 * frac, runs, t * 1e3, raw / t / 1e9), flush=True)
