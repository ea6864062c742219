import sys, time
sys.path.insert(0, "/root/repo")
import torch
import compressed_vec_amd as cv
from compressed_vec_amd import synth
cv.lib().cvec_init(0)
for nvec in (8192, 16384, 32768, 65536, 131072):
    vals = torch.cat([synth.batch_values_torch("f32_xor", min(nvec, 16384), 2048)[0] for _ in range(max(1, nvec // 16384))])
    eo = torch.arange(nvec + 1, device="cuda", dtype=torch.int64) * 2048
    bufs = cv.EncodeBuffers("f32", nvec, vals.numel())
    for _ in range(5):
        cv.encode_device_async(bufs, vals, eo)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    it = 30
    for _ in range(it):
        cv.encode_device_async(bufs, vals, eo)
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / it
    mib = vals.numel() * 4 / 2**20
    print("%6d vectors %5.0f MiB: %.4f ms per call, %.4f ms per GiB, %.0f GB/s" % (nvec, mib, t * 1e3, t * 1e3 * 1024 / mib, vals.numel() * 4 / t / 1e9), flush=True)
