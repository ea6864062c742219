"""Does replaying the three kernels of a filter step as one HIP graph shorten the step?"""
import sys
import time
sys.path.insert(0, "/root/repo")
import torch
import compressed_vec_amd as cv
from compressed_vec_amd import synth

nvec, length = 131072, 2048
vals = torch.cat([synth.batch_values_torch("f32_xor", 16384, length, v_first=c)[0] for c in range(0, nvec, 16384)])
eo = torch.arange(nvec + 1, device="cuda", dtype=torch.int64) * length
b = cv.encode_device("f32", vals, eo)
raw = nvec * length * 4
pred = cv.make_pred(cv.F32, 3.0)
res = b.result_buffer()


def timed(fn, iters=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


t = timed(lambda: b.filter_count_async(pred, result=res))
print("direct launches: %.4f ms  %.0f GB/s" % (t * 1e3, raw / t / 1e9), flush=True)
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    b.filter_count_async(pred, result=res)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        b.filter_count_async(pred, result=res)
torch.cuda.synchronize()
t = timed(g.replay)
print("graph replay:    %.4f ms  %.0f GB/s  count %d" % (t * 1e3, raw / t / 1e9, int(res[0].item())), flush=True)
