"""Cost of the reference tail rule (cvec_set_tail_rule): filter and decode passes over the headline shard under both
rules, ms per pass and the number of elements / matches that differ."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import compressed_vec_amd as cv
from compressed_vec_amd import synth

nvec = 131072
assert cv.lib().cvec_init(0) == 0
vals = torch.cat([synth.batch_values_torch("f32_xor", 16384, 2048, v_first=c0)[0] for c0 in range(0, nvec, 16384)])
eo = torch.arange(nvec + 1, device="cuda", dtype=torch.int64) * 2048
b = cv.encode_device("f32", vals, eo)
pred = cv.make_pred(cv.F32, float(vals[1].item()))
res = b.result_buffer()
out = torch.empty(b.n_sections * 256, dtype=torch.float32, device="cuda")


def timed(fn, iters=50):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


keep = {}
for rule in ("format", "reference"):
    cv.set_tail_rule(rule)
    tf = timed(lambda: b.filter_count_async(pred, result=res))
    cnt = int(res[0].item())
    td = timed(lambda: b.decode_async(out), 20)
    keep[rule] = (cnt, out.clone())
    print("%-9s filter %.4f ms (count %d)   decode %.4f ms" % (rule, tf, cnt, td))
cv.set_tail_rule("format")
d = (keep["format"][1].view(torch.int32) != keep["reference"][1].view(torch.int32))
print("elements that differ: %d of %d, in %d vectors; count difference %d" %
      (int(d.sum().item()), d.numel(), int(d.view(nvec, 2048).any(dim=1).sum().item()), keep["reference"][0] - keep["format"][0]))
