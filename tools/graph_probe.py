#!/usr/bin/env python3
"""The headline step (index, sections, finalize) launched call by call and replayed from a captured HIP graph of 20
steps: prints the time per step of both.  The read entry points take their workspace from the caller and neither
allocate nor synchronise, so they can be captured as they are."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import compressed_vec_amd as cv
import profile_op
assert cv.lib().cvec_init(0) == 0
tname, vals, eo, pred, _ = profile_op.build("cfg5sin")
batch = cv.encode_device(tname, vals, eo)
del vals
res = batch.result_buffer()
p = cv.make_pred(batch.etype, pred)
for _ in range(300):
    batch.filter_count_async(p, result=res)
torch.cuda.synchronize()
def timed(fn, n):
    torch.cuda.synchronize(); t0 = time.perf_counter(); fn(n); torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
def plain(n):
    for _ in range(n):
        batch.filter_count_async(p, result=res)
print("plain launches: %.4f ms per step" % min(timed(plain, 200) for _ in range(5)))
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(3):
        batch.filter_count_async(p, result=res)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        for _ in range(20):
            batch.filter_count_async(p, result=res)
def replay(n):
    for _ in range(n // 20):
        g.replay()
print("graph of 20 steps: %.4f ms per step" % min(timed(replay, 200) for _ in range(5)))
print("count", int(res.cpu()[0]))
