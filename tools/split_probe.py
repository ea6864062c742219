"""Does the header walk of one half of a batch hide behind the section kernel of the other half?  One filter step over
the headline shard as (a) one call, (b) P parts on two streams: the index pass of part i+1 is enqueued on the second
stream behind the index pass of part i, the section + finalize passes follow on the first stream.  Steps are kept
apart (the first stream waits for the second at the end of every step).  python tools/split_probe.py [parts]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import compressed_vec_amd as cv
from compressed_vec_amd import synth

parts = int(sys.argv[1]) if len(sys.argv) > 1 else 2
nvec = 131072
assert cv.lib().cvec_init(0) == 0
vals = torch.cat([synth.batch_values_torch("f32_xor", 16384, 2048, v_first=c0)[0] for c0 in range(0, nvec, 16384)])
eo = torch.arange(nvec + 1, device="cuda", dtype=torch.int64) * 2048
b = cv.encode_device("f32", vals, eo)
pred = cv.make_pred(cv.F32, 3.0)
res = b.result_buffer()


def timed(fn, iters=200):
    for _ in range(100):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


t_full = timed(lambda: b.filter_count_async(pred, result=res))
print("one call: %.4f ms" % (t_full * 1e3))
want = int(res[0].item())

cuts = [nvec * i // parts for i in range(parts + 1)]
sb_all = b.d_sec_base
views = []
for i in range(parts):
    a, e = cuts[i], cuts[i + 1]
    s0 = int(sb_all[a].item())
    s1 = int(sb_all[e].item())
    v = cv.DeviceBatch(b.etype, b.d_bytes, b.d_byte_off[a:e + 1].contiguous(), (sb_all[a:e + 1] - s0).contiguous(), e - a,
                       s1 - s0, b.n_bytes)
    v.workspace()
    views.append(v)
results = [v.result_buffer() for v in views]
s_main = torch.cuda.current_stream()
s_idx = torch.cuda.Stream()
ev_idx = [torch.cuda.Event() for _ in range(parts)]
ev_done = torch.cuda.Event()


def step():
    # index passes: part 0 on the main stream, the others on the side stream, each behind the one before it
    s_idx.wait_stream(s_main)
    views[0].build_index()
    ev_idx[0].record(s_main)
    with torch.cuda.stream(s_idx):
        for i in range(1, parts):
            if i == 1:
                s_idx.wait_event(ev_idx[0])
            views[i].build_index()
            ev_idx[i].record(s_idx)
    for i in range(parts):
        if i:
            s_main.wait_event(ev_idx[i])
        views[i].filter_count_async(pred, result=results[i])


t_split = timed(step)
got = sum(int(r[0].item()) for r in results)
print("%d parts on two streams: %.4f ms (count %d, want %d)" % (parts, t_split * 1e3, got, want))
