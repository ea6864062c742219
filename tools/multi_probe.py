"""Wall time of a three-column MultiVectorFilter (multi_filter: every column gated by the one before it)
over 0.5 GiB f32 columns whose first column matches in a given fraction of the sections; once through
the per-pass header walk, once over resident section indexes."""
import sys
import time
sys.path.insert(0, "/root/repo")
import torch
import compressed_vec_amd as cv

nvec, length = 65536, 2048
g = torch.Generator(device="cuda").manual_seed(3)


def column(p_hit):
    x = torch.randint(1, 6, (nvec, length), device="cuda", generator=g).to(torch.float32)
    hit = torch.rand(nvec, length // 256, device="cuda", generator=g) < p_hit       # sections that hold a 9 at all
    plant = hit.repeat_interleave(256, dim=1) & (torch.rand(nvec, length, device="cuda", generator=g) < 0.3)
    x[plant] = 9.0
    return x.reshape(-1)


eo = torch.arange(nvec + 1, device="cuda", dtype=torch.int64) * length
raw = 3 * nvec * length * 4
for p_hit in (0.5, 0.1, 0.01):
    cols = [cv.encode_device("f32", column(p), eo) for p in (p_hit, 0.5, 0.5)]
    for indexed in (False, True):
        if indexed:
            for c in cols:
                c.build_index()
        want = None
        for rep in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                total = cv.multi_filter(cols, [9.0, 9.0, 3.0])
            torch.cuda.synchronize()
            t = (time.perf_counter() - t0) / 5
        print("first column open %4.1f %%  %s: %.3f ms per query  (%.0f GB/s of the three columns, %d matches)"
              % (100 * p_hit, "resident index" if indexed else "header walk   ", t * 1e3, raw / t / 1e9, total), flush=True)
