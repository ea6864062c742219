"""Does H2D overlap D2H on this box?  One GiB each way, alone and together, page-locked host memory."""
import time
import torch
n = 1 << 30
h_in = torch.empty(n, dtype=torch.uint8).pin_memory()
h_out = torch.empty(n, dtype=torch.uint8).pin_memory()
d_a = torch.empty(n, dtype=torch.uint8, device="cuda")
d_b = torch.empty(n, dtype=torch.uint8, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def run(h2d, d2h, chunks=1):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    c = n // chunks
    for i in range(chunks):
        if h2d:
            with torch.cuda.stream(s1):
                d_a[i * c:(i + 1) * c].copy_(h_in[i * c:(i + 1) * c], non_blocking=True)
        if d2h:
            with torch.cuda.stream(s2):
                h_out[i * c:(i + 1) * c].copy_(d_b[i * c:(i + 1) * c], non_blocking=True)
    torch.cuda.synchronize()
    return time.perf_counter() - t0


for name, a, b, ch in (("h2d", 1, 0, 1), ("d2h", 0, 1, 1), ("both", 1, 1, 1), ("both, 32 chunks", 1, 1, 32)):
    run(a, b, ch)
    t = min(run(a, b, ch) for _ in range(3))
    print("%-16s %.2f ms  %.1f GB/s per direction" % (name, t * 1e3, n / t / 1e9), flush=True)
