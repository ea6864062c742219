#!/usr/bin/env python3
"""What the device delivers for the traffic shapes of the decode pass, as plain copies: 1 GiB read + 1 GiB
written (f32 XOR decode: blob about as large as its output), 0.2 GiB read + 1 GiB written (dense u32) and
1 GiB written alone.  torch's copy_ / fill_ kernels; milliseconds per pass over 50 passes."""
import torch

def timed(fn, n=50):
    for _ in range(10):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n

G = 1 << 30
src = torch.empty(G // 4, dtype=torch.float32, device="cuda").normal_()
dst = torch.empty_like(src)
ms = timed(lambda: dst.copy_(src))
print("copy 1 GiB -> 1 GiB: %.4f ms, %.0f GB/s of traffic (%.0f GB/s of output)" % (ms, 2 * G / ms / 1e6, G / ms / 1e6))
small = src[: G // 20]
ms = timed(lambda: (dst[: G // 20].copy_(small), dst[G // 20:].fill_(1.0)))
print("read 0.2 GiB, write 1 GiB (copy + fill): %.4f ms, %.0f GB/s of output" % (ms, G / ms / 1e6))
ms = timed(lambda: dst.fill_(2.0))
print("write 1 GiB (fill): %.4f ms, %.0f GB/s" % (ms, G / ms / 1e6))
ms = timed(lambda: src.sum())
print("read 1 GiB (sum): %.4f ms, %.0f GB/s" % (ms, G / ms / 1e6))
