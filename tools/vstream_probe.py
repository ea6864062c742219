"""Phase times of the vector-streaming kernel (probe build: make -C compressed-vec_amd/csrc probes, then
CVEC_LIB_PATH=.../libcvec_hip_probes.so python tools/vstream_probe.py [data] [nvec]).  Prints the average
kernel time and the share of wave cycles per phase of the main loop."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import compressed_vec_amd as cv
from compressed_vec_amd import synth

data = sys.argv[1] if len(sys.argv) > 1 else "sinusoid"
nvec = int(sys.argv[2]) if len(sys.argv) > 2 else 131072
assert cv.lib().cvec_init(0) == 0
if os.environ.get("PROBE_PATH"):
    assert cv.lib().cvec_set_read_path(int(os.environ["PROBE_PATH"])) == 0    # 1 = section-slot kernels, 2 = vector-streaming
chunks = []
if data == "sparse":          # the reference's very sparse u32 bench vector (benches/bench.rs:160-170), 26214 of them
    nvec = int(sys.argv[2]) if len(sys.argv) > 2 else 26214
    for c0 in range(0, nvec, 4096):
        chunks.append(synth.batch_values_torch("u32_dense", min(4096, nvec - c0), 10000, v_first=c0)[0])
    vals = torch.cat(chunks)
    keep = torch.zeros(10000, dtype=torch.bool, device="cuda")
    keep[2475:2525] = True
    keep[7475:7525] = True
    vals = torch.where(keep.repeat(nvec), vals, torch.zeros_like(vals))
    eo = torch.arange(nvec + 1, device="cuda", dtype=torch.int64) * 10000
    b = cv.encode_device("u32", vals, eo)
    pred = cv.make_pred(cv.U32, 15)
else:
    for c0 in range(0, nvec, 16384):
        n = min(16384, nvec - c0)
        v = synth.batch_values_torch("f32_xor", n, 2048, v_first=c0)[0] if data == "sinusoid" else \
            synth.splitmix_f32_torch(n, 2048, v_first=c0)[0]
        chunks.append(v)
    vals = torch.cat(chunks)
    eo = torch.arange(nvec + 1, device="cuda", dtype=torch.int64) * 2048
    b = cv.encode_device("f32", vals, eo)
    pred = cv.make_pred(cv.F32, 3.0)
for _ in range(20):
    b.filter_count_async(pred)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    b.filter_count_async(pred)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 50
print("ablate=%s %s: %.4f ms per step, %.0f GB/s" % (os.environ.get("CVEC_ABLATE", "0"), data, dt * 1e3, vals.numel() * 4 / dt / 1e9))
ws = b.workspace()
off = ((b.n_sections + 1) * 8 + 255) // 256 * 256 + ((b.n_sections + 1) * 4 + 255) // 256 * 256 + ((b.n_vectors + 1) * 4 + 255) // 256 * 256
w = ws[off:off + 256].cpu().numpy().view(np.uint64)
ph = w[16:24].astype(np.float64)
names = ["wait+write", "parse", "topup", "issue", "loop", "plan", "carry", "fold"]
if ph.sum() > 0:
    print("  ".join("%s %.1f%%" % (n, 100 * x / ph.sum()) for n, x in zip(names, ph) if n != "-"))
    print("cycles per wave: %.0f" % (ph.sum() / 2048))
