"""Host-inclusive rates of the cvec_*_batch entry points (page-locked arrays) for a few chunk sizes."""
import os
import sys
import time
sys.path.insert(0, "/root/repo")
import numpy as np
import torch
import compressed_vec_amd as cv
from compressed_vec_amd import hostpath, synth

nvec, length = 131072, 2048
vals = torch.cat([synth.batch_values_torch("f32_xor", 16384, length, v_first=c)[0] for c in range(0, nvec, 16384)])
eo = torch.arange(nvec + 1, device="cuda", dtype=torch.int64) * length
dev = cv.encode_device("f32", vals, eo)
raw = nvec * length * 4
host = cv.EncodedBatch(cv.F32, hostpath.pin(dev.d_bytes[:dev.n_bytes].cpu().numpy()),
                       hostpath.pin(dev.d_byte_off.cpu().numpy().view(np.uint64)))
h_out = hostpath.pin(np.zeros(dev.n_sections * 256, dtype=np.float32))
h_vals = hostpath.pin(vals.cpu().numpy())
h_eo = eo.cpu().numpy().view(np.uint64)
h_blob = hostpath.pin(np.zeros(cv.lib().cvec_max_encoded_bytes(cv.F32, nvec, nvec * length), dtype=np.uint8))


def wall(fn, iters=3):
    fn()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    return (time.perf_counter() - t0) / iters


for mib in (8, 16, 32, 64, 128, 2048):
    os.environ["CVEC_PIPE_CHUNK_BYTES"] = str(mib << 20)
    tf = wall(lambda: hostpath.filter_count_host(host, 3.0))
    td = wall(lambda: hostpath.decode_host(host, add_const=2.5, out=h_out))
    te = wall(lambda: hostpath.encode_host("f32", h_vals, h_eo, out=h_blob))
    print("chunk %4d MiB: filter %.1f  decode %.1f  encode %.1f GB/s" % (mib, raw / tf / 1e9, raw / td / 1e9, raw / te / 1e9),
          flush=True)
