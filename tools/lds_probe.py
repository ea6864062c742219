"""Run under rocprofv3 --pmc: count-mode filter passes over a dense u32 batch (sections 195 B apart)."""
import sys
sys.path.insert(0, "/root/repo")
import torch
import compressed_vec_amd as cv
from compressed_vec_amd import synth
kind, tname, p = (sys.argv[1], sys.argv[2], float(sys.argv[3]) if sys.argv[2] == "f32" else int(sys.argv[3])) if len(sys.argv) > 3 else ("u32_dense", "u32", 3)
nvec = 26214
vals = torch.cat([synth.batch_values_torch(kind, min(4096, nvec - c), 10000, v_first=c)[0] for c in range(0, nvec, 4096)])
eo = torch.arange(nvec + 1, device="cuda", dtype=torch.int64) * 10000
b = cv.encode_device(tname, vals, eo)
for _ in range(10):
    b.filter_count_async(p)
torch.cuda.synchronize()
