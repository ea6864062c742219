"""Run under rocprofv3 --kernel-trace: a three-column MultiVectorFilter and match_positions over
0.5 GiB u32 columns."""
import sys
sys.path.insert(0, "/root/repo")
import torch
import compressed_vec_amd as cv
from compressed_vec_amd import synth

nvec, length = 13107, 10000
cols = []
for c in range(3):
    vals = torch.cat([synth.batch_values_torch("u32_dense", min(4096, nvec - s), length, v_first=s + 1000 * c)[0]
                      for s in range(0, nvec, 4096)])
    eo = torch.arange(nvec + 1, device="cuda", dtype=torch.int64) * length
    cols.append(cv.encode_device("u32", vals, eo))
for _ in range(3):
    total = cv.multi_filter(cols, [3, 6, 10])
    first = cols[0].filter_masks(3)
    pos, off = first.match_positions()
torch.cuda.synchronize()
print("matches of all three:", total, " positions of column 0:", pos.size)
