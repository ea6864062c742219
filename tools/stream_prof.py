"""Run under rocprofv3 --kernel-trace: a few pack / unpack passes of the legacy delta stream codec."""
import sys
sys.path.insert(0, "/root/repo")
import torch
import compressed_vec_amd as cv
from compressed_vec_amd import stream as st
nstreams, length = 3277, 10000          # 250 MiB of u64
g = torch.Generator(device="cuda").manual_seed(5)
inc = torch.randint(0, 1000, (nstreams * length,), device="cuda", dtype=torch.int64, generator=g)
vals = torch.cumsum(inc.view(nstreams, length), dim=1).reshape(-1).contiguous()
eo = torch.arange(nstreams + 1, device="cuda", dtype=torch.int64) * length
total = nstreams * length
d_bytes, d_bo, _, d_res = st.pack_device("u64_delta", vals, eo, total)
nbytes = int(d_res[0].item())
out = torch.empty(total, dtype=torch.int64, device="cuda")
ws = torch.empty(cv.lib().cvec_stream_unpack_workspace_bytes(nstreams, total), dtype=torch.uint8, device="cuda")
for _ in range(4):
    st.pack_device("u64_delta", vals, eo, total)
    st.unpack_device("u64_delta", d_bytes, d_bo, nbytes, eo, total, out, ws)
torch.cuda.synchronize()
assert torch.equal(out, vals)
