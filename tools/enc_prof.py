import os, sys
sys.path.insert(0, "/root/repo")
import torch, compressed_vec_amd as cv
from compressed_vec_amd import synth
n=32768
vals=torch.cat([synth.batch_values_torch("f32_xor", 4096, 2048, v_first=c)[0] for c in range(0,n,4096)])
eo=torch.arange(n+1,device="cuda",dtype=torch.int64)*2048
for _ in range(5): b=cv.encode_device("f32", vals, eo)
torch.cuda.synchronize()
