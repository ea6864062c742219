import sys, time, ctypes as C
sys.path.insert(0, "/root/repo")
import torch, compressed_vec_amd as cv
from compressed_vec_amd import synth
n=131072
vals=torch.cat([synth.batch_values_torch("f32_xor", 16384, 2048, v_first=c)[0] for c in range(0,n,16384)])
eo=torch.arange(n+1,device="cuda",dtype=torch.int64)*2048
b=cv.encode_device("f32", vals, eo)
pred=cv.make_pred(cv.F32, 3.0); res=b.result_buffer(); lib=cv.lib()
def run(k):
    for _ in range(5): b.filter_count_async(pred, result=res)
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(k): b.filter_count_async(pred, result=res)
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/k*1e3
print("no timer  ms/step", run(50), run(200))
lib.cvec_timing_begin(200)
print("timer     ms/step", run(50))
ms=[C.c_float(0) for _ in range(3)]; nr=C.c_int32(0)
lib.cvec_timing_end(C.byref(ms[0]),C.byref(ms[1]),C.byref(ms[2]),C.byref(nr))
print([m.value/nr.value for m in ms])
