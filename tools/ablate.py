#!/usr/bin/env python3
"""Times k_sections (HIP events inside the library) on the f32 filter workload; used with the
CVEC_ABLATE / CVEC_WG_ROUNDS / CVEC_WG_PER_CU debug knobs to price the phases of the kernel."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import compressed_vec_amd as cv  # noqa: E402
from compressed_vec_amd import synth  # noqa: E402

nvec = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
chunks = [synth.batch_values_torch("f32_xor", 16384, 2048, v_first=c)[0] for c in range(0, nvec, 16384)]
vals = torch.cat(chunks)
eo = torch.arange(nvec + 1, device="cuda", dtype=torch.int64) * 2048
b = cv.encode_device("f32", vals, eo)
lib = cv.lib()
for _ in range(3):
    b.filter_count_async(3.0)
torch.cuda.synchronize()
lib.cvec_timing_begin(20)
for _ in range(20):
    b.filter_count_async(3.0)
torch.cuda.synchronize()
ms = [C.c_float(0) for _ in range(3)]
n = C.c_int32(0)
lib.cvec_timing_end(C.byref(ms[0]), C.byref(ms[1]), C.byref(ms[2]), C.byref(n))
print("ablate=%s threads=%s  k_index %.4f  k_sections %.4f  k_finalize %.4f ms" % (
    os.environ.get("CVEC_ABLATE", "0"), os.environ.get("CVEC_WG_PER_CU", "default"),
    ms[0].value / n.value, ms[1].value / n.value, ms[2].value / n.value))
