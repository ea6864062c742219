#!/usr/bin/env python3
"""One configuration of the per-configuration table, every read / write operation a fixed number of times, for
rocprofv3 (tools/profile_r02.sh runs it under --kernel-trace --stats and under the FETCH_SIZE / WRITE_SIZE
counters, one pass each):

    python3 tools/profile_op.py <cfg1|cfg2|cfg3|cfg4|cfg5sin|cfg5mix> [iters]

Per configuration: `iters` x (filter + count, filter with per-section masks, decode through AddConst -> VecSink,
encode_all).  The kernels of the four operations are different template instantiations, so the profiler's
per-kernel rows tell them apart.  Prints the algorithmic byte counts as one JSON line."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import compressed_vec_amd as cv  # noqa: E402
from compressed_vec_amd import synth  # noqa: E402

CONFIGS = {
    "cfg1": ("u32", "u32_dense", 26214, 10000, 3, 7),
    "cfg2": ("u32", "u32_sparse", 26214, 10000, 15, 7),
    "cfg3": ("f32", "f32_xor", 26214, 10000, 3.0, 2.5),
    "cfg4": ("u64", "u64_delta", 13107, 10000, 100003, 5),
    "cfg5sin": ("f32", "f32_xor", 131072, 2048, 3.0, 2.5),
    "cfg5mix": ("f32", "f32_splitmix", 131072, 2048, None, 2.5),
}


def build(name):
    tname, kind, nvec, length, pred, add_const = CONFIGS[name]
    chunks = []
    for c0 in range(0, nvec, 4096):
        cn = min(4096, nvec - c0)
        if kind == "f32_splitmix":
            chunks.append(synth.splitmix_f32_torch(cn, length, v_first=c0)[0])
        else:
            chunks.append(synth.batch_values_torch("u32_dense" if kind == "u32_sparse" else kind, cn, length, v_first=c0)[0])
    vals = torch.cat(chunks)
    if pred is None:
        pred = float(vals[12345].item())
    if kind == "u32_sparse":
        keep = torch.zeros(length, dtype=torch.bool, device="cuda")
        keep[2475:2525] = True
        keep[7475:7525] = True
        vals = torch.where(keep.repeat(nvec), vals, torch.zeros_like(vals))
    eo = torch.arange(nvec + 1, device="cuda", dtype=torch.int64) * length
    return tname, vals, eo, pred, add_const


def main():
    name = sys.argv[1]
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    assert cv.lib().cvec_init(0) == 0
    tname, vals, eo, pred, add_const = build(name)
    batch = cv.encode_device(tname, vals, eo)
    out = torch.empty(batch.n_sections * 256, dtype=vals.dtype, device="cuda")
    masks = torch.empty(batch.n_sections * 32, dtype=torch.uint8, device="cuda")
    res = batch.result_buffer()
    p = cv.make_pred(batch.etype, pred)
    torch.cuda.synchronize()
    for _ in range(iters):
        batch.filter_count_async(p, result=res)
    for _ in range(iters):
        batch.filter_count_async(p, result=res, masks=masks)
    for _ in range(iters):
        batch.decode_async(out, add_const=add_const)
    for _ in range(iters):
        cv.encode_device(tname, vals, eo)
    torch.cuda.synchronize()
    raw = vals.numel() * vals.element_size()
    print(json.dumps({"config": name, "iters": iters, "uncompressed_bytes": raw, "compressed_bytes": batch.n_bytes,
                      "n_sections": batch.n_sections, "n_vectors": batch.n_vectors,
                      "algorithmic_bytes": {"filter": raw, "filter_masks": raw + 32 * batch.n_sections,
                                            "decode": raw + 256 * vals.element_size() * batch.n_sections,
                                            "encode": 2 * raw + batch.n_bytes}}), flush=True)


if __name__ == "__main__":
    main()
