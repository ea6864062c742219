#!/usr/bin/env python3
"""Side measurements on batches shaped like the reference's `cargo bench vector` configurations
(benches/bench.rs:131-170): encode_all, decode through AddConst -> VecSink, and filter + count, each
in GB/s of uncompressed bytes, with the values generated on the device.  Not the headline benchmark
(bench.py); prints one JSON object per configuration."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import compressed_vec_amd as cv  # noqa: E402
from compressed_vec_amd import synth  # noqa: E402


def timed(fn, iters=10):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


def run(name, tname, kind, nvec, length, pred, add_const):
    chunks = []
    for c0 in range(0, nvec, 4096):
        if kind == "f32_splitmix":
            chunks.append(synth.splitmix_f32_torch(min(4096, nvec - c0), length, v_first=c0)[0])
        else:
            chunks.append(synth.batch_values_torch("u32_dense" if kind == "u32_sparse" else kind, min(4096, nvec - c0),
                                                   length, v_first=c0)[0])
    vals = torch.cat(chunks)
    if pred is None:
        pred = float(vals[12345].item())        # a value that occurs
    if kind == "u32_sparse":
        # benches/bench.rs:160-170: two runs of 50 values in a vector of nulls (zeros)
        keep = torch.zeros(length, dtype=torch.bool, device="cuda")
        keep[2475:2525] = True
        keep[7475:7525] = True
        vals = torch.where(keep.repeat(nvec), vals, torch.zeros_like(vals))
    eo = torch.arange(nvec + 1, device="cuda", dtype=torch.int64) * length
    raw = vals.numel() * vals.element_size()
    batch = cv.encode_device(tname, vals, eo)
    t_enc = timed(lambda: cv.encode_device(tname, vals, eo), 5)
    out = torch.empty(batch.n_sections * 256, dtype=vals.dtype, device="cuda")
    res = batch.result_buffer()
    p = cv.make_pred(batch.etype, pred)
    want = int((vals == pred).sum().item())
    rec = {"config": name, "vectors": nvec, "vector_len": length, "raw_MiB": raw >> 20,
           "compressed_B_per_elem": round(batch.n_bytes / vals.numel(), 3), "encode_GBs": round(raw / t_enc / 1e9, 1)}
    # both read kernels (cvec_set_read_path): 1 = index + section-slot + finalize, 2 = vector-streaming; 0 = the library's choice
    for path, tag in ((1, "slots"), (2, "vstream"), (0, "auto")):
        assert cv.lib().cvec_set_read_path(path) == 0
        t_dec = timed(lambda: batch.decode_async(out, add_const=add_const))
        t_fil = timed(lambda: batch.filter_count_async(p, result=res), 20)
        got = int(res[0].item())
        if kind != "f32_xor":                   # (10000-element f32 vectors: the zero padding never matches these predicates either)
            assert got == want, (got, want)
        rec["decode_GBs_" + tag] = round(raw / t_dec / 1e9, 1)
        rec["filter_GBs_" + tag] = round(raw / t_fil / 1e9, 1)
        rec["filter_ms_" + tag] = round(t_fil * 1e3, 4)
    rec["filter_matches"] = got
    print(json.dumps(rec), flush=True)


def run_streams(nstreams, length):
    """Legacy delta stream codec (benches/bench.rs:71-93 shape: increasing u64 counters)."""
    from compressed_vec_amd import stream as st
    g = torch.Generator(device="cuda").manual_seed(5)
    inc = torch.randint(0, 1000, (nstreams * length,), device="cuda", dtype=torch.int64, generator=g)
    vals = torch.cumsum(inc.view(nstreams, length), dim=1).reshape(-1).contiguous()
    eo = torch.arange(nstreams + 1, device="cuda", dtype=torch.int64) * length
    total = nstreams * length
    raw = total * 8
    d_bytes, d_bo, _, d_res = st.pack_device("u64_delta", vals, eo, total)
    nbytes = int(d_res[0].item())
    t_pack = timed(lambda: st.pack_device("u64_delta", vals, eo, total), 5)
    out = torch.empty(total, dtype=torch.int64, device="cuda")
    ws = torch.empty(cv.lib().cvec_stream_unpack_workspace_bytes(nstreams, total), dtype=torch.uint8, device="cuda")
    t_unpack = timed(lambda: st.unpack_device("u64_delta", d_bytes, d_bo, nbytes, eo, total, out, ws), 5)
    assert torch.equal(out, vals)
    print(json.dumps({"config": "legacy u64 delta streams", "streams": nstreams, "stream_len": length,
                      "raw_MiB": raw >> 20, "compressed_B_per_elem": round(nbytes / total, 3),
                      "pack_GBs": round(raw / t_pack / 1e9, 1), "unpack_GBs": round(raw / t_unpack / 1e9, 1)}), flush=True)


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 26214          # x 10000 x 4 B = 1 GiB
    run("cfg1 u32 dense", "u32", "u32_dense", n, 10000, 3, 7)
    run("cfg2 u32 sparse", "u32", "u32_sparse", n, 10000, 15, 7)
    run("cfg3 f32 xor", "f32", "f32_xor", n, 10000, 3.0, 2.5)
    run("cfg4 u64 delta", "u64", "u64_delta", n // 2, 10000, 100003, 5)
    # aperiodic f32 (SplitMix64, 12 mantissa bits) at the benchmark's vector shape
    run("cfg5 f32 splitmix", "f32", "f32_splitmix", 131072, 2048, None, 2.5)
    run("cfg5 f32 sinusoid", "f32", "f32_xor", 131072, 2048, 3.0, 2.5)
    run_streams(n // 2, 10000)
