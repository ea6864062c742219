#!/usr/bin/env python3
"""Headline benchmark: device-resident decode+filter throughput on f32 XOR-encoded vectors.

    python bench.py --gpus N --steps K --warmup W

One "step" = one fused decode+filter+count pass (reader.filter_iter(EqualsSink(3.0)) + count_hits,
benches/bench.rs:238-241 of the reference) over the rank's whole batch, which is already resident in
HBM; for N > 1 every rank owns its own shard of vectors (no data exchange) and the steps' match counts
are all-reduced over RCCL, eight steps per collective, asynchronously (a collective overlaps the kernels
of the next steps; every step's count is globally summed before the clock stops).  Workload: 131072 vectors x 2048 f32 per GPU (1 GiB of
uncompressed values per GPU; at N = 8 this is the 1 Mi vectors x 2048 f32 configuration the metric is quoted on).
Rank 0 prints one JSON line; `value` is whole-job GB/s counted in uncompressed bytes."""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak
VEC_LEN = 2048
VECS_PER_GPU = 131072
PRED = 3.0                     # benches/bench.rs:239


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--vectors-per-gpu", type=int, default=VECS_PER_GPU)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the encode / decode / PCIe side measurements")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--streams", type=int, default=int(os.environ.get("BENCH_STREAMS", "1")),
                    help="independent steps in flight: step i runs on stream i %% streams with its own workspace")
    ap.add_argument("--prewarm", type=int, default=300,
                    help="untimed filter passes during set-up, before the warmup steps: the GPU needs a few "
                         "hundred launches to reach its steady clock (0.207 -> 0.195 ms per section kernel)")
    ap.add_argument("--dist-single", action="store_true",
                    help="take the multi-rank code path (process group, all-reduce) with a single rank; for testing")
    return ap.parse_args()


def cpu_baseline(dev_batch, n_sample_vectors, seconds, pred):
    """Times the CPU oracle (scalar C restatement of the reference path, one thread) on the first
    vectors of the same batch."""
    import torch
    import oracle
    oracle.build()
    n = min(n_sample_vectors, dev_batch.n_vectors)
    bo = dev_batch.d_byte_off[:n + 1].cpu().numpy().view(np.uint64).copy()
    blob = dev_batch.d_bytes[:int(bo[-1])].cpu().numpy()
    elems = n * VEC_LEN
    passes, t0 = 0, time.perf_counter()
    total = None
    while True:
        total, _, _, rc = oracle.batch_filter_count(oracle.F32, blob, bo, pred)
        assert rc == 0
        passes += 1
        dt = time.perf_counter() - t0
        if dt >= seconds:
            break
    gbs = passes * elems * 4 / dt / 1e9
    # the same port on 16 threads (the GPU box's CPU share), vectors split evenly; ctypes releases the GIL
    from concurrent.futures import ThreadPoolExecutor
    nthr = 16
    cuts = [n * t // nthr for t in range(nthr + 1)]
    parts = [(blob[int(bo[cuts[t]]):int(bo[cuts[t + 1]])], bo[cuts[t]:cuts[t + 1] + 1] - bo[cuts[t]]) for t in range(nthr)]

    def one(part):
        return oracle.batch_filter_count(oracle.F32, part[0], part[1], pred)[0]

    with ThreadPoolExecutor(nthr) as pool:
        t1 = time.perf_counter()
        mt_passes, mt_total = 0, None
        while time.perf_counter() - t1 < seconds / 4:
            mt_total = sum(pool.map(one, parts))
            mt_passes += 1
        mt_dt = time.perf_counter() - t1
    assert mt_total == total
    mt_gbs = mt_passes * elems * 4 / mt_dt / 1e9
    return {"value": round(gbs, 4), "unit": "GB/s (uncompressed bytes)", "cores": 1, "kind": "port",
            "value_16_threads": round(mt_gbs, 3),
            "sample": "first %d of the same %d-element f32 XOR vectors (%d MiB uncompressed), %d passes in %.1f s, "
                      "filter == %.1f + count, oracle/cvec_oracle.c single thread" %
                      (n, VEC_LEN, elems * 4 >> 20, passes, dt, pred)}, total


def main():
    args = parse_args()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world == 1:
        print("bench.py: --gpus %d must be launched with torch.distributed.run (one rank per GPU)" % args.gpus,
              file=sys.stderr)
        sys.exit(2)
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or args.dist_single
    if use_dist:
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import compressed_vec_amd as cv
    from compressed_vec_amd import synth
    from compressed_vec_amd import _native as N
    lib = cv.lib()
    assert lib.cvec_init(local_rank) == 0

    nvec = args.vectors_per_gpu
    # ---- build this rank's shard in HBM: generate values, encode them on the GPU ----------------------
    v0 = rank * nvec
    chunks = []
    step_v = 16384
    for c0 in range(0, nvec, step_v):
        cn = min(step_v, nvec - c0)
        vals, _ = synth.batch_values_torch("f32_xor", cn, VEC_LEN, v_first=v0 + c0)
        chunks.append(vals)
    values = torch.cat(chunks)
    del chunks
    elem_off = torch.arange(nvec + 1, device="cuda", dtype=torch.int64) * VEC_LEN
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    batch = cv.encode_device("f32", values, elem_off)
    torch.cuda.synchronize()
    t_first_encode = time.perf_counter() - t0
    n_elems = nvec * VEC_LEN
    raw_bytes = n_elems * 4

    # ---- correctness gate before timing: the count must equal a direct count on the raw values ----------
    distinct = torch.unique(values[:1 << 20])
    check_pred = float(distinct[distinct.numel() // 2].item())   # a value that does occur
    got = batch.filter_count(check_pred)
    want = int((values == check_pred).sum().item())
    assert got == want and got > 0, "filter count %d != %d" % (got, want)
    want_timed = int((values == PRED).sum().item())

    pred = cv.make_pred(cv.F32, PRED)
    # The counts are all-reduced over RCCL in buckets: every step writes its result block into the next
    # slot of a bucket, and a full bucket (BUCKET steps, 32 bytes each) goes out as ONE asynchronous
    # all-reduce on RCCL's stream beside the kernels of the following steps.  Three buckets rotate; one is
    # reused only after its reduction has been waited for (a stream-side wait, the host does not block),
    # and the last, partial bucket is reduced before the clock stops, so every step's count is globally
    # summed inside the timed region.  (One collective per step costs ~24 us of GPU time per step: the
    # persistent section kernel leaves the collective's kernel no room until a kernel boundary.)
    BUCKET, NBUCKETS = 8, 3
    buckets = [torch.zeros(BUCKET, N.RESULT_WORDS, dtype=torch.int64, device="cuda") for _ in range(NBUCKETS)]
    results = [buckets[k][j] for k in range(NBUCKETS) for j in range(BUCKET)]
    RING = BUCKET * NBUCKETS
    pending = [None] * NBUCKETS
    state = {"i": 0}
    # --streams S > 1: consecutive steps are independent queries over the same resident batch; step i is
    # enqueued on stream i % S with a workspace of its own, so the short index / finalize kernels of one
    # step run beside the section kernel of another.
    nstreams = max(1, args.streams)
    views = [batch] + [cv.DeviceBatch(batch.etype, batch.d_bytes, batch.d_byte_off, batch.d_sec_base, batch.n_vectors,
                                      batch.n_sections, batch.n_bytes) for _ in range(nstreams - 1)]
    streams = [torch.cuda.current_stream()] + [torch.cuda.Stream() for _ in range(nstreams - 1)]
    for st in streams[1:]:
        st.wait_stream(streams[0])

    def step():
        i = state["i"]
        j = i % RING
        k = j // BUCKET
        state["i"] += 1
        if j % BUCKET == 0 and pending[k] is not None:       # first step into a bucket: its last reduction is done
            for st in streams:
                with torch.cuda.stream(st):
                    pending[k].wait()
            pending[k] = None
        with torch.cuda.stream(streams[i % nstreams]):
            views[i % nstreams].filter_count_async(pred, result=results[j])
            if use_dist and j % BUCKET == BUCKET - 1:
                for st in streams:                           # the bucket's other steps may be on other streams
                    if st is not streams[i % nstreams]:
                        streams[i % nstreams].wait_stream(st)
                pending[k] = dist.all_reduce(buckets[k], op=dist.ReduceOp.SUM, async_op=True)
        return j

    def drain():
        j = (state["i"] - 1) % RING if state["i"] else 0
        if use_dist and state["i"] and j % BUCKET != BUCKET - 1:   # the partial bucket of the last steps
            k = j // BUCKET
            for st in streams[1:]:
                streams[0].wait_stream(st)
            buckets[k][j % BUCKET + 1:].zero_()
            pending[k] = dist.all_reduce(buckets[k], op=dist.ReduceOp.SUM, async_op=True)
        for k in range(NBUCKETS):
            if pending[k] is not None:
                pending[k].wait()
                pending[k] = None
        state["i"] = 0                      # the next loop starts a fresh bucket

    for _ in range(args.prewarm):          # set-up: clocks to steady state (no collective, not a step)
        batch.filter_count_async(pred, result=results[0])
    torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    drain()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    N.check(lib.cvec_timing_begin(min(args.steps, 4096)))   # per-kernel events for the first 4096 steps at most
    t0 = time.perf_counter()
    last = 0
    for _ in range(args.steps):
        last = step()
    drain()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    result = results[last]
    ms = [C.c_float(0), C.c_float(0), C.c_float(0)]
    nrec = C.c_int32(0)
    N.check(lib.cvec_timing_end(C.byref(ms[0]), C.byref(ms[1]), C.byref(ms[2]), C.byref(nrec)))
    total_count = int(result[0].item())
    if use_dist:
        t = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        wt = torch.tensor([want_timed], device="cuda", dtype=torch.int64)
        dist.all_reduce(wt, op=dist.ReduceOp.SUM)
        want_timed = int(wt.item())
    assert total_count == want_timed, "timed predicate count %d != %d" % (total_count, want_timed)

    ms_per_step = elapsed / args.steps * 1e3
    value = world * raw_bytes * args.steps / elapsed / 1e9
    k_sections_ms = ms[1].value / max(nrec.value, 1)
    achieved = raw_bytes / (k_sections_ms * 1e-3) / 1e9

    out = {
        "metric": "device-resident decode+filter GB/s (uncompressed bytes) per GPU and node, f32 XOR",
        "value": round(value, 2),                 # the node: all ranks together
        "value_per_gpu": round(value / world, 2),
        "unit": "GB/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": "cfg5 shard: %d vectors x %d f32 per GPU, XOR+NibblePack sections, "
                               "filter == %.1f + count_hits%s" % (nvec, VEC_LEN, PRED,
                                                                  ", RCCL all-reduce of the count" if world > 1 else ""),
                   "vectors_per_gpu": nvec, "vector_len": VEC_LEN, "parallelism": "shard%d" % world,
                   "steps_in_flight": nstreams, "prewarm_passes": args.prewarm,
                   "compressed_bytes_per_gpu": batch.n_bytes,
                   "compressed_bytes_per_element": round(batch.n_bytes / n_elems, 4)},
    }
    if rank == 0:
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "roofline_traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                if tj.get("vectors_per_gpu") == nvec:
                    traffic = tj.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out["roofline"] = {"bound": "hbm", "kernel": "k_sections<f32, filter>",
                           "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                           "algorithmic_bytes_per_launch": raw_bytes,
                           "avg_launch_ms": round(k_sections_ms, 4),
                           "other_kernels_ms": {"k_index": round(ms[0].value / max(nrec.value, 1), 4),
                                                "k_finalize": round(ms[2].value / max(nrec.value, 1), 4)}}
        if not args.no_cpu_baseline and world == 1:   # reported at N = 1 only
            cb, cpu_total = cpu_baseline(batch, 4096, args.cpu_seconds, PRED)
            out["cpu_baseline"] = cb
        if not args.no_extra and world == 1:
            out["extra"] = side_measurements(torch, cv, batch, values, elem_off, raw_bytes)
            out["extra"]["first_encode_call_s"] = round(t_first_encode, 4)
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


def _time_gpu(torch, fn, iters):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


def side_measurements(torch, cv, batch, values, elem_off, raw_bytes):
    """Reported next to the headline, outside the timed region: decode through the sink stack into a
    buffer, encode, and the host-inclusive (pinned H2D / D2H) filter rate."""
    ex = {}
    out = torch.empty(batch.n_sections * 256, dtype=torch.float32, device="cuda")
    t = _time_gpu(torch, lambda: batch.decode_async(out, add_const=2.5), 10)
    ex["decode_addconst_vecsink_GBs"] = round(raw_bytes / t / 1e9, 2)
    ok = torch.equal((values + 2.5).view(torch.int32), out[:values.numel()].view(torch.int32))
    ex["decode_bit_exact_vs_input"] = bool(ok)
    del out
    t = _time_gpu(torch, lambda: cv.encode_device("f32", values, elem_off), 5)
    ex["encode_all_GBs"] = round(raw_bytes / t / 1e9, 2)
    # two independent filter steps in flight on two streams (the short index / finalize kernels and the
    # launch gaps of one step hide behind the section kernel of the other)
    pred = cv.make_pred(cv.F32, PRED)
    views = [batch, cv.DeviceBatch(batch.etype, batch.d_bytes, batch.d_byte_off, batch.d_sec_base, batch.n_vectors,
                                   batch.n_sections, batch.n_bytes)]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    results = [v.result_buffer() for v in views]
    torch.cuda.synchronize()

    def two_in_flight(iters):
        for i in range(iters):
            with torch.cuda.stream(streams[i & 1]):
                views[i & 1].filter_count_async(pred, result=results[i & 1])
        torch.cuda.synchronize()

    two_in_flight(6)
    t0 = time.perf_counter()
    two_in_flight(40)
    ex["filter_two_steps_in_flight_GBs"] = round(raw_bytes * 40 / (time.perf_counter() - t0) / 1e9, 2)
    # the same step over a batch that keeps its section index resident (the header walk done once, by
    # build_index(), instead of in every pass); the headline never uses it
    indexed = views[1].build_index()
    res = indexed.result_buffer()
    t = _time_gpu(torch, lambda: indexed.filter_count_async(pred, result=res), 40)
    ex["filter_resident_index_GBs"] = round(raw_bytes / t / 1e9, 2)
    ex["filter_resident_index_same_count"] = bool(int(res.cpu()[0]) == int(batch.filter_count_async(pred).cpu()[0]))
    # host-inclusive: inputs start and results end in page-locked host memory; the library pipelines
    # chunks of vectors through three device slots (copy-in, kernels and copy-out overlap)
    from compressed_vec_amd import hostpath
    host = cv.EncodedBatch(cv.F32, hostpath.pin(batch.d_bytes[:batch.n_bytes].cpu().numpy()),
                           hostpath.pin(batch.d_byte_off.cpu().numpy().view(np.uint64)))
    torch.cuda.synchronize()

    def wall(fn, iters):
        fn()
        t0 = time.perf_counter()
        for _ in range(iters):
            fn()
        return (time.perf_counter() - t0) / iters

    t = wall(lambda: hostpath.filter_count_host(host, PRED), 3)
    ex["host_inclusive_filter_GBs"] = round(raw_bytes / t / 1e9, 2)
    h_out = hostpath.pin(np.zeros(batch.n_sections * 256, dtype=np.float32))
    t = wall(lambda: hostpath.decode_host(host, add_const=2.5, out=h_out), 3)
    ex["host_inclusive_decode_GBs"] = round(raw_bytes / t / 1e9, 2)
    h_vals = hostpath.pin(values.cpu().numpy())
    h_eo = elem_off.cpu().numpy().view(np.uint64)
    h_blob = hostpath.pin(np.zeros(cv.lib().cvec_max_encoded_bytes(cv.F32, batch.n_vectors, values.numel()), dtype=np.uint8))
    t = wall(lambda: hostpath.encode_host("f32", h_vals, h_eo, out=h_blob), 3)
    ex["host_inclusive_encode_GBs"] = round(raw_bytes / t / 1e9, 2)
    ex["host_inclusive_note"] = ("cvec_*_batch on page-locked host arrays: chunked H2D + kernels + D2H pipelined over three "
                                 "streams, uncompressed bytes / wall")
    return ex


if __name__ == "__main__":
    main()
