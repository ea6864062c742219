#!/usr/bin/env python3
"""Headline benchmark: device-resident decode+filter throughput on f32 XOR-encoded vectors.

    python bench.py --gpus N --steps K --warmup W

One "step" = one fused decode+filter+count pass (reader.filter_iter(EqualsSink(3.0)) + count_hits,
benches/bench.rs:238-241 of the reference) over the rank's whole batch, which is already resident in
HBM; for N > 1 every rank owns its own shard of vectors (no data exchange) and the steps' match counts
are all-reduced over RCCL, eight steps per collective, asynchronously (a collective overlaps the kernels
of the next steps; every step's count is globally summed before the clock stops).  Workload: 131072 vectors x 2048 f32 per GPU (1 GiB of
uncompressed values per GPU; at N = 8 this is the 1 Mi vectors x 2048 f32 configuration the metric is quoted on).
Rank 0 prints one JSON line; `value` is whole-job GB/s counted in uncompressed bytes.

`--gpus N` with N > 1 and no rank environment starts its own ranks (python -m torch.distributed.run, one
per GPU) as child processes before anything touches the GPU, and exits with their code."""
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
    ap.add_argument("--no-configs", action="store_true",
                    help="skip the per-configuration table (the reference's four bench vectors at 1 GiB and both "
                         "data sets of the headline shape: filter / masks / decode / encode)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--streams", type=int, default=int(os.environ.get("BENCH_STREAMS", "1")),
                    help="independent steps in flight: step i runs on stream i %% streams with its own workspace")
    ap.add_argument("--prewarm", type=int, default=300,
                    help="untimed filter passes during set-up, before the warmup steps: the GPU needs a few "
                         "hundred launches to reach its steady clock (0.207 -> 0.195 ms per section kernel)")
    ap.add_argument("--event-steps", type=int, default=16,
                    help="timed steps whose kernels are bracketed with HIP events for the roofline figure")
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
    # the same port on the threads this process may use (at most 16, the GPU box's CPU share), vectors split
    # evenly; ctypes releases the GIL
    from concurrent.futures import ThreadPoolExecutor
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    nthr = max(1, min(16, avail, int(os.environ.get("OMP_NUM_THREADS", "16") or 16)))
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
            "value_all_threads": round(mt_gbs, 3), "threads": nthr,
            "host_cpu": host_cpu_description(), "host_logical_cpus": os.cpu_count(), "host_cpus_usable": avail,
            "sample": "first %d of the same %d-element f32 XOR vectors (%d MiB uncompressed), %d passes in %.1f s, "
                      "filter == %.1f + count, oracle/cvec_oracle.c single thread" %
                      (n, VEC_LEN, elems * 4 >> 20, passes, dt, pred)}, total


def host_cpu_description():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or platform.machine()


def launch_ranks(n):
    """Starts n ranks of this script (one per GPU) through torch.distributed.run and returns their exit code.
    Called before this process has touched the GPU; the children are fresh processes."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        rc = launch_ranks(args.gpus)
        if rc:
            print("bench.py: the %d ranks ended with exit code %d (their messages are above)" % (args.gpus, rc),
                  file=sys.stderr)
        sys.exit(rc)
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    have = torch.cuda.device_count()          # counting devices does not initialise the GPU
    if have < local_world or args.gpus != world:
        # every rank sees the same numbers and leaves before the rendezvous, so nobody waits for anybody
        if rank == 0:
            if have < local_world:
                print("bench.py: --gpus %d needs %d GPUs on this node, %d visible" % (args.gpus, local_world, have),
                      file=sys.stderr)
            else:
                print("bench.py: --gpus %d but the launcher started %d ranks" % (args.gpus, world), file=sys.stderr)
        sys.exit(3)
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
    check_pred_t = distinct[distinct != 0][distinct.numel() // 2].reshape(1).clone()   # a value that does occur
    if use_dist:
        dist.broadcast(check_pred_t, 0)                          # every rank checks the same value
    check_pred = float(check_pred_t.item())
    got = batch.filter_count(check_pred)
    want_occ = int((values == check_pred).sum().item())
    assert got == want_occ and (got > 0 or rank != 0), "filter count %d != %d" % (got, want_occ)
    want_timed = int((values == PRED).sum().item())

    pred = cv.make_pred(cv.F32, PRED)
    # The counts are all-reduced over RCCL in buckets: every step writes its result block into the next
    # slot of a bucket, and a full bucket (BUCKET steps, 32 bytes each) goes out as ONE asynchronous
    # all-reduce on RCCL's stream beside the kernels of the following steps.  Three buckets rotate; one is
    # reused only after its reduction has been waited for (a stream-side wait, the host does not block),
    # and the last, partial bucket is reduced before the clock stops, so every step's count is globally
    # summed inside the timed region.  (One collective per step costs ~24 us of GPU time per step: the
    # persistent section kernel leaves the collective's kernel no room until a kernel boundary.)
    # --streams S > 1: consecutive steps are independent queries over the same resident batch; step i is
    # enqueued on stream i % S with a workspace of its own, so the short index / finalize kernels of one
    # step run beside the section kernel of another.
    nstreams = max(1, args.streams)
    views = [batch] + [cv.DeviceBatch(batch.etype, batch.d_bytes, batch.d_byte_off, batch.d_sec_base, batch.n_vectors,
                                      batch.n_sections, batch.n_bytes) for _ in range(nstreams - 1)]
    streams = [torch.cuda.current_stream()] + [torch.cuda.Stream() for _ in range(nstreams - 1)]
    for st in streams[1:]:
        st.wait_stream(streams[0])
    from compressed_vec_amd.shard import CountBucketRing
    ring = CountBucketRing(N.RESULT_WORDS, "cuda", bucket=8, nbuckets=3, streams=streams, reduce=use_dist)

    def step():
        row, si = ring.begin_step()
        with torch.cuda.stream(streams[si]):
            views[si].filter_count_async(pred, result=row)
        return ring.end_step()

    drain = ring.drain
    results = ring.rows

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
    # per-kernel HIP events on the launch stream for the first steps of the timed region only: six event
    # records cost ~0.018 ms per step, which would be charged to every step's wall time otherwise
    # (at most a tenth of the steps, two at least: a short run is not dominated by its own instrumentation)
    N.check(lib.cvec_timing_begin(max(2, min(args.steps // 10, args.event_steps))))
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
    # The reference's predicate (3.0) never occurs in this generator's k / 1.4 values, so the comparison above is
    # 0 == 0.  Behind the clock, the same path (ring, streams, workspaces, collective) runs once per stream with a
    # value that does occur, and every one of those steps has to count what a direct count on the values gives.
    pred_occ = cv.make_pred(cv.F32, check_pred)
    if use_dist:
        wo = torch.tensor([want_occ], device="cuda", dtype=torch.int64)
        dist.all_reduce(wo, op=dist.ReduceOp.SUM)
        want_occ = int(wo.item())
    occ_rows = []
    for _ in range(nstreams):
        row, si = ring.begin_step()
        with torch.cuda.stream(streams[si]):
            views[si].filter_count_async(pred_occ, result=row)
        occ_rows.append(ring.end_step())
    drain()
    torch.cuda.synchronize()
    occ_counts = [int(results[r][0].item()) for r in occ_rows]
    assert all(c == want_occ for c in occ_counts) and want_occ > 0, \
        "occurring predicate %r: counts %r != %d" % (check_pred, occ_counts, want_occ)

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
                   "compressed_bytes_per_element": round(batch.n_bytes / n_elems, 4),
                   "tail_rule": "format (every octet decodes to what was encoded; extra.reference_tail_rule has "
                                "the reference crate's tail decoder on the same step)"},
        "checks": {"timed_predicate": PRED, "timed_count": total_count, "timed_count_expected": want_timed,
                   "occurring_predicate": check_pred, "occurring_counts_same_path": occ_counts,
                   "occurring_count_expected": want_occ},
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
                           "avg_launch_ms": round(k_sections_ms, 4), "launches_timed": nrec.value,
                           "other_kernels_ms": {"k_index": round(ms[0].value / max(nrec.value, 1), 4),
                                                "k_finalize": round(ms[2].value / max(nrec.value, 1), 4)}}
        if not args.no_cpu_baseline and world == 1:   # reported at N = 1 only
            cb, cpu_total = cpu_baseline(batch, 4096, args.cpu_seconds, PRED)
            out["cpu_baseline"] = cb
        if not args.no_extra and world == 1:
            out["extra"] = side_measurements(torch, cv, batch, values, elem_off, raw_bytes, check_pred)
            out["extra"]["first_encode_call_s"] = round(t_first_encode, 4)
            if not args.no_configs:
                del values, batch
                torch.cuda.empty_cache()
                out["extra"]["configs"] = config_table(torch, cv, N)
                out["extra"]["operations"] = operation_table(torch, cv)
                # every configuration's counts (the reference's predicate and a value that occurs), masks and decoded
                # values against the values themselves: a false one means a kernel gave wrong results at full size,
                # whatever its throughput says
                out["extra"]["all_configs_exact"] = all(r["filter_count_equals_direct_count"] and r["exact"]["all"]
                                                        for r in out["extra"]["configs"]) and \
                    all(v.get("exact", True) for v in out["extra"]["operations"].values() if isinstance(v, dict)) and \
                    all(out["extra"][k] for k in ("decode_bit_exact_vs_input", "filter_two_steps_in_flight_exact",
                                                  "filter_resident_index_same_count", "encode_all_equals_batch"))
                if not out["extra"]["all_configs_exact"]:
                    print("bench.py: a configuration's filter count differs from the direct count (see extra.configs)",
                          file=sys.stderr, flush=True)
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


def side_measurements(torch, cv, batch, values, elem_off, raw_bytes, check_pred):
    """Reported next to the headline, outside the timed region: decode through the sink stack into a
    buffer, encode, and the host-inclusive (pinned H2D / D2H) filter rate."""
    ex = {}
    out = torch.empty(batch.n_sections * 256, dtype=torch.float32, device="cuda")
    t = _time_gpu(torch, lambda: batch.decode_async(out, add_const=2.5), 10)
    ex["decode_addconst_vecsink_GBs"] = round(raw_bytes / t / 1e9, 2)
    ok = torch.equal((values + 2.5).view(torch.int32), out[:values.numel()].view(torch.int32))
    ex["decode_bit_exact_vs_input"] = bool(ok)
    del out
    # encode_all as an ingest loop runs it: outputs and workspace allocated once, calls enqueued back to back
    # (cvec_encode_dev neither allocates nor synchronises); the bytes must equal the batch that the headline reads
    ebuf = cv.EncodeBuffers("f32", batch.n_vectors, values.numel())
    t = _time_gpu(torch, lambda: cv.encode_device_async(ebuf, values, elem_off), 10)
    ex["encode_all_GBs"] = round(raw_bytes / t / 1e9, 2)
    ex["encode_all_equals_batch"] = bool(int(ebuf.result[0].item()) == batch.n_bytes and
                                         torch.equal(ebuf.d_bytes[:batch.n_bytes], batch.d_bytes[:batch.n_bytes]) and
                                         torch.equal(ebuf.d_byte_off, batch.d_byte_off))
    del ebuf
    t = _time_gpu(torch, lambda: cv.encode_device("f32", values, elem_off), 5)
    ex["encode_all_with_allocation_and_readback_GBs"] = round(raw_bytes / t / 1e9, 2)
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
    # the same two streams and workspaces with a value that occurs: both results against a direct count
    want_occ = int((values == check_pred).sum().item())
    pred_occ = cv.make_pred(cv.F32, check_pred)
    pred, pred_timed = pred_occ, pred
    two_in_flight(4)
    pred = pred_timed
    ex["filter_two_steps_in_flight_exact"] = bool(want_occ > 0 and all(int(r[0].item()) == want_occ for r in results))
    # the same step over a batch that keeps its section index resident (the header walk done once, by
    # build_index(), instead of in every pass); the headline never uses it
    indexed = views[1].build_index()
    res = indexed.result_buffer()
    t = _time_gpu(torch, lambda: indexed.filter_count_async(pred, result=res), 40)
    ex["filter_resident_index_GBs"] = round(raw_bytes / t / 1e9, 2)
    indexed.filter_count_async(pred_occ, result=res)
    ex["filter_resident_index_same_count"] = bool(int(res.cpu()[0]) == want_occ)
    # the reference crate's tail decoder (cvec_read_options.tail_rule) on the headline step: same kernels with the
    # rule fused in; its count of the occurring value differs from the format's rule by the matches that the crate
    # drops in the last octets of a vector
    plain = cv.DeviceBatch(batch.etype, batch.d_bytes, batch.d_byte_off, batch.d_sec_base, batch.n_vectors,
                           batch.n_sections, batch.n_bytes)
    res2 = plain.result_buffer()
    t_fmt = _time_gpu(torch, lambda: plain.filter_count_async(pred, result=res2, tail_rule="format"), 40)
    t_ref = _time_gpu(torch, lambda: plain.filter_count_async(pred, result=res2, tail_rule="reference"), 40)
    plain.filter_count_async(pred_occ, result=res2, tail_rule="reference")
    ref_occ = int(res2.cpu()[0])
    ex["reference_tail_rule"] = {"filter_GBs": round(raw_bytes / t_ref / 1e9, 2),
                                 "format_rule_same_call_GBs": round(raw_bytes / t_fmt / 1e9, 2),
                                 "occurring_count": ref_occ, "occurring_count_format_rule": want_occ,
                                 "note": "parity of this rule rests on the oracle's reading of "
                                         "src/nibblepack_simd.rs:433-450, :546-554 (no reference fixture covers it)"}
    # ingest through the series appender: the same values arrive as scrapes of 64 per vector, sections are encoded as
    # they complete, finish() must give the batch that encode_all gave
    n_vec, vec_len, k = batch.n_vectors, values.numel() // batch.n_vectors, 64
    mat = values.view(n_vec, vec_len)
    blocks = [mat[:, c0:c0 + k].contiguous() for c0 in range(0, vec_len, k)]
    app = cv.SeriesAppender("f32", n_vec)
    for rep in range(2):                                   # the first fill sizes the handle's buffers
        t_plain, t_flush = [], []
        for blk in blocks:
            before = app.stats()["segments"]
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            app.append(blk)
            torch.cuda.synchronize()
            (t_flush if app.stats()["segments"] > before else t_plain).append(time.perf_counter() - t0)
        t0 = time.perf_counter()
        fin = app.finish(vec_len)
        torch.cuda.synchronize()
        t_fin = time.perf_counter() - t0
    ex["series_appender"] = {
        "series": n_vec, "values_per_scrape": k,
        "scrape_ms": round(1e3 * sum(t_plain) / max(len(t_plain), 1), 4),
        "scrape_completing_a_section_ms": round(1e3 * sum(t_flush) / max(len(t_flush), 1), 4),
        "finish_ms": round(1e3 * t_fin, 3),
        "ingest_and_finish_GBs": round(raw_bytes / (sum(t_plain) + sum(t_flush) + t_fin) / 1e9, 2),
        "equals_encode_all": bool(fin.n_bytes == batch.n_bytes and torch.equal(fin.d_bytes[:fin.n_bytes], batch.d_bytes[:batch.n_bytes])
                                  and torch.equal(fin.d_byte_off, batch.d_byte_off))}
    del app, fin, blocks, mat
    torch.cuda.empty_cache()
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


def _popcount_u8(torch, t):
    """Number of set bits of a uint8 tensor."""
    x = t.view(torch.int32) if t.numel() % 4 == 0 else t.to(torch.int32)
    x = x - ((x >> 1) & 0x55555555)
    x = (x & 0x33333333) + ((x >> 2) & 0x33333333)
    x = (x + (x >> 4)) & 0x0f0f0f0f
    return ((x * 0x01010101) >> 24 & 0xff).sum()


def operation_table(torch, cv):
    """The operations around the single-vector filter, at the shapes of the reference's benches: MultiVectorFilter
    over a dense and a very sparse u32 column in both orders (benches/bench.rs:192-203), match_positions
    (src/filter.rs:302-321), the legacy stream codecs (src/nibblepacking.rs:15-45, :239-325).  Rates in GB/s of
    uncompressed values touched per call (wall time, Python included); every result is checked against a direct
    computation on the values."""
    from compressed_vec_amd import stream, synth
    ops = {}
    nvec, length = 26214, 10000
    chunks = [synth.batch_values_torch("u32_dense", min(4096, nvec - c0), length, v_first=c0)[0] for c0 in range(0, nvec, 4096)]
    dense_v = torch.cat(chunks)
    del chunks
    keep = torch.zeros(length, dtype=torch.bool, device="cuda")
    keep[2475:2525] = True
    keep[7475:7525] = True
    # (the sparse column's values are the dense generator's at another phase, with 15 beside every other 3 of the
    # dense column, so that some rows match in both columns)
    idx = torch.arange(dense_v.numel(), device="cuda")
    sparse_v = torch.where((dense_v == 3) & (idx % 2 == 0), torch.full_like(dense_v, 15), torch.roll(dense_v, 1234))
    sparse_v = torch.where(keep.repeat(nvec), sparse_v, torch.zeros_like(dense_v))
    del idx
    eo = torch.arange(nvec + 1, device="cuda", dtype=torch.int64) * length
    dense = cv.encode_device("u32", dense_v, eo)
    sparse = cv.encode_device("u32", sparse_v, eo)
    raw2 = 2 * dense_v.numel() * 4
    want = int(((dense_v == 3) & (sparse_v == 15)).sum().item())
    for name, cols, preds in (("multi_filter_dense_then_sparse", [dense, sparse], [3, 15]),
                              ("multi_filter_sparse_then_dense", [sparse, dense], [15, 3])):
        got = cv.multi_filter(cols, preds)
        t = _time_gpu(torch, lambda: cv.multi_filter(cols, preds), 10)
        ops[name] = {"GBs": round(raw2 / t / 1e9, 1), "ms": round(t * 1e3, 4), "count": got, "count_expected": want,
                     "exact": bool(got == want and want > 0),
                     "shape": "2 columns x %d vectors x %d u32 (bench.rs:192-203: == 3 on the dense, == 15 on the sparse)" % (nvec, length)}
    fc = dense.filter_masks(3)
    pos, off = fc.match_positions()
    want_pos = int((dense_v == 3).sum().item())
    first = torch.nonzero(dense_v[:length] == 3).view(-1).cpu().numpy()
    t = _time_gpu(torch, lambda: fc.match_positions(), 5)
    # the same with the positions left on the device, into buffers of an earlier call
    dpos, doff = fc.match_positions_async()
    td = _time_gpu(torch, lambda: fc.match_positions_async(dpos, doff), 20)
    same = bool(int(doff[-1].item()) == want_pos and np.array_equal(dpos[:int(off[1])].cpu().numpy().view(np.uint32), first))
    ops["match_positions"] = {"ms": round(t * 1e3, 3), "positions": int(off[-1]), "positions_per_s": round(int(off[-1]) / t, 0),
                              "device_resident_ms": round(td * 1e3, 4), "device_resident_positions_per_s": round(want_pos / td, 0),
                              "exact": bool(int(off[-1]) == want_pos and np.array_equal(pos[:int(off[1])], first) and same),
                              "shape": "%d vectors x %d u32, == 3; ms: positions copied to the host in the timed call, "
                                       "device_resident_ms: left on the device" % (nvec, length)}
    del dpos, doff
    del dense, sparse, dense_v, sparse_v, fc
    torch.cuda.empty_cache()
    n, length = 13107, 10000
    g = torch.Generator(device="cuda")
    g.manual_seed(5)
    inc = torch.cumsum(torch.randint(0, 1000, (n, length), device="cuda", generator=g, dtype=torch.int64), dim=1).reshape(-1)
    flt = (torch.randn(n * length, device="cuda", generator=g, dtype=torch.float64) * 100).view(torch.int64)
    seo = torch.arange(n + 1, device="cuda", dtype=torch.int64) * length
    raw = n * length * 8
    for kind, d_vals in (("u64_delta", inc), ("f64_xor", flt)):
        d_bytes, d_bo, _, d_res = stream.pack_device(kind, d_vals, seo, n * length)
        nb = int(d_res.cpu()[0])
        t_p = _time_gpu(torch, lambda: stream.pack_device(kind, d_vals, seo, n * length), 5)
        back = stream.unpack_device(kind, d_bytes, d_bo, nb, seo, n * length)
        t_u = _time_gpu(torch, lambda: stream.unpack_device(kind, d_bytes, d_bo, nb, seo, n * length), 5)
        out = back[0] if isinstance(back, (tuple, list)) else back
        ops["stream_" + kind] = {"pack_GBs": round(raw / t_p / 1e9, 1), "unpack_GBs": round(raw / t_u / 1e9, 1),
                                 "packed_B_per_value": round(nb / (n * length), 3),
                                 "exact": bool(torch.equal(out.view(torch.int64)[:n * length], d_vals)),
                                 "shape": "%d streams x %d values of 8 bytes" % (n, length)}
    return ops


def config_table(torch, cv, N):
    """The reference's `cargo bench vector` configurations (benches/bench.rs:131-170) as batches of 1 GiB
    of uncompressed values, and both data sets at the headline shape: for each, encode_all, decode through
    AddConst -> VecSink, filter + count and filter with per-section masks, in GB/s of uncompressed bytes,
    with the per-kernel times of the filter pass and the fraction of the HBM peak its dominant kernel
    reaches (uncompressed bytes / kernel time / 8 TB/s, the headline's formula)."""
    from compressed_vec_amd import synth
    lib = cv.lib()
    rows = []

    def kernel_ms(fn, iters):
        N.check(lib.cvec_timing_begin(iters))
        for _ in range(iters):
            fn()
        torch.cuda.synchronize()
        ms = [C.c_float(0), C.c_float(0), C.c_float(0)]
        nrec = C.c_int32(0)
        N.check(lib.cvec_timing_end(C.byref(ms[0]), C.byref(ms[1]), C.byref(ms[2]), C.byref(nrec)))
        return [m.value / max(nrec.value, 1) for m in ms]

    def one(name, tname, kind, nvec, length, pred, add_const):
        chunks = []
        for c0 in range(0, nvec, 4096):
            cn = min(4096, nvec - c0)
            if kind == "f32_splitmix":
                chunks.append(synth.splitmix_f32_torch(cn, length, v_first=c0)[0])
            else:
                chunks.append(synth.batch_values_torch("u32_dense" if kind == "u32_sparse" else kind, cn, length,
                                                       v_first=c0)[0])
        vals = torch.cat(chunks)
        del chunks
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
        ebuf = cv.EncodeBuffers(tname, nvec, vals.numel())
        t_enc = _time_gpu(torch, lambda: cv.encode_device_async(ebuf, vals, eo), 10)
        enc_ok = bool(int(ebuf.result[0].item()) == batch.n_bytes and
                      torch.equal(ebuf.d_bytes[:batch.n_bytes], batch.d_bytes[:batch.n_bytes]))
        del ebuf
        out = torch.empty(batch.n_sections * 256, dtype=vals.dtype, device="cuda")
        masks = torch.empty(batch.n_sections * 32, dtype=torch.uint8, device="cuda")
        res = batch.result_buffer()
        p = cv.make_pred(batch.etype, pred)
        want = int((vals == pred).sum().item())
        t_dec = _time_gpu(torch, lambda: batch.decode_async(out, add_const=add_const), 10)
        # bit for bit (vectors are padded to whole sections in the output)
        dec_ok = bool(torch.equal(out.view(nvec, -1)[:, :length].contiguous().view(-1).view(torch.uint8),
                                  (vals + add_const).view(torch.uint8)))
        t_fil = _time_gpu(torch, lambda: batch.filter_count_async(p, result=res), 30)
        got = int(res[0].item())
        t_msk = _time_gpu(torch, lambda: batch.filter_count_async(p, result=res, masks=masks), 10)
        # a value that occurs (the reference's predicates 3.0 never do in the f32 data): count-only pass, mask pass
        # (its count and the bits of its masks) against a direct comparison on the values
        nz = vals[:1 << 22]
        occ = nz[nz != 0][7].item()
        want_occ = int((vals == occ).sum().item())
        po = cv.make_pred(batch.etype, occ)
        batch.filter_count_async(po, result=res)
        got_occ = int(res[0].item())
        batch.filter_count_async(po, result=res, masks=masks)
        got_occ_m = int(res[0].item())
        pop = int(_popcount_u8(torch, masks))
        exact = {"reference_predicate": bool(got == want), "occurring_predicate": occ, "occurring_count": got_occ,
                 "occurring_count_expected": want_occ, "mask_pass_count": got_occ_m, "mask_bits_set": pop,
                 "decode_equals_values_plus_const": dec_ok, "encode_repeatable": enc_ok}
        exact["all"] = bool(got == want and got_occ == want_occ and got_occ_m == want_occ and pop == want_occ and
                            want_occ > 0 and dec_ok and enc_ok)
        k_fil = kernel_ms(lambda: batch.filter_count_async(p, result=res), 20)
        k_dec = kernel_ms(lambda: batch.decode_async(out, add_const=add_const), 10)
        dom = max(k_fil)
        rows.append({"config": name, "vectors": nvec, "vector_len": length, "raw_MiB": raw >> 20,
                     "compressed_B_per_elem": round(batch.n_bytes / vals.numel(), 3),
                     "encode_GBs": round(raw / t_enc / 1e9, 1), "decode_addconst_GBs": round(raw / t_dec / 1e9, 1),
                     "filter_GBs": round(raw / t_fil / 1e9, 1), "filter_masks_GBs": round(raw / t_msk / 1e9, 1),
                     "filter_ms": round(t_fil * 1e3, 4),
                     "filter_kernels_ms": {"index_or_memset": round(k_fil[0], 4), "sections_or_vstream": round(k_fil[1], 4),
                                           "finalize": round(k_fil[2], 4)},
                     "decode_kernels_ms": {"index_or_memset": round(k_dec[0], 4), "sections_or_vstream": round(k_dec[1], 4),
                                           "finalize": round(k_dec[2], 4)},
                     "filter_roofline_frac": round(raw / (dom * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if dom > 0 else None,
                     # decode moves the compressed bytes in and the uncompressed bytes out
                     "decode_roofline_frac": round((raw + batch.n_bytes) / (max(k_dec) * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                     if max(k_dec) > 0 else None,
                     "filter_count_equals_direct_count": bool(got == want and got_occ == want_occ),
                     "exact": exact})
        del vals, batch, out, masks
        torch.cuda.empty_cache()

    n = 26214                                       # x 10000 x 4 B = 1 GiB
    one("cfg1 u32 dense (bench.rs:131-141)", "u32", "u32_dense", n, 10000, 3, 7)
    one("cfg2 u32 sparse (bench.rs:160-170)", "u32", "u32_sparse", n, 10000, 15, 7)
    one("cfg3 f32 xor (bench.rs:143-150)", "f32", "f32_xor", n, 10000, 3.0, 2.5)
    one("cfg4 u64 delta (bench.rs:152-158)", "u64", "u64_delta", n // 2, 10000, 100003, 5)
    one("cfg5 f32 xor sinusoid, 2048-element vectors", "f32", "f32_xor", VECS_PER_GPU, VEC_LEN, PRED, 2.5)
    one("cfg5 f32 xor splitmix, 2048-element vectors", "f32", "f32_splitmix", VECS_PER_GPU, VEC_LEN, None, 2.5)
    return rows


if __name__ == "__main__":
    main()
