"""Randomised soak of the device path against the CPU oracle: well-formed ragged batches and
batches with randomly damaged bytes (headers, section headers, octet headers, bit streams).

For every seed and element type the statuses, the per-vector counts of an equality filter and of a
OneOf filter, the masks of the healthy vectors and the decoded values of the healthy vectors must equal
the oracle's.  Prints one line per mismatch (seed, type, what differed) and a summary; exits non-zero
on any mismatch.  Usage: python tools/fuzz_parity.py [--seconds 90] [--seed0 0]
"""
import argparse
import ctypes as C
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

cv = importlib.import_module("compressed-vec_amd")
import oracle  # noqa: E402
from test_gpu_parity import _random_vectors, _bits, ET, NPD  # noqa: E402


def damage(rng, blob, byte_off, n_hits):
    """Flip or overwrite n_hits random bytes; biased toward structural bytes near section starts."""
    blob = bytearray(blob)
    nv = len(byte_off) - 1
    for _ in range(n_hits):
        v = int(rng.integers(0, nv))
        lo, hi = int(byte_off[v]), int(byte_off[v + 1])
        if hi <= lo:
            continue
        how = rng.integers(0, 4)
        if how == 0:                                  # somewhere in the 16-byte header
            p = lo + int(rng.integers(0, min(16, hi - lo)))
        elif how == 1 and hi - lo > 20:               # the first section's header
            p = lo + 16 + int(rng.integers(0, min(12, hi - lo - 16)))
        else:                                         # anywhere
            p = int(rng.integers(lo, hi))
        if rng.random() < 0.5:
            blob[p] ^= 1 << int(rng.integers(0, 8))
        else:
            blob[p] = int(rng.integers(0, 256))
    return bytes(blob)


STATS = {"vectors": 0, "bad_status": 0, "count_mismatch": 0, "elements_decoded": 0, "matches": 0}


def claimed_slots(blob, byte_off, v):
    """Section slots the library gives vector v: ceil(num_elements / 256) from its header, cut to one more
    than its bytes could hold."""
    lo, hi = int(byte_off[v]), int(byte_off[v + 1])
    if hi - lo < 16:
        return 0
    ne = int.from_bytes(bytes(blob[lo + 8:lo + 12]), "little")
    c, bound = (ne + 255) // 256, hi - lo - 16
    return bound + 1 if c > bound else c


def count_mismatch(et, blob, byte_off, v):
    """True when the well-formed sections the oracle walks in vector v are not exactly the header's
    claim: the library reports SectionCountMismatch (100) there, which the reference has no notion of."""
    lo, hi = int(byte_off[v]), int(byte_off[v + 1])
    b = np.ascontiguousarray(blob[lo:hi])
    if oracle.reader_check(et, b)[0] != 0:
        return False
    cnt = C.c_size_t(0)
    rc = oracle.lib().ocv_reader_sections(et, b.ctypes.data_as(C.c_void_p), b.size, C.byref(cnt), None, 0)
    slots = claimed_slots(blob, byte_off, v)
    return cnt.value > slots or (rc == 0 and cnt.value != slots)


def one_round(seed, tname, hits):
    rng = np.random.default_rng(seed)
    flat, off = _random_vectors(rng, tname, int(rng.integers(1, 40)), max_len=int(rng.integers(10, 3000)))
    # the decode pass picks its loop by the batch's average section size: narrow the values of some batches, widen
    # those of others, so that every loop is soaked
    regime = seed % 4
    if flat.size and regime == 1:
        flat = (flat.view(np.uint32 if flat.itemsize == 4 else np.uint64) & (0xFF800000 if tname == "f32" else 0xF)).view(flat.dtype)
    elif flat.size and regime == 2:
        if tname == "f32":
            flat = (flat * rng.standard_normal(flat.size).astype(np.float32)).astype(np.float32)
        else:
            flat = flat ^ rng.integers(0, 1 << 31, size=flat.size).astype(flat.dtype)
    et = ET[tname]
    blob, byte_off = oracle.batch_encode(et, flat, off)
    blob = bytes(np.asarray(blob, dtype=np.uint8))
    if hits:
        blob = damage(rng, blob, byte_off, hits)
    blob = np.frombuffer(blob, dtype=np.uint8).copy()
    nv = len(byte_off) - 1
    host = cv.EncodedBatch(tname, blob, np.asarray(byte_off, dtype=np.uint64))
    dev = host.to_device()
    errs = []
    pick = flat[rng.integers(0, flat.size)] if flat.size else NPD[tname](0)
    preds = [pick, NPD[tname](0), [pick, NPD[tname](1), NPD[tname](0)]]
    sb = dev.d_sec_base.cpu().numpy()
    if [int(sb[v + 1] - sb[v]) for v in range(nv)] != [claimed_slots(blob, byte_off, v) for v in range(nv)]:
        errs.append(("section slots",))
        return errs
    for p in preds:
        total, per, st, masks = dev.filter_count(p, per_vector=True, return_status=True, return_masks=True)
        wtotal, wper, wst, _ = oracle.batch_filter_count(et, blob, byte_off, p)
        STATS["matches"] += int(total)
        if int(total) != int(np.asarray(per).sum()):
            errs.append(("total != sum(per)", p))
        for v in range(nv):
            if st[v] == 100:
                if not count_mismatch(et, blob, byte_off, v):
                    errs.append(("spurious count mismatch", p, v))
                continue
            if st[v] != wst[v] and count_mismatch(et, blob, byte_off, v):
                continue                        # an earlier decode error inside the claimed slots
            if st[v] != wst[v]:
                errs.append(("filter status", p, v, int(st[v]), int(wst[v])))
            elif per[v] != wper[v]:
                errs.append(("filter count", p, v, int(per[v]), int(wper[v])))
            elif st[v] == 0:
                _, wm = oracle.filter_count(et, blob[int(byte_off[v]):int(byte_off[v + 1])], p, return_masks=True)
                gm = np.asarray(masks[32 * int(sb[v]):32 * int(sb[v + 1])])
                if not np.array_equal(gm, wm):
                    errs.append(("masks", p, v))
    dec, doff, dst = dev.decode(return_status=True)
    det = dev.error_details(dst)                # where every failed vector stopped, and the error's payload
    doff = np.asarray(doff)
    wdec, wst, _ = oracle.batch_decode(et, blob, byte_off, doff)
    if seed % 5 == 0:
        # the reference crate's tail decoder, per call, with an AddConstSink in front
        base = NPD[tname](3)
        rdec, _, rst = dev.decode(add_const=base, return_status=True, tail_rule="reference")
        with oracle.ref_tail_quirk():
            qdec, qst, _ = oracle.batch_decode(et, blob, byte_off, doff, add_base=base)
        for v in range(nv):
            a, b = int(doff[v]), int(doff[v + 1])
            if rst[v] == 0 and qst[v] == 0 and not np.array_equal(_bits(np.asarray(rdec[a:b])), _bits(np.asarray(qdec[a:b]))):
                errs.append(("decode values, reference tail rule", v))
    STATS["vectors"] += nv
    STATS["bad_status"] += int(np.count_nonzero(np.asarray(dst)))
    STATS["count_mismatch"] += int(np.count_nonzero(np.asarray(dst) == 100))
    for v in range(nv):
        if dst[v] == 100:
            if not count_mismatch(et, blob, byte_off, v):
                errs.append(("spurious count mismatch in decode", v))
            continue
        if dst[v] != wst[v]:
            if not count_mismatch(et, blob, byte_off, v):
                errs.append(("decode status", v, int(dst[v]), int(wst[v])))
        elif dst[v] != 0:
            rc, k, payload = oracle.error_detail(et, blob[int(byte_off[v]):int(byte_off[v + 1])])
            STATS["details"] = STATS.get("details", 0) + 1
            if rc != dst[v] or (int(det[v, 0]), int(det[v, 1])) != (k, payload):
                errs.append(("error detail", v, int(dst[v]), (int(det[v, 0]), int(det[v, 1])), (k, payload)))
        elif dst[v] == 0:
            a, b = int(doff[v]), int(doff[v + 1])
            STATS["elements_decoded"] += b - a
            if not np.array_equal(_bits(np.asarray(dec[a:b])), _bits(np.asarray(wdec[a:b]))):
                errs.append(("decode values", v))
    if seed % 3 == 0 and nv:
        # the host-buffer entry points cut the batch into chunks of whole vectors: any chunk size must give
        # what the resident batch gave
        os.environ["CVEC_PIPE_CHUNK_BYTES"] = str(int(rng.integers(600, 60000)))
        try:
            ht, hper, hst = cv.filter_count_host(host, preds[0], per_vector=True, return_status=True)
            total, per, st = dev.filter_count(preds[0], per_vector=True, return_status=True)
            if ht != total or not np.array_equal(hper, per) or list(hst) != list(st):
                errs.append(("host filter differs", os.environ["CVEC_PIPE_CHUNK_BYTES"]))
            hv, hoff, hdst = cv.decode_host(host, return_status=True)
            if list(hdst) != list(dst) or not np.array_equal(np.asarray(hoff), doff):
                errs.append(("host decode status/offsets differ", os.environ["CVEC_PIPE_CHUNK_BYTES"]))
            else:
                for v in range(nv):
                    a, b = int(doff[v]), int(doff[v + 1])
                    if dst[v] == 0 and not np.array_equal(_bits(np.asarray(hv[a:b])), _bits(np.asarray(dec[a:b]))):
                        errs.append(("host decode values differ", v, os.environ["CVEC_PIPE_CHUNK_BYTES"]))
                        break
            if not hits:
                enc = cv.encode_host(tname, flat, off)
                if not np.array_equal(np.asarray(enc.blob), blob) or not np.array_equal(np.asarray(enc.byte_off), np.asarray(byte_off)):
                    errs.append(("host encode differs", os.environ["CVEC_PIPE_CHUNK_BYTES"]))
        finally:
            del os.environ["CVEC_PIPE_CHUNK_BYTES"]
    return errs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=90)
    ap.add_argument("--seed0", type=int, default=0)
    args = ap.parse_args()
    t0 = time.time()
    seed, bad, rounds = args.seed0, 0, 0
    last = t0
    while time.time() - t0 < args.seconds:
        for tname in ("u32", "u64", "f32"):
            hits = [0, 1, 3, 12][seed % 4]
            errs = one_round(seed, tname, hits)
            rounds += 1
            for e in errs:
                bad += 1
                print("MISMATCH seed", seed, tname, "hits", hits, e, flush=True)
        seed += 1
        if time.time() - last > 20:
            last = time.time()
            print("progress: seeds", seed - args.seed0, "rounds", rounds, "mismatches", bad, flush=True)
        if bad > 20:
            break
    print("fuzz done: rounds", rounds, "mismatches", bad, STATS, flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
