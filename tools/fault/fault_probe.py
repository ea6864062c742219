#!/usr/bin/env python3
"""Per-section check of the count-only pass against a direct count on the values; prints the wrong sections and the
in-kernel fault log of a probe build (CVEC_LIB_PATH selects the library)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import compressed_vec_amd as cv  # noqa: E402
import profile_op  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
    passes = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    lib = cv.lib()
    assert lib.cvec_init(0) == 0
    tname, vals, eo, pred, add_const = profile_op.build(name)
    if tname == "f32":
        pred = float(vals[vals != 0][7].item())
    batch = cv.encode_device(tname, vals, eo)
    nvec = eo.numel() - 1
    length = int(eo[1].item())
    nsv = (length + 255) // 256
    ns = batch.n_sections
    assert ns == nvec * nsv, (ns, nvec, nsv)
    m = (vals.view(nvec, length) == pred)
    mp = torch.zeros(nvec, nsv * 256, dtype=torch.bool, device=vals.device)
    mp[:, :length] = m
    truth = mp.view(nvec * nsv, 256).sum(dim=1).to(torch.int32)
    want = int(truth.sum().item())
    ws = batch.workspace()
    off = ((ns + 1) * 8 + 255) // 256 * 256
    has_log = hasattr(lib, "cvec_debug_fault_log")
    try:
        lib.cvec_debug_fault_log
    except AttributeError:
        has_log = False
    logbuf = (C.c_uint32 * (8 + 8 * 512))()
    if has_log:
        lib.cvec_debug_fault_log(logbuf, 1)
    tot_wrong = 0
    for rep in range(passes):
        t = batch.filter_count(pred)
        torch.cuda.synchronize()
        sr = ws[off:off + 4 * ns].view(torch.int32)
        cnt = sr & 0xffff
        st = sr >> 16
        bad = (cnt != truth).nonzero().flatten()
        nb = int(bad.numel())
        tot_wrong += nb
        line = "%s pass %d: total %s (%d vs %d), wrong sections %d of %d, statuses!=0: %d" % (
            name, rep, "ok" if t == want else "WRONG", t, want, nb, ns, int((st != 0).sum().item()))
        print(line)
        if nb:
            b = bad[:12].cpu().numpy()
            g = cnt[bad[:12]].cpu().numpy()
            w = truth[bad[:12]].cpu().numpy()
            print("   first wrong: " + ", ".join("slot %d got %d want %d" % (int(x), int(y), int(z)) for x, y, z in zip(b, g, w)))
            if tname != "f32":
                # which other value would give the counts seen?  (chunk = 16 consecutive slots)
                c0 = int(b[0]) // 16 * 16
                sec_vals = torch.zeros(16, 256, dtype=vals.dtype, device=vals.device)
                for j in range(16):
                    s_ = c0 + j
                    v_, k_ = s_ // nsv, s_ % nsv
                    row = vals.view(nvec, length)[v_, k_ * 256:(k_ + 1) * 256]
                    sec_vals[j, :row.numel()] = row
                for j in range(16):
                    alts = " ".join("%+d:%d" % (d, int((sec_vals[j] == pred + d).sum().item())) for d in range(-4, 5))
                    print("     slot %d got %d want %d | counts of pred+d: %s" % (c0 + j, int(cnt[c0 + j].item()), int(truth[c0 + j].item()), alts))
            diff = (cnt[bad] - truth[bad]).cpu().numpy()
            vals_, counts_ = np.unique(diff, return_counts=True)
            print("   got-want histogram: " + ", ".join("%d:%d" % (int(a), int(c)) for a, c in zip(vals_, counts_)))
        if has_log:
            lib.cvec_debug_fault_log(logbuf, 1)
            n = logbuf[0]
            print("   fault log: %d events" % n)
            for i in range(min(n, 10)):
                r = [logbuf[8 + 8 * i + j] for j in range(8)]
                hw = r[6]
                print("     kind %d a=%d b=0x%x c=0x%x d=0x%x e=0x%x  hw_id=0x%08x (wave %d simd %d cu %d sh %d se %d) xcc %d" % (
                    r[0], r[1], r[2], r[3], r[4], r[5], hw, hw & 15, (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 12) & 1,
                    (hw >> 13) & 7, r[7] & 15))
    print("%s: wrong sections over %d passes: %d" % (name, passes, tot_wrong))


if __name__ == "__main__":
    main()
