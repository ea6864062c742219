#!/usr/bin/env python3
"""Which element positions does the faulty count-only pass misjudge?  Every section holds exactly one element equal
to the predicate, at a position that differs from section to section; a section that counts 0 names its position."""
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
    nvec = eo.numel() - 1
    length = int(eo[1].item())
    nsv = (length + 255) // 256
    pred = int(pred)
    vals = torch.where(vals == pred, torch.full_like(vals, pred - 1), vals)
    v2 = vals.view(nvec, length)
    slot = torch.arange(nvec * nsv, device=vals.device, dtype=torch.int64).view(nvec, nsv)
    pos = (slot * 37) % 256
    k = torch.arange(nsv, device=vals.device, dtype=torch.int64).view(1, nsv)
    at = k * 256 + pos
    ok = at < length
    rows = torch.arange(nvec, device=vals.device, dtype=torch.int64).view(nvec, 1).expand(nvec, nsv)
    v2[rows[ok], at[ok]] = pred
    truth = ok.view(-1).to(torch.int32)
    posf = pos.view(-1)
    batch = cv.encode_device(tname, vals, eo)
    ns = batch.n_sections
    assert ns == nvec * nsv
    want = int(truth.sum().item())
    ws = batch.workspace()
    off = ((ns + 1) * 8 + 255) // 256 * 256
    hist0 = np.zeros(256, dtype=np.int64)
    nplus = 0
    for rep in range(passes):
        t = batch.filter_count(pred)
        torch.cuda.synchronize()
        sr = ws[off:off + 4 * ns].view(torch.int32)
        cnt = sr & 0xffff
        miss = ((cnt == 0) & (truth == 1)).nonzero().flatten()
        plus = (cnt > truth).nonzero().flatten()
        nplus += int(plus.numel())
        hist0 += np.bincount(posf[miss].cpu().numpy(), minlength=256)
        print("%s pass %d: total %d want %d; sections that lost their match %d, sections with extra matches %d" % (
            name, rep, t, want, int(miss.numel()), int(plus.numel())))
        if rep == 0 and miss.numel():
            m = miss[:48].cpu().numpy()
            print("   lost: " + " ".join("%d@%d" % (int(s), int(posf[s].item())) for s in m))
    print("lost matches by element position (octet k, element e): total %d; extra-match sections %d" % (int(hist0.sum()), nplus))
    h = hist0.reshape(32, 8)
    for kk in range(32):
        print("  octet %2d: %s" % (kk, " ".join("%6d" % int(x) for x in h[kk])))
    print("  by element: " + " ".join("%d" % int(x) for x in h.sum(axis=0)))


if __name__ == "__main__":
    main()
