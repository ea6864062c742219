#!/usr/bin/env python3
"""Count-only OneOf passes over a 1 GiB bench configuration against a direct count on the values:

    python3 tools/oneof_check.py [cfg4|cfg1|...] [number of values] [passes]

Exits non-zero on any difference."""
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
    nval = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    passes = int(sys.argv[3]) if len(sys.argv) > 3 else 6
    assert cv.lib().cvec_init(0) == 0
    tname, vals, eo, pred, add_const = profile_op.build(name)
    uniq = torch.unique(vals[:100000])
    uniq = uniq[uniq != 0][:nval]
    preds = [x.item() for x in uniq]
    batch = cv.encode_device(tname, vals, eo)
    nvec = eo.numel() - 1
    length = int(eo[1].item())
    m = torch.zeros_like(vals, dtype=torch.bool)
    for p in preds:
        m |= vals == p
    want_per = m.view(nvec, length).sum(dim=1).cpu().numpy().astype(np.uint64)
    want = int(want_per.sum())
    bad = 0
    for rep in range(passes):
        t = batch.filter_count(preds)
        tot, per = batch.filter_count(preds, per_vector=True)
        nbad = int((np.asarray(per) != want_per).sum())
        print("%s one of %d values, pass %d: total %d (want %d), per-vector pass %d, vectors that differ %d" % (
            name, len(preds), rep, t, want, tot, nbad))
        bad += int(t != want) + int(tot != want) + nbad
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
