#!/usr/bin/env python3
"""Repeated passes over one 1 GiB bench configuration, every total and every per-vector count against a direct
count on the values (the section kernel hands out its ranges on demand: which wave decodes which section differs
from pass to pass, the results must not):

    python3 tools/exact_check.py [cfg1|cfg2|cfg3|cfg4|cfg5sin|cfg5mix] [passes]

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
    passes = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    assert cv.lib().cvec_init(0) == 0
    tname, vals, eo, pred, add_const = profile_op.build(name)
    if tname == "f32":
        pred = float(vals[vals != 0][7].item())         # a value that occurs
    batch = cv.encode_device(tname, vals, eo)
    nvec = eo.numel() - 1
    length = int(eo[1].item())
    want_per = (vals.view(nvec, length) == pred).sum(dim=1).cpu().numpy().astype(np.uint64)
    want = int(want_per.sum())
    bad_total = bad_per = 0
    for rep in range(passes):
        if rep % 3 == 0:
            batch.filter_count(pred, per_vector=True, return_status=True, return_masks=True)
        t = batch.filter_count(pred)
        tot, per = batch.filter_count(pred, per_vector=True)
        bad_total += int(t != want)
        bad_per += int(tot != want or not np.array_equal(np.asarray(per), want_per))
    print("%s: %d matches expected; wrong total-only passes %d, wrong per-vector passes %d of %d" % (name, want, bad_total, bad_per, passes))
    sys.exit(1 if bad_total or bad_per else 0)


if __name__ == "__main__":
    main()
