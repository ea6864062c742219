"""The reference's bench configurations at the benchmark's size (1 GiB of values per batch): per-vector counts of the
filter pass against a direct count on the values, masks against a direct comparison, decode against the values,
three passes in a row (the section kernel hands out its ranges on demand, so which wave decodes which section
differs from pass to pass; the results must not).  Sizes beyond what the CPU oracle covers in seconds: the oracle
checks the same code paths on the small batches of the other test files."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


@pytest.fixture(scope="module")
def cv():
    import torch
    assert torch.cuda.is_available(), "these tests need a GPU"
    import compressed_vec_amd as cv
    assert cv.lib().cvec_init(0) == 0
    return cv


@pytest.mark.parametrize("name", ["cfg1", "cfg2", "cfg3", "cfg4", "cfg5sin", "cfg5mix"])
def test_full_size_configuration_per_vector_counts_masks_and_decode(cv, name):
    import torch
    import profile_op
    tname, vals, eo, pred, add_const = profile_op.build(name)
    nvec = eo.numel() - 1
    length = int(eo[1].item())
    if tname == "f32":                          # a value that occurs (the configuration's own predicate may not)
        pred = float(vals[vals != 0][7].item())
    batch = cv.encode_device(tname, vals, eo)
    want_per = (vals.view(nvec, length) == pred).sum(dim=1).cpu().numpy().astype(np.uint64)
    assert want_per.sum() > 0
    nsec = (length + 255) // 256
    # the expected masks: one bit per element, padded to whole sections (the predicate is not zero: padding never matches)
    hit = torch.zeros(nvec, nsec * 256, dtype=torch.bool, device=vals.device)
    hit[:, :length] = vals.view(nvec, length) == pred
    weights = (2 ** torch.arange(8, device=vals.device, dtype=torch.int32)).to(torch.uint8)
    want_masks = (hit.view(-1, 8).to(torch.uint8) * weights).sum(dim=1).to(torch.uint8).cpu().numpy()
    del hit
    for rep in range(3):
        total, per, st, masks = batch.filter_count(pred, per_vector=True, return_status=True, return_masks=True)
        assert not np.asarray(st).any()
        assert np.array_equal(np.asarray(per), want_per), (name, rep, np.nonzero(np.asarray(per) != want_per)[0][:8])
        assert int(total) == int(want_per.sum())
        assert np.array_equal(np.asarray(masks), want_masks), (name, rep)
        assert batch.filter_count(pred) == int(want_per.sum())          # the count-only shapes and the short finalize
        r = batch.filter_count_async([pred, pred]).cpu().numpy()         # the general filter mode (a two-value list)
        assert int(r[0]) == int(want_per.sum()) and int(r[1]) == 0
    out = torch.zeros(batch.n_sections * 256, dtype=vals.dtype, device=vals.device)
    for rep in range(2):
        out.zero_()
        res = batch.decode_async(out).cpu().numpy()
        assert int(res[1]) == 0
        got = out.view(nvec, nsec * 256)[:, :length]
        assert torch.equal(got.contiguous().view(torch.uint8), vals.view(nvec, length).contiguous().view(torch.uint8)), (name, rep)


@pytest.mark.parametrize("name,nval", [("cfg4", 3), ("cfg4", 9), ("cfg1", 3), ("cfg1", 16), ("cfg3", 3)])
def test_full_size_one_of_counts(cv, name, nval):
    """OneOf predicates at the benchmark's size, count-only and per vector, against a direct count: every list
    length compiles to a kernel of its own with a register allocation of its own, and one of them (u64, 3-4
    values) once kept a 64-bit shift count in the last register of its allocation, which gfx950 can read as
    VGPR0 when three or more waves share a SIMD (tools/mb/topvgpr.hip) - wrong counts that no small batch shows."""
    import torch
    import profile_op
    tname, vals, eo, pred, add_const = profile_op.build(name)
    nvec = eo.numel() - 1
    length = int(eo[1].item())
    uniq = torch.unique(vals[:100000])
    preds = [x.item() for x in uniq[uniq != 0][:nval]]
    batch = cv.encode_device(tname, vals, eo)
    hit = torch.zeros_like(vals, dtype=torch.bool)
    for p in preds:
        hit |= vals == p
    want_per = hit.view(nvec, length).sum(dim=1).cpu().numpy().astype(np.uint64)
    del hit
    assert want_per.sum() > 0
    for rep in range(3):
        assert batch.filter_count(preds) == int(want_per.sum()), (name, nval, rep)
        total, per = batch.filter_count(preds, per_vector=True)
        assert int(total) == int(want_per.sum())
        assert np.array_equal(np.asarray(per), want_per), (name, nval, rep)
