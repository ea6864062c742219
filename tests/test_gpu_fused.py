"""The parity tests that read batches, run again with the one-kernel read path forced
(cvec_set_read_path(CVEC_PATH_FUSED): the section kernel walks the headers of its own vectors and folds the
per-vector results itself).  By default only large batches of short vectors take it; forced, every test
batch does, and all must agree with the oracle on the same inputs: statuses, partial counts, masks and
decoded values, damaged vectors, empty vectors and vectors of several hundred sections included."""
import importlib.util
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

_here = os.path.dirname(os.path.abspath(__file__))
_spec = importlib.util.spec_from_file_location("_gpu_parity_for_fused", os.path.join(_here, "test_gpu_parity.py"))
P = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(P)
_spec5 = importlib.util.spec_from_file_location("_gpu_cfg5_for_fused", os.path.join(_here, "test_gpu_cfg5.py"))
P5 = importlib.util.module_from_spec(_spec5)
_spec5.loader.exec_module(P5)

cv = P.cv


@pytest.fixture(autouse=True)
def _force_path(cv):
    from compressed_vec_amd import _native as N
    assert cv.lib().cvec_set_read_path(N.PATH_FUSED) == 0
    yield
    assert cv.lib().cvec_set_read_path(N.PATH_AUTO) == 0


test_decode_bit_exact_vs_oracle = P.test_decode_bit_exact_vs_oracle
test_f32_roundtrip_is_bitwise_unless_section_is_lossy = P.test_f32_roundtrip_is_bitwise_unless_section_is_lossy
test_filter_count_and_masks_vs_oracle = P.test_filter_count_and_masks_vs_oracle
test_reference_test_cases_through_the_api_mirror = P.test_reference_test_cases_through_the_api_mirror
test_null_section_through_addconst_keeps_base_bits = P.test_null_section_through_addconst_keeps_base_bits
test_empty_batch_and_empty_vectors = P.test_empty_batch_and_empty_vectors
test_malformed_vectors_report_status_and_partial_counts = P.test_malformed_vectors_report_status_and_partial_counts
test_damaged_element_count_is_bounded_by_the_bytes = P.test_damaged_element_count_is_bounded_by_the_bytes
test_randomly_damaged_batches_vs_oracle = P.test_randomly_damaged_batches_vs_oracle
test_overwide_shift_octets_take_the_exact_path = P.test_overwide_shift_octets_take_the_exact_path
test_multi_vector_filter_and_positions_vs_oracle = P.test_multi_vector_filter_and_positions_vs_oracle
test_large_ragged_batch_vs_oracle = P.test_large_ragged_batch_vs_oracle
test_host_buffer_entry_points_chunked_vs_oracle = P.test_host_buffer_entry_points_chunked_vs_oracle
test_reference_bench_configurations_vs_oracle = P.test_reference_bench_configurations_vs_oracle
test_resident_index_gives_the_same_results = P.test_resident_index_gives_the_same_results
test_cfg5_sinusoid_vectors_vs_oracle = P5.test_cfg5_sinusoid_vectors_vs_oracle
test_cfg5_splitmix_vectors_vs_oracle = P5.test_cfg5_splitmix_vectors_vs_oracle
test_sparse_bench_vector_built_through_append_nulls = P5.test_sparse_bench_vector_built_through_append_nulls


@pytest.mark.parametrize("kind,n,vlen", [("f32_xor", 16384, 2048), ("u32_dense", 12000, 700)])
def test_all_read_kernels_agree_on_a_batch_that_auto_sends_to_the_fused_kernel(cv, oracle, kind, n, vlen):
    """Batches of many short vectors, large enough for the automatic choice to pick the one-kernel path: totals,
    per-vector counts, statuses, masks and decoded values must equal the section-slot kernels' and (on a sample of
    vectors) the oracle's; the workspace it leaves must serve an indexed pass."""
    from compressed_vec_amd import _native as N
    from compressed_vec_amd import synth
    rng = np.random.default_rng(77)
    if kind == "f32_xor":
        vals, eo = synth.batch_values("f32_xor", n, vlen)
        et, oet, pred = "f32", oracle.F32, float(vals[1])
    else:
        lens = rng.integers(0, vlen, size=n)
        eo = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
        vals = rng.integers(0, 1 << 20, size=int(eo[-1])).astype(np.uint32)
        vals[rng.random(vals.size) < 0.2] = 77
        et, oet, pred = "u32", oracle.U32, 77
    dev = cv.encode_batch(et, vals, eo).to_device()
    out = {}
    for path in (N.PATH_SLOTS, N.PATH_FUSED, N.PATH_AUTO):
        assert cv.lib().cvec_set_read_path(path) == 0
        r = dev.filter_count(pred, per_vector=True, return_status=True, return_masks=True)
        t = dev.filter_count(pred)
        dec = dev.decode(add_const=3)
        out[path] = r + (dec[0].view(np.uint32), np.uint64(t))
    a, b, c = out[N.PATH_SLOTS], out[N.PATH_FUSED], out[N.PATH_AUTO]
    for x, y in ((a, b), (a, c)):
        assert x[0] == y[0] and all(np.array_equal(p, r) for p, r in zip(x[1:], y[1:]))
    assert a[0] > 0 and int(a[-1]) == a[0]
    # the descriptors and the tag that the one-kernel pass leaves in the workspace are the batch's section index
    assert cv.lib().cvec_set_read_path(N.PATH_FUSED) == 0
    dev.filter_count(pred)
    dev._keep_index = dev._index_valid = True
    r = dev.filter_count(pred, per_vector=True, return_status=True, return_masks=True)
    dev._keep_index = dev._index_valid = False
    assert r[0] == a[0] and all(np.array_equal(p, q) for p, q in zip(r[1:], a[1:4]))
    host = dev.download()
    sample = host.byte_off[:301]
    wtotal, wper, wst, rc = oracle.batch_filter_count(oet, host.blob[:int(sample[-1])], sample, pred)
    assert rc == 0 and np.array_equal(b[1][:300], wper) and not b[2].any()
