"""The parity tests that read batches, run again with the vector-streaming read kernel forced
(cvec_set_read_path(CVEC_PATH_VSTREAM)): by default small test batches take the section-slot kernels,
and both must agree with the oracle on the same inputs: statuses, partial counts, masks and decoded
values, damaged vectors included."""
import importlib.util
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

_here = os.path.dirname(os.path.abspath(__file__))
_spec = importlib.util.spec_from_file_location("_gpu_parity_for_vstream", os.path.join(_here, "test_gpu_parity.py"))
P = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(P)
_spec5 = importlib.util.spec_from_file_location("_gpu_cfg5_for_vstream", os.path.join(_here, "test_gpu_cfg5.py"))
P5 = importlib.util.module_from_spec(_spec5)
_spec5.loader.exec_module(P5)

cv = P.cv


@pytest.fixture(autouse=True)
def _force_path(cv):
    from compressed_vec_amd import _native as N
    assert cv.lib().cvec_set_read_path(N.PATH_VSTREAM) == 0
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
test_cfg5_sinusoid_vectors_vs_oracle = P5.test_cfg5_sinusoid_vectors_vs_oracle
test_cfg5_splitmix_vectors_vs_oracle = P5.test_cfg5_splitmix_vectors_vs_oracle
test_sparse_bench_vector_built_through_append_nulls = P5.test_sparse_bench_vector_built_through_append_nulls


def test_both_kernels_agree_on_a_batch_that_auto_sends_to_vstream(cv, oracle):
    """20000 short ragged u32 vectors: large enough for the automatic choice to pick the vector-streaming
    kernel; totals, per-vector counts, statuses and masks must equal the section-slot kernel's and (on
    a sample of vectors) the oracle's."""
    from compressed_vec_amd import _native as N
    rng = np.random.default_rng(2024)
    lens = rng.integers(0, 1200, size=20000)
    eo = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    vals = rng.integers(0, 40, size=int(eo[-1])).astype(np.uint32)
    vals[rng.random(vals.size) < 0.3] = 0
    zero_runs = rng.integers(0, 20000, size=400)
    for v in zero_runs:                                       # some all-zero and some constant vectors
        vals[int(eo[v]):int(eo[v + 1])] = 0 if v % 2 else 9
    dev = cv.encode_batch("u32", vals, eo).to_device()
    out = {}
    for path in (N.PATH_SLOTS, N.PATH_VSTREAM, N.PATH_AUTO):
        assert cv.lib().cvec_set_read_path(path) == 0
        out[path] = dev.filter_count(9, per_vector=True, return_status=True, return_masks=True)
        dec = dev.decode(add_const=3)
        out[path] = out[path] + (dec[0],)
    a, b, c = out[N.PATH_SLOTS], out[N.PATH_VSTREAM], out[N.PATH_AUTO]
    for x, y in ((a, b), (a, c)):
        assert x[0] == y[0] and all(np.array_equal(p, r) for p, r in zip(x[1:], y[1:]))
    host = dev.download()
    sample = host.byte_off[:501]
    wtotal, wper, wst, rc = oracle.batch_filter_count(oracle.U32, host.blob[:int(sample[-1])], sample, 9)
    assert rc == 0 and np.array_equal(b[1][:500], wper) and not b[2].any()
