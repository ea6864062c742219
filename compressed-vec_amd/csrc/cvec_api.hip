// cvec_api.hip — the extern "C" entry points declared in include/cvec.h.
#include <stdio.h>
#include <string.h>

#include <vector>

#include "cvec_device.hpp"

namespace cvec {
// cvec_decode.hip
hipError_t launch_sec_base(const uint8_t *d_bytes, const uint64_t *d_byte_off, uint64_t n, uint64_t *d_sec_base,
                           uint64_t *d_scratch, hipStream_t stream);
uint64_t scan_scratch_words(uint64_t n);
hipError_t launch_read(int mode, const cvec_dbatch_view &b, const PredArgs &pa, const SinkArgs &sa, uint64_t *desc,
                       uint32_t *sec_res, int32_t *vstatus, uint64_t *d_result, uint64_t *d_per_vector,
                       int32_t *d_status, uint8_t *d_masks, void *d_out, hipStream_t stream);
hipError_t timing_begin(int max_records);
hipError_t timing_end(float ms[3], int *n_records);
// cvec_filter_ops.hip
struct MaskColumns {
    const uint8_t *masks[CVEC_MAX_PRED_VALUES];
    const uint64_t *sec_base[CVEC_MAX_PRED_VALUES];
    const uint32_t *valid[CVEC_MAX_PRED_VALUES];
    uint32_t k;
};
hipError_t launch_valid_sections(const uint64_t *sec_base, uint64_t n, const uint64_t *desc, const uint32_t *sec_res,
                                 const int32_t *vstatus, uint32_t *valid, hipStream_t stream);
hipError_t launch_multi_and(const MaskColumns &cols, uint64_t n_rows, uint64_t n_slots0, uint8_t *out_masks,
                            uint16_t *popc, uint8_t *ended, uint64_t *per_row, uint32_t *row_sections,
                            uint64_t *result, hipStream_t stream);
hipError_t launch_match_positions(const uint8_t *masks, const uint64_t *sec_base, uint64_t n, const uint32_t *valid,
                                  uint64_t n_slots, uint64_t *counts, uint64_t *scan, uint64_t *scan_scratch,
                                  uint64_t *pos_off, uint32_t *positions, uint64_t cap, hipStream_t stream);
// cvec_stream.hip / cvec_encode.hip: legacy stream codecs
size_t stream_unpack_workspace_bytes(uint64_t n_streams, uint64_t n_values_total);
hipError_t launch_stream_unpack(int kind, const uint8_t *d_bytes, const uint64_t *d_byte_off, uint64_t n_bytes,
                                const uint64_t *d_elem_off, uint64_t n, uint64_t n_values_total, uint64_t *d_out,
                                int32_t *d_status, uint64_t *d_result, void *d_ws, hipStream_t stream);
size_t stream_pack_workspace_bytes(uint64_t n_streams, uint64_t n_values_total);
hipError_t launch_stream_pack(int kind, const uint64_t *d_values, const uint64_t *d_elem_off, uint64_t n,
                              uint64_t n_values_total, uint8_t *d_bytes, uint64_t *d_byte_off, int32_t *d_status,
                              uint64_t *d_result, void *d_ws, hipStream_t stream);
// cvec_encode.hip
size_t encode_workspace_bytes(uint64_t n_vectors, uint64_t n_elements_total);
hipError_t launch_encode(int32_t etype, const void *d_values, const uint64_t *d_elem_off, uint64_t n_vectors,
                         uint64_t n_elements_total, uint8_t *d_bytes, size_t bytes_cap, uint64_t *d_byte_off,
                         uint64_t *d_sec_base, uint64_t *d_result, void *d_workspace, size_t workspace_bytes,
                         hipStream_t stream);
} // namespace cvec

using namespace cvec;

static thread_local char g_hip_err[256] = "";

static cvec_status hip_fail(hipError_t e, const char *what) {
    snprintf(g_hip_err, sizeof(g_hip_err), "%s: %s", what, hipGetErrorString(e));
    return CVEC_ERR_HIP;
}
#define HIP_TRY(expr)                                   \
    do {                                                \
        hipError_t e__ = (expr);                        \
        if (e__ != hipSuccess) return hip_fail(e__, #expr); \
    } while (0)

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
static inline bool etype_ok(int32_t t) { return t == CVEC_U32 || t == CVEC_U64 || t == CVEC_F32; }
static inline int elem_width(int32_t t) { return t == CVEC_U64 ? 8 : 4; }

namespace {
struct DevBuf {
    void *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t n) { return hipMalloc(&p, n ? n : 16); }
    template <class T> T *as() { return (T *)p; }
};

// Uploads blob + offsets and builds sec_base; fills `view`.
cvec_status upload_batch(int32_t etype, const uint8_t *h_bytes, const uint64_t *h_byte_off, uint64_t n,
                         DevBuf &bytes, DevBuf &boff, DevBuf &sbase, cvec_dbatch_view *view) {
    if (!etype_ok(etype) || !h_byte_off || (n && !h_bytes)) return CVEC_ERR_BAD_ARGUMENT;
    for (uint64_t v = 0; v < n; v++)
        if (h_byte_off[v + 1] < h_byte_off[v]) return CVEC_ERR_BAD_ARGUMENT;
    const uint64_t nbytes = h_byte_off[n];
    HIP_TRY(bytes.alloc(align_up(nbytes + 16, 256)));
    HIP_TRY(boff.alloc((n + 1) * 8));
    HIP_TRY(sbase.alloc((n + 1) * 8));
    if (nbytes) HIP_TRY(hipMemcpy(bytes.p, h_bytes, nbytes, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(boff.p, h_byte_off, (n + 1) * 8, hipMemcpyHostToDevice));
    uint64_t nsec = 0;
    cvec_status st = cvec_sec_base_dev(bytes.as<uint8_t>(), boff.as<uint64_t>(), n, sbase.as<uint64_t>(), &nsec, nullptr);
    if (st) return st;
    view->etype = etype;
    view->_pad = 0;
    view->n_vectors = n;
    view->n_sections = nsec;
    view->n_bytes = nbytes;
    view->d_bytes = bytes.as<uint8_t>();
    view->d_byte_off = boff.as<uint64_t>();
    view->d_sec_base = sbase.as<uint64_t>();
    return CVEC_OK;
}

cvec_status summary_status(const uint64_t res[CVEC_RESULT_WORDS]) {
    return res[1] ? (cvec_status)res[2] : CVEC_OK;
}
} // namespace

extern "C" {

const char *cvec_version(void) { return "cvec-mi355x 0.1 (gfx950)"; }

const char *cvec_last_hip_error(void) { return g_hip_err; }

const char *cvec_status_name(cvec_status s) {
    switch (s) {
    case CVEC_OK: return "Ok";
    case CVEC_NOT_ENOUGH_SPACE: return "NotEnoughSpace";
    case CVEC_INPUT_TOO_SHORT: return "InputTooShort";
    case CVEC_BAD_OFFSET: return "BadOffset";
    case CVEC_INVALID_SECTION_TYPE: return "InvalidSectionType";
    case CVEC_INVALID_FORMAT: return "InvalidFormat";
    case CVEC_INVALID_NUM_ROWS: return "InvalidNumRows";
    case CVEC_WRONG_VECTOR_TYPE: return "WrongVectorType";
    case CVEC_SCROLL_ERR: return "ScrollErr";
    case CVEC_ERR_SECTION_COUNT: return "SectionCountMismatch";
    case CVEC_ERR_BAD_ARGUMENT: return "BadArgument";
    case CVEC_ERR_HIP: return "HipError";
    case CVEC_ERR_NO_DEVICE: return "NoDevice";
    case CVEC_ERR_WORKSPACE: return "WorkspaceTooSmall";
    default: return "Unknown";
    }
}

cvec_status cvec_init(int device) {
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count == 0) return CVEC_ERR_NO_DEVICE;
    if (device < 0 || device >= count) return CVEC_ERR_BAD_ARGUMENT;
    HIP_TRY(hipSetDevice(device));
    (void)hipFree(nullptr);
    return CVEC_OK;
}

cvec_status cvec_timing_begin(int32_t max_records) {
    if (max_records <= 0 || max_records > 4096) return CVEC_ERR_BAD_ARGUMENT;
    HIP_TRY(timing_begin(max_records));
    return CVEC_OK;
}

cvec_status cvec_timing_end(float *ms_index, float *ms_sections, float *ms_finalize, int32_t *n_records) {
    float ms[3];
    int n = 0;
    HIP_TRY(timing_end(ms, &n));
    if (ms_index) *ms_index = ms[0];
    if (ms_sections) *ms_sections = ms[1];
    if (ms_finalize) *ms_finalize = ms[2];
    if (n_records) *n_records = n;
    return CVEC_OK;
}

// Workspace layout (each part 256-byte aligned):
//   desc[n_sections] u64 | sec_res[n_sections] u32 | vstatus[n_vectors] i32 | scan scratch
struct ReadWs {
    uint64_t *desc;
    uint32_t *sec_res;
    int32_t *vstatus;
    uint64_t *scan;
    size_t total;
};
static ReadWs carve_read_ws(void *base, uint64_t n_vectors, uint64_t n_sections) {
    ReadWs w;
    size_t off = 0;
    uint8_t *b = (uint8_t *)base;
    w.desc = (uint64_t *)(b + off); off += align_up((n_sections + 1) * 8, 256);
    w.sec_res = (uint32_t *)(b + off); off += align_up((n_sections + 1) * 4, 256);
    w.vstatus = (int32_t *)(b + off); off += align_up((n_vectors + 1) * 4, 256);
    w.scan = (uint64_t *)(b + off); off += align_up(scan_scratch_words(n_vectors) * 8, 256);
    w.total = off;
    return w;
}

size_t cvec_workspace_bytes(uint64_t n_vectors, uint64_t n_sections) {
    return carve_read_ws(nullptr, n_vectors, n_sections).total;
}

cvec_status cvec_sec_base_dev(const uint8_t *d_bytes, const uint64_t *d_byte_off, uint64_t n_vectors,
                              uint64_t *d_sec_base, uint64_t *h_n_sections, void *stream) {
    if (!d_byte_off || !d_sec_base || (n_vectors && !d_bytes)) return CVEC_ERR_BAD_ARGUMENT;
    hipStream_t s = (hipStream_t)stream;
    uint64_t *scratch = nullptr;
    HIP_TRY(hipMalloc((void **)&scratch, scan_scratch_words(n_vectors) * 8));
    hipError_t e = launch_sec_base(d_bytes, d_byte_off, n_vectors, d_sec_base, scratch, s);
    uint64_t total = 0;
    if (e == hipSuccess) e = hipMemcpyAsync(&total, d_sec_base + n_vectors, 8, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(scratch);
    if (e != hipSuccess) return hip_fail(e, "cvec_sec_base_dev");
    if (h_n_sections) *h_n_sections = total;
    return CVEC_OK;
}

static cvec_status check_view(const cvec_dbatch_view *b) {
    if (!b || !etype_ok(b->etype)) return CVEC_ERR_BAD_ARGUMENT;
    if (!b->d_byte_off || !b->d_sec_base) return CVEC_ERR_BAD_ARGUMENT;
    if (b->n_vectors && !b->d_bytes) return CVEC_ERR_BAD_ARGUMENT;
    if (((uintptr_t)b->d_bytes & 15) != 0) return CVEC_ERR_BAD_ARGUMENT;
    if (b->n_bytes >= (1ull << 40)) return CVEC_ERR_BAD_ARGUMENT;
    return CVEC_OK;
}

cvec_status cvec_filter_count_dev(const cvec_dbatch_view *batch, const cvec_pred *pred, uint64_t *d_result,
                                  uint64_t *d_per_vector, int32_t *d_status, uint8_t *d_masks, void *d_workspace,
                                  size_t workspace_bytes, void *stream) {
    cvec_status st = check_view(batch);
    if (st) return st;
    if (!pred || !d_result || !d_workspace) return CVEC_ERR_BAD_ARGUMENT;
    if (pred->kind != CVEC_PRED_EQ && pred->kind != CVEC_PRED_ONEOF) return CVEC_ERR_BAD_ARGUMENT;
    const uint32_t npred = pred->kind == CVEC_PRED_EQ ? 1u : pred->count;
    if (npred == 0 || npred > CVEC_MAX_PRED_VALUES) return CVEC_ERR_BAD_ARGUMENT;
    ReadWs w = carve_read_ws(d_workspace, batch->n_vectors, batch->n_sections);
    if (w.total > workspace_bytes) return CVEC_ERR_WORKSPACE;
    PredArgs pa;
    memset(&pa, 0, sizeof(pa));
    pa.count = npred;
    for (uint32_t i = 0; i < npred; i++) {
        uint64_t v = pred->values[i];
        if (batch->etype != CVEC_U64) v &= 0xffffffffull;
        pa.v[i] = v;
        // pred_matches_zero (reference src/filter.rs:119-121, :151-154): num::Zero::is_zero, so -0.0 counts
        bool z = batch->etype == CVEC_F32 ? ((v & 0x7fffffffull) == 0) : (v == 0);
        if (z) pa.match_zero = 1;
    }
    SinkArgs sa;
    memset(&sa, 0, sizeof(sa));
    hipError_t e = launch_read(0, *batch, pa, sa, w.desc, w.sec_res, w.vstatus, d_result, d_per_vector, d_status,
                               d_masks, nullptr, (hipStream_t)stream);
    if (e != hipSuccess) return hip_fail(e, "cvec_filter_count_dev");
    return CVEC_OK;
}

cvec_status cvec_decode_dev(const cvec_dbatch_view *batch, const cvec_sink_chain *chain, void *d_out,
                            uint64_t *d_result, int32_t *d_status, void *d_workspace, size_t workspace_bytes,
                            void *stream) {
    cvec_status st = check_view(batch);
    if (st) return st;
    if (!d_result || !d_workspace || (batch->n_sections && !d_out)) return CVEC_ERR_BAD_ARGUMENT;
    if (((uintptr_t)d_out & 15) != 0) return CVEC_ERR_BAD_ARGUMENT; // the kernel stores 8 / 16 bytes per lane
    ReadWs w = carve_read_ws(d_workspace, batch->n_vectors, batch->n_sections);
    if (w.total > workspace_bytes) return CVEC_ERR_WORKSPACE;
    PredArgs pa;
    memset(&pa, 0, sizeof(pa));
    SinkArgs sa;
    memset(&sa, 0, sizeof(sa));
    if (chain && chain->add_const) {
        sa.add_const = 1;
        sa.base = batch->etype == CVEC_U64 ? chain->add_const_bits : (chain->add_const_bits & 0xffffffffull);
    }
    hipError_t e = launch_read(1, *batch, pa, sa, w.desc, w.sec_res, w.vstatus, d_result, nullptr, d_status, nullptr,
                               d_out, (hipStream_t)stream);
    if (e != hipSuccess) return hip_fail(e, "cvec_decode_dev");
    return CVEC_OK;
}

static_assert(CVEC_MAX_COLUMNS == CVEC_MAX_PRED_VALUES, "MaskColumns is sized by CVEC_MAX_PRED_VALUES");

cvec_status cvec_valid_sections_dev(const cvec_dbatch_view *batch, const void *d_workspace, size_t workspace_bytes,
                                    uint32_t *d_valid, void *stream) {
    cvec_status st = check_view(batch);
    if (st) return st;
    if (!d_workspace || (batch->n_vectors && !d_valid)) return CVEC_ERR_BAD_ARGUMENT;
    ReadWs w = carve_read_ws((void *)d_workspace, batch->n_vectors, batch->n_sections);
    if (w.total > workspace_bytes) return CVEC_ERR_WORKSPACE;
    hipError_t e = launch_valid_sections(batch->d_sec_base, batch->n_vectors, w.desc, w.sec_res, w.vstatus, d_valid,
                                         (hipStream_t)stream);
    if (e != hipSuccess) return hip_fail(e, "cvec_valid_sections_dev");
    return CVEC_OK;
}

// multi workspace: popc[n] u16 | ended[n] u8
size_t cvec_multi_workspace_bytes(uint64_t n_sections_col0) {
    return align_up((n_sections_col0 + 1) * 2, 256) + align_up(n_sections_col0 + 1, 256);
}

cvec_status cvec_multi_filter_count_dev(const cvec_mask_column *cols, uint32_t k, uint64_t n_rows,
                                        uint64_t *d_result, uint64_t *d_per_row, uint32_t *d_row_sections,
                                        uint8_t *d_masks_out, void *d_workspace, size_t workspace_bytes,
                                        void *stream) {
    if (!cols || k == 0 || k > CVEC_MAX_COLUMNS || !d_result || !d_workspace) return CVEC_ERR_BAD_ARGUMENT;
    MaskColumns mc;
    memset(&mc, 0, sizeof(mc));
    mc.k = k;
    for (uint32_t c = 0; c < k; c++) {
        if (!cols[c].d_sec_base || (n_rows && !cols[c].d_valid) || (cols[c].n_sections && !cols[c].d_masks))
            return CVEC_ERR_BAD_ARGUMENT;
        if (((uintptr_t)cols[c].d_masks & 15) != 0) return CVEC_ERR_BAD_ARGUMENT;
        mc.masks[c] = cols[c].d_masks; mc.sec_base[c] = cols[c].d_sec_base; mc.valid[c] = cols[c].d_valid;
    }
    if (((uintptr_t)d_masks_out & 15) != 0) return CVEC_ERR_BAD_ARGUMENT;
    const uint64_t n0 = cols[0].n_sections;
    if (cvec_multi_workspace_bytes(n0) > workspace_bytes) return CVEC_ERR_WORKSPACE;
    uint16_t *popc = (uint16_t *)d_workspace;
    uint8_t *ended = (uint8_t *)d_workspace + align_up((n0 + 1) * 2, 256);
    hipError_t e = launch_multi_and(mc, n_rows, n0, d_masks_out, popc, ended, d_per_row, d_row_sections, d_result,
                                    (hipStream_t)stream);
    if (e != hipSuccess) return hip_fail(e, "cvec_multi_filter_count_dev");
    return CVEC_OK;
}

// positions workspace: counts[n] u64 | scan[n+1] u64 | scan scratch
size_t cvec_positions_workspace_bytes(uint64_t n_sections) {
    return align_up((n_sections + 1) * 8, 256) * 2 + align_up(scan_scratch_words(n_sections) * 8, 256);
}

cvec_status cvec_match_positions_dev(const cvec_mask_column *col, uint64_t n_vectors, uint64_t *d_pos_off,
                                     uint32_t *d_positions, uint64_t positions_cap, void *d_workspace,
                                     size_t workspace_bytes, void *stream) {
    if (!col || !col->d_sec_base || !d_pos_off || !d_workspace) return CVEC_ERR_BAD_ARGUMENT;
    if ((n_vectors && !col->d_valid) || (col->n_sections && !col->d_masks) || (positions_cap && !d_positions))
        return CVEC_ERR_BAD_ARGUMENT;
    if (((uintptr_t)col->d_masks & 15) != 0) return CVEC_ERR_BAD_ARGUMENT;
    const uint64_t ns = col->n_sections;
    if (cvec_positions_workspace_bytes(ns) > workspace_bytes) return CVEC_ERR_WORKSPACE;
    uint8_t *b = (uint8_t *)d_workspace;
    uint64_t *counts = (uint64_t *)b;
    uint64_t *scan = (uint64_t *)(b + align_up((ns + 1) * 8, 256));
    uint64_t *scratch = (uint64_t *)(b + 2 * align_up((ns + 1) * 8, 256));
    hipError_t e = launch_match_positions(col->d_masks, col->d_sec_base, n_vectors, col->d_valid, ns, counts, scan,
                                          scratch, d_pos_off, d_positions, positions_cap, (hipStream_t)stream);
    if (e != hipSuccess) return hip_fail(e, "cvec_match_positions_dev");
    return CVEC_OK;
}

static inline bool stream_kind_ok(int32_t k) { return k == CVEC_STREAM_U64_DELTA || k == CVEC_STREAM_F64_XOR; }

size_t cvec_stream_max_packed_bytes(int32_t kind, uint64_t n_streams, uint64_t n_values_total) {
    // every octet at most 2 + 64 bytes; a stream of n values has ceil(n/8) octets; XOR streams add 8 raw bytes
    (void)kind;
    return (size_t)((n_values_total / 8 + n_streams) * 66 + n_streams * 8 + 64);
}
size_t cvec_stream_pack_workspace_bytes(uint64_t n_streams, uint64_t n_values_total) {
    return stream_pack_workspace_bytes(n_streams, n_values_total);
}
cvec_status cvec_stream_pack_dev(int32_t kind, const uint64_t *d_values, const uint64_t *d_elem_off,
                                 uint64_t n_streams, uint64_t n_values_total, uint8_t *d_bytes, size_t bytes_cap,
                                 uint64_t *d_byte_off, int32_t *d_status, uint64_t *d_result, void *d_workspace,
                                 size_t workspace_bytes, void *stream) {
    if (!stream_kind_ok(kind) || !d_elem_off || !d_bytes || !d_byte_off || !d_result || !d_workspace)
        return CVEC_ERR_BAD_ARGUMENT;
    if (n_values_total && !d_values) return CVEC_ERR_BAD_ARGUMENT;
    if (((uintptr_t)d_bytes & 15) != 0) return CVEC_ERR_BAD_ARGUMENT;
    if (bytes_cap < cvec_stream_max_packed_bytes(kind, n_streams, n_values_total)) return CVEC_NOT_ENOUGH_SPACE;
    if (stream_pack_workspace_bytes(n_streams, n_values_total) > workspace_bytes) return CVEC_ERR_WORKSPACE;
    hipError_t e = launch_stream_pack(kind, d_values, d_elem_off, n_streams, n_values_total, d_bytes, d_byte_off,
                                      d_status, d_result, d_workspace, (hipStream_t)stream);
    if (e != hipSuccess) return hip_fail(e, "cvec_stream_pack_dev");
    return CVEC_OK;
}

size_t cvec_stream_unpack_workspace_bytes(uint64_t n_streams, uint64_t n_values_total) {
    return stream_unpack_workspace_bytes(n_streams, n_values_total);
}
cvec_status cvec_stream_unpack_dev(int32_t kind, const uint8_t *d_bytes, const uint64_t *d_byte_off, uint64_t n_bytes,
                                   const uint64_t *d_elem_off, uint64_t n_streams, uint64_t n_values_total,
                                   uint64_t *d_out, int32_t *d_status, uint64_t *d_result, void *d_workspace,
                                   size_t workspace_bytes, void *stream) {
    if (!stream_kind_ok(kind) || !d_byte_off || !d_elem_off || !d_status || !d_result || !d_workspace)
        return CVEC_ERR_BAD_ARGUMENT;
    if ((n_bytes && !d_bytes) || (n_values_total && !d_out)) return CVEC_ERR_BAD_ARGUMENT;
    if (((uintptr_t)d_bytes & 15) != 0 || n_bytes >= (1ull << 40)) return CVEC_ERR_BAD_ARGUMENT;
    if (stream_unpack_workspace_bytes(n_streams, n_values_total) > workspace_bytes) return CVEC_ERR_WORKSPACE;
    hipError_t e = launch_stream_unpack(kind, d_bytes, d_byte_off, n_bytes, d_elem_off, n_streams, n_values_total, d_out,
                                        d_status, d_result, d_workspace, (hipStream_t)stream);
    if (e != hipSuccess) return hip_fail(e, "cvec_stream_unpack_dev");
    return CVEC_OK;
}

size_t cvec_max_encoded_bytes(int32_t etype, uint64_t n_vectors, uint64_t n_elements_total) {
    // every vector: 16-byte header; every section at most 12 + 32 * (2 + 4W) bytes; a vector of n
    // elements has ceil(n/256) sections, so sum of sections <= n_elements_total/256 + n_vectors.
    const uint64_t per_section = etype == CVEC_U64 ? DELTA_HDR + 32 * 66 : DELTA_HDR + 32 * 34;
    const uint64_t max_sections = n_elements_total / FIXED_LEN + n_vectors;
    return (size_t)(n_vectors * HEADER_BYTES + max_sections * per_section + 64);
}

size_t cvec_encode_workspace_bytes(uint64_t n_vectors, uint64_t n_elements_total) {
    return encode_workspace_bytes(n_vectors, n_elements_total);
}

cvec_status cvec_encode_dev(int32_t etype, const void *d_values, const uint64_t *d_elem_off, uint64_t n_vectors,
                            uint64_t n_elements_total, uint8_t *d_bytes, size_t bytes_cap, uint64_t *d_byte_off,
                            uint64_t *d_sec_base, uint64_t *d_result, void *d_workspace, size_t workspace_bytes,
                            void *stream) {
    if (!etype_ok(etype) || !d_elem_off || !d_bytes || !d_byte_off || !d_sec_base || !d_result || !d_workspace)
        return CVEC_ERR_BAD_ARGUMENT;
    if (n_elements_total && !d_values) return CVEC_ERR_BAD_ARGUMENT;
    if (((uintptr_t)d_bytes & 15) != 0) return CVEC_ERR_BAD_ARGUMENT;
    if (bytes_cap < cvec_max_encoded_bytes(etype, n_vectors, n_elements_total)) return CVEC_NOT_ENOUGH_SPACE;
    if (workspace_bytes < encode_workspace_bytes(n_vectors, n_elements_total)) return CVEC_ERR_WORKSPACE;
    hipError_t e = launch_encode(etype, d_values, d_elem_off, n_vectors, n_elements_total, d_bytes, bytes_cap,
                                 d_byte_off, d_sec_base, d_result, d_workspace, workspace_bytes, (hipStream_t)stream);
    if (e != hipSuccess) return hip_fail(e, "cvec_encode_dev");
    return CVEC_OK;
}

// ---- host-buffer convenience -------------------------------------------------------------------


cvec_status cvec_filter_count_batch(int32_t etype, const uint8_t *h_bytes, const uint64_t *h_byte_off,
                                    uint64_t n_vectors, const cvec_pred *pred, uint64_t *h_total,
                                    uint64_t *h_per_vector, int32_t *h_status) {
    if (!h_total) return CVEC_ERR_BAD_ARGUMENT;
    DevBuf bytes, boff, sbase, ws, res, per, stt;
    cvec_dbatch_view view;
    cvec_status st = upload_batch(etype, h_bytes, h_byte_off, n_vectors, bytes, boff, sbase, &view);
    if (st) return st;
    const size_t wsb = cvec_workspace_bytes(view.n_vectors, view.n_sections);
    HIP_TRY(ws.alloc(wsb));
    HIP_TRY(res.alloc(CVEC_RESULT_WORDS * 8));
    if (h_per_vector) HIP_TRY(per.alloc(n_vectors * 8));
    if (h_status) HIP_TRY(stt.alloc(n_vectors * 4));
    st = cvec_filter_count_dev(&view, pred, res.as<uint64_t>(), h_per_vector ? per.as<uint64_t>() : nullptr,
                               h_status ? stt.as<int32_t>() : nullptr, nullptr, ws.p, wsb, nullptr);
    if (st) return st;
    uint64_t r[CVEC_RESULT_WORDS];
    HIP_TRY(hipMemcpy(r, res.p, sizeof(r), hipMemcpyDeviceToHost));
    if (h_per_vector && n_vectors) HIP_TRY(hipMemcpy(h_per_vector, per.p, n_vectors * 8, hipMemcpyDeviceToHost));
    if (h_status && n_vectors) HIP_TRY(hipMemcpy(h_status, stt.p, n_vectors * 4, hipMemcpyDeviceToHost));
    *h_total = r[0];
    return summary_status(r);
}

cvec_status cvec_decode_batch(int32_t etype, const uint8_t *h_bytes, const uint64_t *h_byte_off, uint64_t n_vectors,
                              const cvec_sink_chain *chain, void *h_out, uint64_t out_cap_elems, uint64_t *h_out_off,
                              int32_t *h_status) {
    DevBuf bytes, boff, sbase, ws, res, out, stt;
    cvec_dbatch_view view;
    cvec_status st = upload_batch(etype, h_bytes, h_byte_off, n_vectors, bytes, boff, sbase, &view);
    if (st) return st;
    const uint64_t n_out = view.n_sections * FIXED_LEN;
    if (h_out_off) {
        std::vector<uint64_t> sb(n_vectors + 1);
        HIP_TRY(hipMemcpy(sb.data(), sbase.p, (n_vectors + 1) * 8, hipMemcpyDeviceToHost));
        for (uint64_t v = 0; v <= n_vectors; v++) h_out_off[v] = sb[v] * FIXED_LEN;
    }
    if (n_out > out_cap_elems || (n_out && !h_out)) return CVEC_NOT_ENOUGH_SPACE;
    const size_t wsb = cvec_workspace_bytes(view.n_vectors, view.n_sections);
    const int w = elem_width(etype);
    HIP_TRY(ws.alloc(wsb));
    HIP_TRY(res.alloc(CVEC_RESULT_WORDS * 8));
    HIP_TRY(out.alloc(n_out * w));
    if (h_status) HIP_TRY(stt.alloc(n_vectors * 4));
    HIP_TRY(hipMemset(out.p, 0, n_out * w));
    st = cvec_decode_dev(&view, chain, out.p, res.as<uint64_t>(), h_status ? stt.as<int32_t>() : nullptr, ws.p, wsb,
                         nullptr);
    if (st) return st;
    uint64_t r[CVEC_RESULT_WORDS];
    HIP_TRY(hipMemcpy(r, res.p, sizeof(r), hipMemcpyDeviceToHost));
    if (n_out) HIP_TRY(hipMemcpy(h_out, out.p, n_out * w, hipMemcpyDeviceToHost));
    if (h_status && n_vectors) HIP_TRY(hipMemcpy(h_status, stt.p, n_vectors * 4, hipMemcpyDeviceToHost));
    return summary_status(r);
}

cvec_status cvec_encode_batch(int32_t etype, const void *h_values, const uint64_t *h_elem_off, uint64_t n_vectors,
                              uint8_t *h_bytes, size_t bytes_cap, uint64_t *h_byte_off, uint64_t *h_total_bytes) {
    if (!etype_ok(etype) || !h_elem_off || !h_byte_off || !h_bytes) return CVEC_ERR_BAD_ARGUMENT;
    for (uint64_t v = 0; v < n_vectors; v++) {
        if (h_elem_off[v + 1] < h_elem_off[v]) return CVEC_ERR_BAD_ARGUMENT;
        // finish(): total_num_rows > u32::MAX is InvalidNumRows (reference src/vector.rs:347-349)
        if (h_elem_off[v + 1] - h_elem_off[v] > 0xffffffffull) return CVEC_INVALID_NUM_ROWS;
    }
    const uint64_t n_el = h_elem_off[n_vectors] - h_elem_off[0];
    const int w = elem_width(etype);
    DevBuf vals, eoff, bytes, boff, sbase, ws, res;
    const size_t cap = cvec_max_encoded_bytes(etype, n_vectors, n_el);
    const size_t wsb = cvec_encode_workspace_bytes(n_vectors, n_el);
    HIP_TRY(vals.alloc(n_el * w + 16));
    HIP_TRY(eoff.alloc((n_vectors + 1) * 8));
    HIP_TRY(bytes.alloc(align_up(cap, 256)));
    HIP_TRY(boff.alloc((n_vectors + 1) * 8));
    HIP_TRY(sbase.alloc((n_vectors + 1) * 8));
    HIP_TRY(ws.alloc(wsb));
    HIP_TRY(res.alloc(CVEC_RESULT_WORDS * 8));
    std::vector<uint64_t> eo(n_vectors + 1);
    for (uint64_t v = 0; v <= n_vectors; v++) eo[v] = h_elem_off[v] - h_elem_off[0];
    if (n_el) HIP_TRY(hipMemcpy(vals.p, (const uint8_t *)h_values + h_elem_off[0] * w, n_el * w, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(eoff.p, eo.data(), (n_vectors + 1) * 8, hipMemcpyHostToDevice));
    cvec_status st = cvec_encode_dev(etype, vals.p, eoff.as<uint64_t>(), n_vectors, n_el, bytes.as<uint8_t>(), cap,
                                     boff.as<uint64_t>(), sbase.as<uint64_t>(), res.as<uint64_t>(), ws.p, wsb, nullptr);
    if (st) return st;
    uint64_t r[CVEC_RESULT_WORDS];
    HIP_TRY(hipMemcpy(r, res.p, sizeof(r), hipMemcpyDeviceToHost));
    if (h_total_bytes) *h_total_bytes = r[0];
    if (r[0] > bytes_cap) return CVEC_NOT_ENOUGH_SPACE;
    if (r[0]) HIP_TRY(hipMemcpy(h_bytes, bytes.p, r[0], hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(h_byte_off, boff.p, (n_vectors + 1) * 8, hipMemcpyDeviceToHost));
    return CVEC_OK;
}

} // extern "C"
