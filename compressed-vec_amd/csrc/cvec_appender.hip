// cvec_appender.hip — VectorAppender<T, W> (reference src/vector.rs:207-379) behind the C ABI.
//
// One vector is far below the granularity of a GPU launch, so the handle keeps what the reference keeps in
// `write_buf` / `stats` on the host: the values appended so far, how many of them sit in the open (not yet
// complete) section, and the number of explicit null sections (src/vector.rs:321-327).  finish() and reader()
// encode on the GPU through cvec_encode_batch and write the null-section count into the vector header
// (offset 12, FixedSectStats, src/vector.rs:160-169), so the bytes equal the reference appender's.
#include <cstring>
#include <new>
#include <vector>

#include "cvec.h"

namespace {
constexpr size_t FIXED = 256;
constexpr size_t HEADER = 16;
inline size_t width(int32_t etype) { return etype == CVEC_U64 ? 8 : 4; }
} // namespace

struct cvec_appender {
    int32_t etype;
    size_t initial_capacity;
    std::vector<uint8_t> values; // element bytes, nulls as zeros
    uint64_t len;                // elements appended so far, nulls included
    uint32_t open;               // elements in the open section (the reference's write_buf.len())
    uint32_t null_sections;      // explicit null sections written by append_nulls
    std::vector<uint8_t> enc;    // scratch for the encoded vector
};

static void appender_clear(cvec_appender *a) {
    a->values.clear();
    a->len = 0;
    a->open = 0;
    a->null_sections = 0;
}

// Encodes the first n_elems values, padded with zeros to total elements, as one vector whose header counts
// `nulls` explicit null sections; *len receives the size, the bytes go to out if it is large enough.
static cvec_status appender_encode(cvec_appender *a, uint64_t n_elems, uint64_t total, uint32_t nulls, uint8_t *out,
                                   size_t cap, size_t *len) {
    const size_t w = width(a->etype);
    std::vector<uint8_t> padded;
    const uint8_t *vals = a->values.data();
    if (total > n_elems) {
        padded.assign((size_t)total * w, 0);
        memcpy(padded.data(), a->values.data(), (size_t)n_elems * w);
        vals = padded.data();
    }
    const uint64_t elem_off[2] = {0, total};
    uint64_t byte_off[2] = {0, 0}, total_bytes = 0;
    a->enc.resize(cvec_max_encoded_bytes(a->etype, 1, total) + 64);
    const uint64_t dummy = 0;
    cvec_status st = cvec_encode_batch(a->etype, total ? (const void *)vals : (const void *)&dummy, elem_off, 1, a->enc.data(),
                                       a->enc.size(), byte_off, &total_bytes);
    if (st) return st;
    if (total_bytes < HEADER) return CVEC_ERR_HIP;
    a->enc[12] = (uint8_t)(nulls & 0xffu);
    a->enc[13] = (uint8_t)((nulls >> 8) & 0xffu);
    if (len) *len = (size_t)total_bytes;
    if (out == nullptr || cap < total_bytes) return CVEC_ERR_CAPACITY;
    memcpy(out, a->enc.data(), (size_t)total_bytes);
    return CVEC_OK;
}

extern "C" {

cvec_status cvec_appender_new(int32_t etype, size_t initial_capacity, cvec_appender **out) {
    if (!out || (etype != CVEC_U32 && etype != CVEC_U64 && etype != CVEC_F32)) return CVEC_ERR_BAD_ARGUMENT;
    *out = nullptr;
    // try_new writes the 8-byte BinaryVector header into a buffer of initial_capacity bytes
    // (src/vector.rs:223-234, :128-131): scroll reports a buffer that cannot hold it
    if (initial_capacity < 8) return CVEC_NOT_ENOUGH_SPACE;
    cvec_appender *a = new (std::nothrow) cvec_appender();
    if (!a) return CVEC_ERR_HIP;
    a->etype = etype;
    a->initial_capacity = initial_capacity;
    appender_clear(a);
    *out = a;
    return CVEC_OK;
}

void cvec_appender_free(cvec_appender *a) { delete a; }

cvec_status cvec_appender_reset(cvec_appender *a) {
    if (!a) return CVEC_ERR_BAD_ARGUMENT;
    appender_clear(a);
    return CVEC_OK;
}

uint64_t cvec_appender_num_elements(const cvec_appender *a) { return a ? a->len : 0; }

cvec_status cvec_appender_append(cvec_appender *a, const void *values, size_t n) {
    if (!a || (n && !values)) return CVEC_ERR_BAD_ARGUMENT;
    const size_t w = width(a->etype);
    const uint8_t *p = (const uint8_t *)values;
    a->values.insert(a->values.end(), p, p + n * w);
    a->len += n;
    a->open = (uint32_t)((a->open + n) % FIXED);
    return CVEC_OK;
}

cvec_status cvec_appender_append_nulls(cvec_appender *a, size_t num_nulls) {
    if (!a) return CVEC_ERR_BAD_ARGUMENT;
    // src/vector.rs:311-335: zeros first fill the open section; whole runs of 256 behind an empty write buffer
    // become explicit null sections, counted in the header; what is left opens the next section
    size_t left = num_nulls;
    if (left && a->open) {
        const size_t fill = left < FIXED - a->open ? left : FIXED - a->open;
        a->open = (uint32_t)((a->open + fill) % FIXED);
        left -= fill;
    }
    if (a->open == 0) {
        a->null_sections += (uint32_t)(left / FIXED);
        a->open = (uint32_t)(left % FIXED);
    }
    a->values.insert(a->values.end(), num_nulls * width(a->etype), (uint8_t)0);
    a->len += num_nulls;
    return CVEC_OK;
}

cvec_status cvec_appender_finish(cvec_appender *a, uint64_t total_num_rows, uint8_t *out, size_t cap, size_t *len,
                                 uint64_t detail[2]) {
    if (!a) return CVEC_ERR_BAD_ARGUMENT;
    if (detail) detail[0] = detail[1] = 0;
    // src/vector.rs:344-349
    if (a->len > total_num_rows) {
        if (detail) { detail[0] = total_num_rows; detail[1] = a->len; }
        return CVEC_INVALID_NUM_ROWS;
    }
    if (total_num_rows > 0xffffffffull) {
        if (detail) { detail[0] = total_num_rows; detail[1] = 0xffffffffull; }
        return CVEC_INVALID_NUM_ROWS;
    }
    // :352-359: the open section is rounded out with zeros, then null sections follow up to total_num_rows
    const uint64_t padded = (a->len + FIXED - 1) / FIXED * FIXED;
    uint32_t nulls = a->null_sections;
    if (total_num_rows > padded) nulls += (uint32_t)((total_num_rows - padded + FIXED - 1) / FIXED);
    cvec_status st = appender_encode(a, a->len, total_num_rows, nulls, out, cap, len);
    if (st == CVEC_OK) appender_clear(a); // :368 reset(); a too-small `out` leaves the appender as it was
    return st;
}

cvec_status cvec_appender_reader(cvec_appender *a, uint8_t *out, size_t cap, size_t *len) {
    if (!a) return CVEC_ERR_BAD_ARGUMENT;
    const uint64_t flushed = a->len - a->open; // values in the open section are not in the vector yet (:372-378)
    return appender_encode(a, flushed, flushed, a->null_sections, out, cap, len);
}

} // extern "C"
