// cvec_appender.hip — VectorAppender<T, W> (reference src/vector.rs:207-379) behind the C ABI.
//
// One vector is far below the granularity of a GPU launch, so the handle lives on the host.  Like the reference's
// appender it holds ENCODED sections, not the values that went into them (src/vector.rs:270-281: a section is
// encoded when it fills, the buffer holds compressed bytes): complete sections wait as raw values only until
// FLUSH_SECTIONS of them have gathered, then that run is encoded on the GPU in one call (cvec_encode_batch) and only
// its bytes are kept.  Memory held = the encoded bytes + at most FLUSH_SECTIONS raw sections + the open section (the
// reference's write_buf) + the runs of explicit null sections (src/vector.rs:321-327) as counts.  finish() and
// reader() encode what is still raw, put the null sections' bytes between the value sections and write the header
// (FixedSectStats, src/vector.rs:160-169), so the bytes equal the reference appender's.
#include <cstring>
#include <new>
#include <vector>

#include "cvec.h"

namespace cvec {
void record_message(const char *msg); // cvec_api.hip: the text behind cvec_last_hip_error()
}

namespace {
constexpr size_t FIXED = 256;
constexpr size_t HEADER = 16;
constexpr size_t FLUSH_SECTIONS = 1024; // complete sections that may wait as raw values (1 MiB of u32 / f32)
inline size_t width(int32_t etype) { return etype == CVEC_U64 ? 8 : 4; }
} // namespace

struct cvec_appender {
    int32_t etype;
    size_t initial_capacity;
    // Sections are independent of their neighbours, so the handle keeps a vector as a list of runs: runs of
    // sections that hold appended values (nulls among them as zeros) and runs of explicit null sections
    // (src/vector.rs:320-327), which are a count and nothing else - a sparse vector costs what its values cost,
    // as in the reference.  The value sections, in order: first those already encoded (`encoded`, their bytes back
    // to back), then those still raw (`pending`, 256 elements each).
    struct Run { bool null; uint64_t sections; };
    std::vector<Run> runs;
    std::vector<uint8_t> encoded; // bytes of the value sections encoded so far
    std::vector<uint8_t> pending; // element bytes of the complete value sections not encoded yet
    std::vector<uint8_t> open;    // the open section (the reference's write_buf), fewer than 256 elements
    uint64_t len;                 // elements appended so far, nulls included
    uint32_t null_sections;       // explicit null sections written by append_nulls
    std::vector<uint8_t> enc;     // scratch of one encode call
    std::vector<uint8_t> tail;    // scratch: the raw sections of a finish / reader call (pending + the padded open one)
};

static void appender_clear(cvec_appender *a) {
    a->runs.clear();
    a->encoded.clear();
    a->pending.clear();
    a->open.clear();
    a->len = 0;
    a->null_sections = 0;
}

static void add_run(std::vector<cvec_appender::Run> &runs, bool null, uint64_t sections) {
    if (sections == 0) return;
    if (!runs.empty() && runs.back().null == null) runs.back().sections += sections;
    else runs.push_back({null, sections});
}

// Encodes n_sections raw sections (256 elements each) as one vector into a->enc; the section bytes are
// a->enc[HEADER, *total).
static cvec_status encode_sections(cvec_appender *a, const uint8_t *vals, uint64_t n_sections, uint64_t *total) {
    const uint64_t n_el = n_sections * FIXED;
    const uint64_t elem_off[2] = {0, n_el};
    uint64_t byte_off[2] = {0, 0};
    a->enc.resize(cvec_max_encoded_bytes(a->etype, 1, n_el) + 64);
    const uint64_t dummy = 0;
    cvec_status st = cvec_encode_batch(a->etype, n_el ? (const void *)vals : (const void *)&dummy, elem_off, 1, a->enc.data(),
                                       a->enc.size(), byte_off, total);
    if (st) return st;
    return *total < HEADER ? CVEC_ERR_HIP : CVEC_OK;
}

// The raw complete sections become encoded bytes.  A failure (no device, ...) leaves them raw: nothing is lost, the
// handle only holds more memory, and finish() / reader() report the error when they need the encoder themselves.
static void flush_pending(cvec_appender *a) {
    const size_t w = width(a->etype);
    const uint64_t n_sections = a->pending.size() / (FIXED * w);
    if (n_sections == 0) return;
    uint64_t total = 0;
    if (encode_sections(a, a->pending.data(), n_sections, &total) != CVEC_OK) return;
    a->encoded.insert(a->encoded.end(), a->enc.begin() + HEADER, a->enc.begin() + (size_t)total);
    a->pending.clear();
}

// The open section has filled up: it becomes a complete value section (encode_section, src/vector.rs:270-281).
static void close_open(cvec_appender *a) {
    a->pending.insert(a->pending.end(), a->open.begin(), a->open.end());
    a->open.clear();
    add_run(a->runs, false, 1);
    if (a->pending.size() >= FLUSH_SECTIONS * FIXED * width(a->etype)) flush_pending(a);
}

// Size of the section at p (src/section.rs:252-271 and the num_bytes of each kind); 0 if it does not fit in `left`.
static size_t section_bytes(const uint8_t *p, size_t left, size_t w) {
    if (left < 1) return 0;
    size_t n = 0;
    switch (p[0]) {
    case 0x00: n = 1; break;                                                           // Null
    case 0x05: n = 1 + w; break;                                                       // Const
    case 0x01: if (left < 3) return 0; n = 3 + (size_t)(p[1] | (p[2] << 8)); break;    // NibblePack
    case 0x03: if (left < 3) return 0; n = 12 + (size_t)(p[1] | (p[2] << 8)); break;   // Delta
    case 0x06: if (left < 3) return 0; n = (size_t)(p[1] | (p[2] << 8)); break;        // Xor: total incl. header
    default: return 0;
    }
    return n != 0 && n <= left ? n : 0;
}

// Writes the vector made of `runs`: its value sections are a->encoded followed by the encoding of `raw_sections`
// raw sections at `raw`; *len receives the size, the bytes go to out if it is large enough.
static cvec_status appender_emit(cvec_appender *a, const std::vector<cvec_appender::Run> &runs, const uint8_t *raw,
                                 uint64_t raw_sections, uint64_t num_elements, uint32_t nulls, uint8_t *out, size_t cap,
                                 size_t *len) {
    const size_t w = width(a->etype);
    uint64_t total = HEADER;
    if (raw_sections) {
        cvec_status st = encode_sections(a, raw, raw_sections, &total);
        if (st) return st;
    }
    uint64_t null_bytes = 0;
    for (const auto &r : runs) if (r.null) null_bytes += r.sections;
    const uint64_t size = HEADER + a->encoded.size() + (total - HEADER) + null_bytes;
    if (len) *len = (size_t)size;
    if (out == nullptr || cap < size) return CVEC_ERR_CAPACITY;
    const uint32_t num_bytes = (uint32_t)(size - 4); // BinaryVector.num_bytes (src/vector.rs:54, :139)
    memset(out, 0, HEADER);
    memcpy(out, &num_bytes, 4);
    out[4] = 0x10;                 // VectorType::FixedSection256
    out[5] = (uint8_t)a->etype;    // VectorSubType
    const uint32_t ne = (uint32_t)num_elements;
    memcpy(out + 8, &ne, 4);       // FixedSectStats (src/vector.rs:163-186)
    out[12] = (uint8_t)(nulls & 0xffu);
    out[13] = (uint8_t)((nulls >> 8) & 0xffu);
    // the value sections come from two pieces, in order; a section never straddles them
    const uint8_t *src = a->encoded.data(), *src_end = src + a->encoded.size();
    const uint8_t *next = a->enc.data() + HEADER, *next_end = a->enc.data() + (raw_sections ? (size_t)total : HEADER);
    uint8_t *dst = out + HEADER;
    for (const auto &r : runs) {
        if (r.null) {
            memset(dst, 0, (size_t)r.sections); // NullFixedSect: the byte 0x00 (src/section.rs:440-443)
            dst += r.sections;
            continue;
        }
        for (uint64_t k = 0; k < r.sections; k++) {
            if (src == src_end && next != nullptr) { src = next; src_end = next_end; next = nullptr; }
            const size_t n = section_bytes(src, (size_t)(src_end - src), w);
            if (n == 0) return CVEC_ERR_HIP;
            memcpy(dst, src, n);
            src += n;
            dst += n;
        }
    }
    if (src == src_end && next != nullptr) { src = next; src_end = next_end; }
    if (src != src_end || dst != out + size) return CVEC_ERR_HIP;
    return CVEC_OK;
}

// No exception leaves the library: the containers of the handle can run out of memory.
#define CVEC_NO_THROW(...)                                                           \
    try {                                                                            \
        __VA_ARGS__                                                                  \
    } catch (const std::bad_alloc &) {                                               \
        cvec::record_message("cvec_appender: out of host memory");                   \
        return CVEC_ERR_HIP;                                                         \
    }

extern "C" {

cvec_status cvec_appender_new(int32_t etype, size_t initial_capacity, cvec_appender **out) {
    if (!out || (etype != CVEC_U32 && etype != CVEC_U64 && etype != CVEC_F32)) return CVEC_ERR_BAD_ARGUMENT;
    *out = nullptr;
    // try_new writes the 8-byte BinaryVector header into a buffer of initial_capacity bytes
    // (src/vector.rs:223-234, :128-131): scroll reports a buffer that cannot hold it
    if (initial_capacity < 8) return CVEC_NOT_ENOUGH_SPACE;
    cvec_appender *a = new (std::nothrow) cvec_appender();
    if (!a) { cvec::record_message("cvec_appender_new: out of host memory"); return CVEC_ERR_HIP; }
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
    CVEC_NO_THROW(
    const size_t w = width(a->etype);
    const uint8_t *p = (const uint8_t *)values;
    size_t left = n;
    while (left) {
        const size_t room = FIXED - a->open.size() / w;
        const size_t take = left < room ? left : room;
        a->open.insert(a->open.end(), p, p + take * w);
        p += take * w;
        left -= take;
        a->len += take;
        if (a->open.size() == FIXED * w) close_open(a);
    }
    return CVEC_OK;)
}

cvec_status cvec_appender_append_nulls(cvec_appender *a, size_t num_nulls) {
    if (!a) return CVEC_ERR_BAD_ARGUMENT;
    // src/vector.rs:311-335: zeros first fill the open section; whole runs of 256 behind an empty write buffer
    // become explicit null sections, counted in the header; what is left opens the next section
    CVEC_NO_THROW(
    const size_t w = width(a->etype);
    size_t left = num_nulls;
    if (left && !a->open.empty()) {
        const size_t room = FIXED - a->open.size() / w;
        const size_t fill = left < room ? left : room;
        a->open.insert(a->open.end(), fill * w, (uint8_t)0);
        left -= fill;
        a->len += fill;
        if (a->open.size() == FIXED * w) close_open(a);
    }
    if (a->open.empty()) {
        a->open.reserve((left % FIXED) * w); // before anything is counted: a failure leaves the handle consistent
        add_run(a->runs, true, left / FIXED);
        a->null_sections += (uint32_t)(left / FIXED);
        a->open.insert(a->open.end(), (left % FIXED) * w, (uint8_t)0);
        a->len += left;
    }
    return CVEC_OK;)
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
    // :352-359: the open section is rounded out with zeros, then null sections follow up to total_num_rows.
    // Built beside the appender's state: a call that fails (out too small) leaves the appender as it was (complete
    // sections that were still raw may have become encoded bytes, which changes nothing a caller can see), so the
    // usual pair of calls - one for the size, one with the buffer - encodes only the open section twice.
    CVEC_NO_THROW(
    const size_t w = width(a->etype);
    flush_pending(a);
    std::vector<cvec_appender::Run> runs = a->runs;
    a->tail = a->pending; // empty unless the flush failed
    uint64_t raw_sections = a->tail.size() / (FIXED * w);
    if (!a->open.empty()) {
        a->tail.insert(a->tail.end(), a->open.begin(), a->open.end());
        a->tail.resize((size_t)(raw_sections + 1) * FIXED * w, 0);
        raw_sections += 1;
        add_run(runs, false, 1);
    }
    const uint64_t padded = (a->len + FIXED - 1) / FIXED * FIXED;
    uint32_t nulls = a->null_sections;
    if (total_num_rows > padded) {
        const uint64_t more = (total_num_rows - padded + FIXED - 1) / FIXED;
        nulls += (uint32_t)more;
        add_run(runs, true, more);
    }
    cvec_status st = appender_emit(a, runs, a->tail.data(), raw_sections, total_num_rows, nulls, out, cap, len);
    if (st == CVEC_OK) appender_clear(a); // :368 reset(); a too-small `out` leaves the appender as it was
    return st;)
}

cvec_status cvec_appender_reader(cvec_appender *a, uint8_t *out, size_t cap, size_t *len) {
    if (!a) return CVEC_ERR_BAD_ARGUMENT;
    CVEC_NO_THROW(
    const size_t w = width(a->etype);
    flush_pending(a);
    const uint64_t flushed = a->len - a->open.size() / w; // values in the open section are not in the vector yet (:372-378)
    return appender_emit(a, a->runs, a->pending.data(), a->pending.size() / (FIXED * w), flushed, a->null_sections, out, cap,
                         len);)
}

cvec_status cvec_appender_stats(const cvec_appender *a, uint64_t *encoded_bytes, uint64_t *raw_value_bytes) {
    if (!a) return CVEC_ERR_BAD_ARGUMENT;
    if (encoded_bytes) *encoded_bytes = a->encoded.size();
    if (raw_value_bytes) *raw_value_bytes = a->pending.size() + a->open.size();
    return CVEC_OK;
}

} // extern "C"
