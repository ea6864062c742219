// cvec_series.hip — one VectorAppender<T, W> (reference src/vector.rs:207-379) per series, many series at once,
// resident in HBM.
//
// A scrape hands every series a few new values.  Per series the reference keeps an open section of up to 255 raw
// values (`vect_buf`, src/vector.rs:210, :300-307) and the encoded bytes of the sections that are complete
// (`write_buf`).  Here both live on the device for all n series of the handle:
//   stage[n][256], fill[n]   the open sections (raw values) and how many values each holds
//   len[n], nulls[n]         elements appended (nulls included) and explicit null sections (src/vector.rs:321-327)
//   segments                 every append call that completes sections encodes exactly those sections, for all
//                            series at once, with the batch encoder; the result (one short vector per series, header
//                            included, empty where nothing completed) is kept as a segment at its exact size
//                            A run of whole null sections (append_nulls behind an empty open section, finish's
//                            padding) is one 0x00 byte per section in the vector: such a segment is only a count per
//                            series, whatever the length of the run
// Merging: ragged appends that complete sections for a few series per call would leave one segment (and one offset
// per series) per call.  Segments carry a level; as soon as MERGE_FANIN segments of one level have gathered at the
// end of the list they are gathered into one segment of the next level (the kernels of emit(), aimed at the arena of
// that level), and the arena of the lower level is rewound: a byte is copied once per level, a handle holds at most
// MERGE_FANIN - 1 segments per level, and what it holds is bounded by twice its contents.  A long run of null sections
// (more than MERGE_MAX_NULL_RUN per series on average) is never written out as bytes: everything in front of it is
// merged into one segment that stays where it is, and so does the run.
// seal() adds what finish() adds (the open section padded with zeros as an encoded segment, the null sections up
// to total_num_rows as a count, src/vector.rs:344-359) and computes the byte offsets of the final vectors;
// emit() writes, per series, one header (element and null-section counts, src/vector.rs:160-169) followed by the
// section bytes of all segments in order.  Sections are independent of their neighbours, so the bytes equal the
// reference appender's for the same sequence of calls.
#include <memory>
#include <new>
#include <vector>

#include "cvec_device.hpp"

namespace cvec {
uint64_t scan_scratch_words(uint64_t n);
hipError_t launch_exclusive_scan(const uint64_t *d_in, uint64_t n, uint64_t *d_out, uint64_t *d_scratch, hipStream_t stream);
size_t encode_workspace_bytes(uint64_t n_vectors, uint64_t n_elements_total);
hipError_t launch_encode(int32_t etype, const void *d_values, const uint64_t *d_elem_off, uint64_t n_vectors,
                         uint64_t n_elements_total, uint8_t *d_bytes, size_t bytes_cap, uint64_t *d_byte_off,
                         uint64_t *d_sec_base, uint64_t *d_result, void *d_workspace, size_t workspace_bytes,
                         hipStream_t stream);
cvec_status record_hip_error(hipError_t e, const char *what);
} // namespace cvec

using namespace cvec;

namespace {

constexpr uint32_t SER_THREADS = 256;
constexpr uint32_t SER_WAVES = SER_THREADS / 64;

struct Buf { // grow-only device buffer
    void *p = nullptr;
    size_t cap = 0;
    ~Buf() { if (p) (void)hipFree(p); }
    hipError_t need(size_t n) {
        n = n ? n : 16;
        if (p && cap >= n) return hipSuccess;
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        hipError_t e = hipMalloc(&p, n);
        if (e == hipSuccess) cap = n;
        return e;
    }
    template <class T> T *as() const { return (T *)p; }
};

struct Segment { // the arrays live in the handle's arena
    uint8_t *bytes = nullptr;
    uint64_t *off = nullptr;     // n + 1
    uint64_t *null_run = nullptr; // instead of bytes / off: series v continues with null_run[v] null sections (one 0x00 byte each)
    uint64_t n_bytes = 0;
    uint32_t level = 0;           // how many merges its contents have been through
    bool pinned = false;          // stays as it is: a long run of null sections, or what lies in front of one
};
constexpr uint32_t MERGE_FANIN = 16;        // segments of one level that are merged into one of the next
constexpr uint64_t MERGE_MAX_NULL_RUN = 64; // null sections per series (on average) up to which a run is merged as bytes

// Segment storage: slabs of device memory handed out front to back.  A finish() or reset() rewinds the slabs
// and keeps them, so a handle that is filled again and again allocates only while it grows.
struct Arena {
    struct Slab { uint8_t *p; size_t cap, used; };
    std::vector<Slab> slabs;
    static constexpr size_t SLAB_MIN = (size_t)4 << 20, SLAB_MAX = (size_t)1 << 30;
    ~Arena() { release(); }
    void release() {
        for (Slab &s : slabs) (void)hipFree(s.p);
        slabs.clear();
    }
    void rewind() { for (Slab &s : slabs) s.used = 0; }
    hipError_t take(size_t bytes, void **out) {
        bytes = (bytes + 255) / 256 * 256;
        for (Slab &s : slabs)
            if (s.cap - s.used >= bytes) { *out = s.p + s.used; s.used += bytes; return hipSuccess; }
        size_t cap = 8 * bytes;
        cap = cap < SLAB_MIN ? SLAB_MIN : cap > SLAB_MAX ? SLAB_MAX : cap;
        cap = cap < bytes ? bytes : cap;
        Slab s{nullptr, cap, bytes};
        hipError_t e = hipMalloc((void **)&s.p, cap);
        if (e != hipSuccess && cap > bytes) { s.cap = bytes; e = hipMalloc((void **)&s.p, bytes); } // a tight fit as the last resort
        if (e != hipSuccess) return e;
        slabs.push_back(s);
        *out = s.p;
        return hipSuccess;
    }
    // Room for `bytes` in a slab that exists already, or nullptr: for an allocation whose final size is not
    // known yet (trim() gives back what it did not need).
    void *take_if_room(size_t bytes) {
        bytes = (bytes + 255) / 256 * 256;
        for (Slab &s : slabs)
            if (s.cap - s.used >= bytes) { void *p = s.p + s.used; s.used += bytes; return p; }
        return nullptr;
    }
    // The most recent allocation of its slab shrinks to `bytes`.
    void trim(void *p, size_t bytes) {
        bytes = (bytes + 255) / 256 * 256;
        for (Slab &s : slabs)
            if ((uint8_t *)p >= s.p && (uint8_t *)p < s.p + s.used) { s.used = (size_t)((uint8_t *)p - s.p) + bytes; return; }
    }
    size_t reserved() const { size_t t = 0; for (const Slab &s : slabs) t += s.cap; return t; }
};

// ---- kernels ---------------------------------------------------------------------------------------------

// One lane per series: how many values arrive and how many of open + new leave as complete sections.
__global__ void k_ser_plan(uint64_t n, const uint64_t *__restrict__ elem_off, uint64_t per_vector, int nulls_mode,
                           const uint32_t *__restrict__ fill, uint64_t *__restrict__ len, uint32_t *__restrict__ nulls,
                           uint64_t *__restrict__ flush_cnt, uint64_t *__restrict__ null_run) {
    const uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n) return;
    const uint64_t c = elem_off ? elem_off[v + 1] - elem_off[v] : per_vector;
    const uint32_t f = fill[v];
    uint64_t flush = ((uint64_t)f + c) / FIXED_LEN * FIXED_LEN, run = 0;
    if (nulls_mode) {
        // src/vector.rs:311-335: zeros fill the open section first; whole runs of 256 behind an empty open
        // section are explicit null sections
        uint64_t left = c;
        uint32_t open = f;
        if (left && open) {
            const uint64_t room = FIXED_LEN - open;
            const uint64_t fillup = left < room ? left : room;
            open = (uint32_t)((open + fillup) % FIXED_LEN);
            left -= fillup;
        }
        if (open == 0) nulls[v] += (uint32_t)(left / FIXED_LEN);
        // only the section that the zeros complete goes through the encoder; the whole sections of zeros behind it
        // are one byte each, whatever their number, and are kept as a count
        const uint64_t sections = flush / FIXED_LEN;
        flush = (f != 0 && sections != 0) ? FIXED_LEN : 0;
        run = sections - flush / FIXED_LEN;
    }
    len[v] += c;
    flush_cnt[v] = flush;
    null_run[v] = run;
}

// One wave per series: the first out_off[v+1] - out_off[v] elements of (open section ++ new values ++ zeros).
template <class U>
__global__ __launch_bounds__(SER_THREADS) void k_ser_flush(uint64_t n, const U *__restrict__ values,
                                                           const uint64_t *__restrict__ elem_off, uint64_t per_vector,
                                                           const U *__restrict__ stage, const uint32_t *__restrict__ fill,
                                                           const uint64_t *__restrict__ out_off, U *__restrict__ out) {
    const uint64_t v = (uint64_t)blockIdx.x * SER_WAVES + threadIdx.x / 64;
    const uint32_t lane = threadIdx.x & 63;
    if (v >= n) return;
    const uint64_t o0 = out_off[v], cnt = out_off[v + 1] - o0;
    if (cnt == 0) return;
    const uint32_t f = fill[v];
    uint64_t base = 0, c = 0;
    if (values) {
        base = elem_off ? elem_off[v] : v * per_vector;
        c = elem_off ? elem_off[v + 1] - base : per_vector;
    }
    for (uint64_t i = lane; i < cnt; i += 64) {
        U x = 0;
        if (i < f) x = stage[v * FIXED_LEN + i];
        else if (i - f < c) x = values[base + (i - f)];
        out[o0 + i] = x;
    }
}

// One wave per series: what stays behind in the open section after the complete sections left.
template <class U>
__global__ __launch_bounds__(SER_THREADS) void k_ser_stage(uint64_t n, const U *__restrict__ values,
                                                           const uint64_t *__restrict__ elem_off, uint64_t per_vector,
                                                           U *__restrict__ stage, uint32_t *__restrict__ fill) {
    const uint64_t v = (uint64_t)blockIdx.x * SER_WAVES + threadIdx.x / 64;
    const uint32_t lane = threadIdx.x & 63;
    if (v >= n) return;
    const uint64_t base = elem_off ? elem_off[v] : v * per_vector;
    const uint64_t c = elem_off ? elem_off[v + 1] - base : per_vector;
    const uint32_t f = fill[v];
    const uint64_t tot = f + c, closed = tot / FIXED_LEN * FIXED_LEN;
    const uint32_t nf = (uint32_t)(tot - closed);
    U *row = stage + v * FIXED_LEN;
    if (closed == 0) {
        for (uint64_t i = f + lane; i < tot; i += 64) row[i] = values ? values[base + (i - f)] : (U)0;
    } else {
        // closed >= 256 > f: everything that stays comes from the new values
        for (uint32_t i = lane; i < nf; i += 64) row[i] = values ? values[base + (closed - f) + i] : (U)0;
    }
    if (lane == 0 && c) fill[v] = nf;
}

// The same for a scrape of a few values per series: one lane per series (a wave per series would move one value).
template <class U>
__global__ void k_ser_stage_few(uint64_t n, const U *__restrict__ values, uint64_t per_vector, U *__restrict__ stage,
                                uint32_t *__restrict__ fill) {
    const uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n) return;
    const uint64_t base = v * per_vector, c = per_vector;
    const uint32_t f = fill[v];
    const uint64_t tot = f + c, closed = tot / FIXED_LEN * FIXED_LEN;
    const uint32_t nf = (uint32_t)(tot - closed);
    U *row = stage + v * FIXED_LEN;
    if (closed == 0) {
        for (uint64_t i = f; i < tot; i++) row[i] = values ? values[base + (i - f)] : (U)0;
    } else {
        for (uint32_t i = 0; i < nf; i++) row[i] = values ? values[base + (closed - f) + i] : (U)0;
    }
    fill[v] = nf;
}

__global__ void k_ser_max_len(uint64_t n, const uint64_t *__restrict__ len, unsigned long long *__restrict__ out) {
    const uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t x = v < n ? len[v] : 0;
    for (int d = 32; d; d >>= 1) {
        const uint64_t y = __shfl_xor(x, d);
        x = y > x ? y : x;
    }
    // one atomic per workgroup: atomics on one address complete one after the other
    __shared__ uint64_t s_max[SER_WAVES];
    if ((threadIdx.x & 63) == 0) s_max[threadIdx.x >> 6] = x;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (uint32_t w = 1; w < SER_WAVES; w++) x = s_max[w] > x ? s_max[w] : x;
        if (x) atomicMax(out, (unsigned long long)x);
    }
}

// One lane per series: elements of the closing segment (open section + zeros up to total_rows).
__global__ void k_ser_tail_plan(uint64_t n, uint64_t total_rows, const uint64_t *__restrict__ len,
                                const uint32_t *__restrict__ fill, uint64_t *__restrict__ tail_cnt,
                                uint64_t *__restrict__ null_run) {
    const uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n) return;
    // src/vector.rs:352-359: the open section is rounded out with zeros (the encoder pads a short last section),
    // then null sections follow until total_rows is covered
    const uint32_t f = fill[v];
    const uint64_t padded = len[v] - f + (f ? FIXED_LEN : 0);
    tail_cnt[v] = f;
    null_run[v] = total_rows > padded ? (total_rows - padded + FIXED_LEN - 1) / FIXED_LEN : 0;
}

// One lane per series: size of the final vector and its number of sections.
__global__ void k_ser_sizes(uint64_t n, int reader, uint64_t total_rows, const uint64_t *__restrict__ len,
                            const uint32_t *__restrict__ fill, const uint64_t *const *__restrict__ seg_off,
                            const uint64_t *const *__restrict__ seg_null, uint32_t n_seg,
                            uint64_t *__restrict__ out_size, uint64_t *__restrict__ out_nsec) {
    const uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n) return;
    uint64_t size = HEADER_BYTES;
    for (uint32_t s = 0; s < n_seg; s++) {
        const uint64_t *o = seg_off[s];
        size += o ? o[v + 1] - o[v] - HEADER_BYTES : seg_null[s][v];
    }
    const uint64_t elems = reader ? len[v] - fill[v] : total_rows;
    out_size[v] = size;
    out_nsec[v] = (elems + FIXED_LEN - 1) / FIXED_LEN;
}

// One wave per series: header, then the section bytes of every segment in order.
__global__ __launch_bounds__(SER_THREADS) void k_ser_gather(uint64_t n, int etype, int reader, uint64_t total_rows,
                                                            const uint64_t *__restrict__ len, const uint32_t *__restrict__ fill,
                                                            const uint32_t *__restrict__ nulls,
                                                            const uint64_t *const *__restrict__ seg_off,
                                                            const uint8_t *const *__restrict__ seg_bytes,
                                                            const uint64_t *const *__restrict__ seg_null, uint32_t n_seg,
                                                            const uint64_t *__restrict__ byte_off, uint8_t *__restrict__ bytes) {
    const uint64_t v = (uint64_t)blockIdx.x * SER_WAVES + threadIdx.x / 64;
    const uint32_t lane = threadIdx.x & 63;
    if (v >= n) return;
    const uint64_t start = byte_off[v], end = byte_off[v + 1];
    uint8_t *dst = bytes + start;
    if (lane < HEADER_BYTES) {
        const uint64_t l = len[v];
        const uint64_t elems = reader ? l - fill[v] : total_rows;
        uint32_t nn = nulls[v];
        if (!reader) {
            // src/vector.rs:352-359: the open section is rounded out, then null sections up to total_rows
            const uint64_t padded = (l + FIXED_LEN - 1) / FIXED_LEN * FIXED_LEN;
            if (total_rows > padded) nn += (uint32_t)((total_rows - padded + FIXED_LEN - 1) / FIXED_LEN);
        }
        const uint32_t num_bytes = (uint32_t)(end - start - 4); // BinaryVector.num_bytes (src/vector.rs:54, :139)
        uint8_t b = 0;
        if (lane < 4) b = (uint8_t)(num_bytes >> (8 * lane));
        else if (lane == 4) b = 0x10;                           // VectorType::FixedSection256
        else if (lane == 5) b = (uint8_t)etype;                 // VectorSubType
        else if (lane >= 8 && lane < 12) b = (uint8_t)((uint32_t)elems >> (8 * (lane - 8)));
        else if (lane == 12) b = (uint8_t)(nn & 0xff);          // FixedSectStats.num_null_sections (src/vector.rs:163-186)
        else if (lane == 13) b = (uint8_t)((nn >> 8) & 0xff);
        dst[lane] = b;
    }
    dst += HEADER_BYTES;
    for (uint32_t s = 0; s < n_seg; s++) {
        const uint64_t *o = seg_off[s];
        if (o == nullptr) { // a run of null sections: NullFixedSect is the one byte 0x00 (src/section.rs:440-443)
            const uint64_t m = seg_null[s][v];
            for (uint64_t i = lane; i < m; i += 64) dst[i] = 0;
            dst += m;
            continue;
        }
        const uint64_t s0 = o[v] + HEADER_BYTES, s1 = o[v + 1];
        const uint8_t *src = seg_bytes[s] + s0;
        const uint64_t m = s1 - s0;
        // sections start at any byte: bytes up to the destination's next 16-byte boundary, whole 16-byte
        // pieces (aligned stores, loads at whatever phase the source has), bytes after the last piece
        uint64_t head = (16 - ((uintptr_t)dst & 15)) & 15;
        head = head < m ? head : m;
        if (lane < head) dst[lane] = src[lane];
        const uint64_t pieces = (m - head) / 16;
        for (uint64_t j = lane; j < pieces; j += 64) {
            const uint8_t *sp = src + head + 16 * j;
            uint4 x;
            x.x = *(const u32u *)(sp + 0);
            x.y = *(const u32u *)(sp + 4);
            x.z = *(const u32u *)(sp + 8);
            x.w = *(const u32u *)(sp + 12);
            *(uint4 *)(dst + head + 16 * j) = x;
        }
        const uint64_t done = head + 16 * pieces;
        if (lane < m - done) dst[done + lane] = src[done + lane];
        dst += m;
    }
}

inline uint32_t lane_blocks(uint64_t n) { return (uint32_t)((n + SER_THREADS - 1) / SER_THREADS); }
inline uint32_t wave_blocks(uint64_t n) { return (uint32_t)((n + SER_WAVES - 1) / SER_WAVES); }

} // namespace

struct cvec_series_appender {
    int32_t etype = 0;
    int device = 0;
    uint64_t n = 0;
    uint32_t width = 4;
    Buf stage, fill, len, nulls;         // per series state
    Buf cnt, off, scan_tmp, vals;        // per call: elements leaving per series, their scan, the values
    Buf run, run_scan;                   // per call: null sections per series that are kept as a count, their scan
    Buf enc_bytes, enc_off, enc_sec, enc_res, enc_ws;
    Buf out_size, out_off, out_nsec, out_sec, tab_off, tab_bytes, tab_null, maxlen;
    Buf up_vals, up_off;                 // device copies of host arguments (the *_host calls)
    Arena arena;                         // segments of level 0, the closing segments of a seal
    std::vector<std::unique_ptr<Arena>> upper; // upper[l - 1]: the merged segments of level l
    Arena pinned;                        // segments that are never merged again
    uint64_t merges = 0;                 // merge operations since the handle was made
    std::vector<Segment> segs;
    std::vector<Segment> tail;           // finish()'s closing segments (open section, null run), between seal and emit
    bool uniform = true;                 // every call since the last reset gave all series the same count:
    uint32_t ufill = 0;                  // ... then all open sections hold ufill values and the host needs no read-back
    int sealed = 0;                      // 0 open, 1 sealed for finish, 2 sealed for reader
    uint64_t sealed_rows = 0, sealed_bytes = 0, sealed_sections = 0;
    uint64_t resident_bytes = 0;

};

#define SER_TRY(expr)                                                        \
    do {                                                                     \
        hipError_t e__ = (expr);                                             \
        if (e__ != hipSuccess) return record_hip_error(e__, #expr);          \
    } while (0)

namespace {

struct DeviceScope { // the handle's device is current inside every call, the caller's afterwards
    int prev = -1;
    explicit DeviceScope(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) (void)hipSetDevice(dev); else prev = -1;
    }
    ~DeviceScope() { if (prev >= 0) (void)hipSetDevice(prev); }
};

cvec_status clear_state(cvec_series_appender *a, hipStream_t st) {
    SER_TRY(hipMemsetAsync(a->fill.p, 0, a->n * 4 + 16, st));
    SER_TRY(hipMemsetAsync(a->len.p, 0, a->n * 8 + 16, st));
    SER_TRY(hipMemsetAsync(a->nulls.p, 0, a->n * 4 + 16, st));
    SER_TRY(hipStreamSynchronize(st));
    a->segs.clear();
    a->tail.clear();
    a->arena.rewind();
    for (auto &u : a->upper) u->rewind();
    a->pinned.rewind();
    a->sealed = 0;
    a->uniform = true;
    a->ufill = 0;
    a->resident_bytes = 0;
    return CVEC_OK;
}

// Encodes `total` values (series v owns [off[v], off[v+1])) and keeps the result at its exact size.
cvec_status encode_segment(cvec_series_appender *a, const void *vals, const uint64_t *off, uint64_t total, Segment *seg,
                           hipStream_t st) {
    const uint64_t n = a->n;
    const size_t cap = cvec_max_encoded_bytes(a->etype, n, total);
    const size_t wsb = encode_workspace_bytes(n, total);
    SER_TRY(a->enc_sec.need((n + 1) * 8));
    SER_TRY(a->enc_res.need(CVEC_RESULT_WORDS * 8));
    SER_TRY(a->enc_ws.need(wsb));
    Segment s;
    SER_TRY(a->arena.take((n + 1) * 8, (void **)&s.off));
    // The encoder writes straight into the arena when a slab has room for the worst case (what the section bytes
    // did not need goes back); otherwise into a scratch buffer, and the arena gets a copy of the exact size.
    uint8_t *direct = (uint8_t *)a->arena.take_if_room(cap + 256);
    uint8_t *dst = direct;
    if (!direct) {
        SER_TRY(a->enc_bytes.need(cap + 256));
        dst = a->enc_bytes.as<uint8_t>();
    }
    SER_TRY(launch_encode(a->etype, vals, off, n, total, dst, cap, s.off, a->enc_sec.as<uint64_t>(), a->enc_res.as<uint64_t>(),
                          a->enc_ws.p, wsb, st));
    uint64_t n_bytes = 0;
    SER_TRY(hipMemcpyAsync(&n_bytes, a->enc_res.p, 8, hipMemcpyDeviceToHost, st));
    SER_TRY(hipStreamSynchronize(st)); // also: the encode buffers are reused by the next call, whatever its stream
    if (n_bytes < n * HEADER_BYTES || n_bytes > cap) return CVEC_ERR_HIP;
    if (direct) {
        a->arena.trim(direct, n_bytes + 16);
        s.bytes = direct;
    } else {
        SER_TRY(a->arena.take(n_bytes + 16, (void **)&s.bytes));
        SER_TRY(hipMemcpyAsync(s.bytes, dst, n_bytes, hipMemcpyDeviceToDevice, st));
        SER_TRY(hipStreamSynchronize(st));
    }
    s.n_bytes = n_bytes;
    *seg = s;
    return CVEC_OK;
}

cvec_status pin_all(cvec_series_appender *a, hipStream_t st);
cvec_status merge_levels(cvec_series_appender *a, hipStream_t st);

// Keeps the counts in a->run as a segment of null sections if any series has one.
// `known` >= 0: the sum of the counts, which the host already knows (uniform calls).
cvec_status null_run_segment(cvec_series_appender *a, std::vector<Segment> *to, hipStream_t st, int64_t known = -1) {
    const uint64_t n = a->n;
    uint64_t total = (uint64_t)known;
    if (known < 0) {
        SER_TRY(launch_exclusive_scan(a->run.as<uint64_t>(), n, a->run_scan.as<uint64_t>(), a->scan_tmp.as<uint64_t>(), st));
        SER_TRY(hipMemcpyAsync(&total, a->run_scan.as<uint64_t>() + n, 8, hipMemcpyDeviceToHost, st));
        SER_TRY(hipStreamSynchronize(st));
    }
    if (total == 0) return CVEC_OK;
    Segment s;
    Arena *from = &a->arena;
    if (to == &a->segs && total > MERGE_MAX_NULL_RUN * n) {
        cvec_status cs = pin_all(a, st);
        if (cs) return cs;
        from = &a->pinned;
        s.pinned = true;
    }
    SER_TRY(from->take(n * 8, (void **)&s.null_run));
    SER_TRY(hipMemcpyAsync(s.null_run, a->run.p, n * 8, hipMemcpyDeviceToDevice, st));
    SER_TRY(hipStreamSynchronize(st));
    s.n_bytes = total;
    to->push_back(s);
    if (to == &a->segs) {
        a->resident_bytes += n * 8;
        return merge_levels(a, st);
    }
    return CVEC_OK;
}

// Uploads the pointer tables of `count` segments starting at `first` of `list` (then of `list2`, if given).
cvec_status upload_tables(cvec_series_appender *a, const Segment *first, size_t count, const std::vector<Segment> *list2,
                          uint32_t *n_seg_out, hipStream_t st) {
    std::vector<const uint64_t *> h_off, h_null;
    std::vector<const uint8_t *> h_bytes;
    for (size_t i = 0; i < count; i++) { h_off.push_back(first[i].off); h_bytes.push_back(first[i].bytes); h_null.push_back(first[i].null_run); }
    if (list2)
        for (const Segment &g : *list2) { h_off.push_back(g.off); h_bytes.push_back(g.bytes); h_null.push_back(g.null_run); }
    const uint32_t n_seg = (uint32_t)h_off.size();
    SER_TRY(a->tab_off.need((size_t)n_seg * 8));
    SER_TRY(a->tab_bytes.need((size_t)n_seg * 8));
    SER_TRY(a->tab_null.need((size_t)n_seg * 8));
    if (n_seg) {
        SER_TRY(hipMemcpyAsync(a->tab_off.p, h_off.data(), (size_t)n_seg * 8, hipMemcpyHostToDevice, st));
        SER_TRY(hipMemcpyAsync(a->tab_bytes.p, h_bytes.data(), (size_t)n_seg * 8, hipMemcpyHostToDevice, st));
        SER_TRY(hipMemcpyAsync(a->tab_null.p, h_null.data(), (size_t)n_seg * 8, hipMemcpyHostToDevice, st));
        SER_TRY(hipStreamSynchronize(st)); // the host vectors go out of scope
    }
    *n_seg_out = n_seg;
    return CVEC_OK;
}

uint64_t segment_resident_bytes(const cvec_series_appender *a, const Segment &g) { return g.null_run ? a->n * 8 : g.n_bytes; }

// Replaces the segments [first, segs.size()) by one segment in `dst` that holds, per series, a header's worth of
// bytes (never read: a segment is laid out like an encoded vector) and their section bytes in order, null runs
// written out as 0x00 bytes.
cvec_status merge_trailing(cvec_series_appender *a, size_t first, Arena &dst, uint32_t level, bool pinned, hipStream_t st) {
    const uint64_t n = a->n;
    const size_t count = a->segs.size() - first;
    if (count == 0) return CVEC_OK;
    uint32_t n_seg = 0;
    cvec_status cs = upload_tables(a, a->segs.data() + first, count, nullptr, &n_seg, st);
    if (cs) return cs;
    Segment m;
    SER_TRY(dst.take((n + 1) * 8, (void **)&m.off));
    // (sizes as for a reader; the element counts that k_ser_sizes also derives are not used here)
    k_ser_sizes<<<lane_blocks(n), SER_THREADS, 0, st>>>(n, 1, 0, a->len.as<uint64_t>(), a->fill.as<uint32_t>(),
                                                        a->tab_off.as<const uint64_t *>(), a->tab_null.as<const uint64_t *>(), n_seg,
                                                        a->out_size.as<uint64_t>(), a->out_nsec.as<uint64_t>());
    SER_TRY(launch_exclusive_scan(a->out_size.as<uint64_t>(), n, m.off, a->scan_tmp.as<uint64_t>(), st));
    uint64_t total = 0;
    SER_TRY(hipMemcpyAsync(&total, m.off + n, 8, hipMemcpyDeviceToHost, st));
    SER_TRY(hipStreamSynchronize(st));
    SER_TRY(dst.take(total + 16, (void **)&m.bytes));
    k_ser_gather<<<wave_blocks(n), SER_THREADS, 0, st>>>(n, a->etype, 1, 0, a->len.as<uint64_t>(), a->fill.as<uint32_t>(),
                                                         a->nulls.as<uint32_t>(), a->tab_off.as<const uint64_t *>(),
                                                         a->tab_bytes.as<const uint8_t *>(), a->tab_null.as<const uint64_t *>(), n_seg,
                                                         m.off, m.bytes);
    SER_TRY(hipGetLastError());
    SER_TRY(hipStreamSynchronize(st)); // the sources may be overwritten once their arena is rewound
    m.n_bytes = total;
    m.level = level;
    m.pinned = pinned;
    for (size_t i = first; i < a->segs.size(); i++) a->resident_bytes -= segment_resident_bytes(a, a->segs[i]);
    a->segs.resize(first);
    a->segs.push_back(m);
    a->resident_bytes += total;
    a->merges++;
    return CVEC_OK;
}

Arena &level_arena(cvec_series_appender *a, uint32_t level) {
    if (level == 0) return a->arena;
    while (a->upper.size() < level) a->upper.emplace_back(new Arena());
    return *a->upper[level - 1];
}

// After a segment was added: while MERGE_FANIN segments of the last one's level end the list, they become one of the
// next level.  The segments of a level lie behind every pinned segment and are all of its arena's contents.
cvec_status merge_levels(cvec_series_appender *a, hipStream_t st) {
    for (;;) {
        if (a->segs.empty() || a->segs.back().pinned) return CVEC_OK;
        const uint32_t level = a->segs.back().level;
        size_t first = a->segs.size();
        while (first > 0 && !a->segs[first - 1].pinned && a->segs[first - 1].level == level) first--;
        if (a->segs.size() - first < MERGE_FANIN) return CVEC_OK;
        cvec_status cs = merge_trailing(a, first, level_arena(a, level + 1), level + 1, false, st);
        if (cs) return cs;
        level_arena(a, level).rewind();
    }
}

// In front of a long run of null sections: everything that is not pinned yet becomes one pinned segment, and the
// arenas of all levels are free again.
cvec_status pin_all(cvec_series_appender *a, hipStream_t st) {
    size_t first = a->segs.size();
    while (first > 0 && !a->segs[first - 1].pinned) first--;
    if (first != a->segs.size()) {
        cvec_status cs = merge_trailing(a, first, a->pinned, 0, true, st);
        if (cs) return cs;
    }
    a->arena.rewind();
    for (auto &u : a->upper) u->rewind();
    return CVEC_OK;
}

template <class U>
cvec_status append_t(cvec_series_appender *a, const void *d_values, const uint64_t *d_elem_off, uint64_t per_vector,
                     int nulls_mode, hipStream_t st) {
    const uint64_t n = a->n;
    uint64_t *cnt = a->cnt.as<uint64_t>(), *off = a->off.as<uint64_t>();
    k_ser_plan<<<lane_blocks(n), SER_THREADS, 0, st>>>(n, d_elem_off, per_vector, nulls_mode, a->fill.as<uint32_t>(),
                                                       a->len.as<uint64_t>(), a->nulls.as<uint32_t>(), cnt, a->run.as<uint64_t>());
    // A scrape gives every series the same number of values: as long as all calls did, the host knows every
    // series' fill and with it what this call completes, and the counts need not be read back.
    if (d_elem_off != nullptr) a->uniform = false;
    uint64_t total = 0;
    int64_t known_run = -1;
    if (a->uniform) {
        const uint64_t tot = (uint64_t)a->ufill + per_vector, sections = tot / FIXED_LEN;
        uint64_t flush = sections * FIXED_LEN;
        if (nulls_mode) flush = (a->ufill != 0 && sections != 0) ? FIXED_LEN : 0; // as k_ser_plan
        total = n * flush;
        known_run = (int64_t)(nulls_mode ? n * (sections - flush / FIXED_LEN) : 0);
        a->ufill = (uint32_t)(tot % FIXED_LEN);
        if (total) SER_TRY(launch_exclusive_scan(cnt, n, off, a->scan_tmp.as<uint64_t>(), st));
    } else {
        SER_TRY(launch_exclusive_scan(cnt, n, off, a->scan_tmp.as<uint64_t>(), st));
        SER_TRY(hipMemcpyAsync(&total, off + n, 8, hipMemcpyDeviceToHost, st));
        SER_TRY(hipStreamSynchronize(st));
    }
    if (total) {
        SER_TRY(a->vals.need(total * sizeof(U)));
        k_ser_flush<U><<<wave_blocks(n), SER_THREADS, 0, st>>>(n, (const U *)d_values, d_elem_off, per_vector,
                                                               a->stage.as<U>(), a->fill.as<uint32_t>(), off, a->vals.as<U>());
    }
    if (d_elem_off == nullptr && per_vector <= 8)
        k_ser_stage_few<U><<<lane_blocks(n), SER_THREADS, 0, st>>>(n, (const U *)d_values, per_vector, a->stage.as<U>(),
                                                                   a->fill.as<uint32_t>());
    else
        k_ser_stage<U><<<wave_blocks(n), SER_THREADS, 0, st>>>(n, (const U *)d_values, d_elem_off, per_vector, a->stage.as<U>(),
                                                               a->fill.as<uint32_t>());
    SER_TRY(hipGetLastError());
    if (total) {
        Segment s;
        cvec_status cs = encode_segment(a, a->vals.p, off, total, &s, st);
        if (cs) return cs;
        a->segs.push_back(s);
        a->resident_bytes += s.n_bytes;
        cs = merge_levels(a, st);
        if (cs) return cs;
    } else {
        SER_TRY(hipStreamSynchronize(st)); // the caller's values may be reused when the call returns
    }
    if (nulls_mode) return null_run_segment(a, &a->segs, st, known_run); // behind the section that the zeros completed
    return CVEC_OK;
}

cvec_status append_any(cvec_series_appender *a, const void *d_values, const uint64_t *d_elem_off, uint64_t per_vector,
                       int nulls_mode, void *stream) {
    if (!a) return CVEC_ERR_BAD_ARGUMENT;
    if (a->sealed) return CVEC_ERR_BAD_ARGUMENT; // between seal and emit the contents are fixed
    if (a->n == 0) return CVEC_OK;
    if (!d_elem_off && per_vector == 0) return CVEC_OK;
    DeviceScope scope(a->device);
    hipStream_t st = (hipStream_t)stream;
    return a->width == 8 ? append_t<uint64_t>(a, d_values, d_elem_off, per_vector, nulls_mode, st)
                         : append_t<uint32_t>(a, d_values, d_elem_off, per_vector, nulls_mode, st);
}

} // namespace

extern "C" {

cvec_status cvec_series_appender_new(int32_t etype, uint64_t n_series, cvec_series_appender **out) {
    if (!out || (etype != CVEC_U32 && etype != CVEC_U64 && etype != CVEC_F32)) return CVEC_ERR_BAD_ARGUMENT;
    *out = nullptr;
    if (n_series > 0xffffffffull) return CVEC_ERR_BAD_ARGUMENT;
    cvec_series_appender *a = new (std::nothrow) cvec_series_appender();
    if (!a) return CVEC_ERR_HIP;
    a->etype = etype;
    a->n = n_series;
    a->width = etype == CVEC_U64 ? 8 : 4;
    hipError_t e = hipGetDevice(&a->device);
    const uint64_t n = n_series;
    if (e == hipSuccess) e = a->stage.need(n * FIXED_LEN * a->width);
    if (e == hipSuccess) e = a->fill.need(n * 4 + 16);
    if (e == hipSuccess) e = a->len.need(n * 8 + 16);
    if (e == hipSuccess) e = a->nulls.need(n * 4 + 16);
    if (e == hipSuccess) e = a->cnt.need((n + 1) * 8);
    if (e == hipSuccess) e = a->off.need((n + 1) * 8);
    if (e == hipSuccess) e = a->scan_tmp.need(scan_scratch_words(n) * 8);
    if (e == hipSuccess) e = a->run.need((n + 1) * 8);
    if (e == hipSuccess) e = a->run_scan.need((n + 1) * 8);
    if (e == hipSuccess) e = a->out_size.need((n + 1) * 8);
    if (e == hipSuccess) e = a->out_off.need((n + 1) * 8);
    if (e == hipSuccess) e = a->out_nsec.need((n + 1) * 8);
    if (e == hipSuccess) e = a->out_sec.need((n + 1) * 8);
    if (e == hipSuccess) e = a->maxlen.need(16);
    if (e != hipSuccess) { delete a; return record_hip_error(e, "cvec_series_appender_new"); }
    cvec_status cs = clear_state(a, nullptr);
    if (cs) { delete a; return cs; }
    *out = a;
    return CVEC_OK;
}

void cvec_series_appender_free(cvec_series_appender *a) {
    if (!a) return;
    DeviceScope scope(a->device);
    delete a;
}

cvec_status cvec_series_appender_reset(cvec_series_appender *a) {
    if (!a) return CVEC_ERR_BAD_ARGUMENT;
    DeviceScope scope(a->device);
    return clear_state(a, nullptr);
}

cvec_status cvec_series_appender_append_dev(cvec_series_appender *a, const void *d_values, const uint64_t *d_elem_off,
                                            uint64_t per_series, void *stream) {
    if (a && !d_values && (d_elem_off || per_series)) return CVEC_ERR_BAD_ARGUMENT;
    return append_any(a, d_values, d_elem_off, per_series, 0, stream);
}

cvec_status cvec_series_appender_append_host(cvec_series_appender *a, const void *h_values, const uint64_t *h_elem_off,
                                             uint64_t per_series, void *stream) {
    if (!a) return CVEC_ERR_BAD_ARGUMENT;
    if (a->sealed) return CVEC_ERR_BAD_ARGUMENT;
    const uint64_t n = a->n;
    if (n == 0) return CVEC_OK;
    uint64_t total = n * per_series;
    if (h_elem_off) {
        for (uint64_t v = 0; v < n; v++)
            if (h_elem_off[v + 1] < h_elem_off[v]) return CVEC_ERR_BAD_ARGUMENT;
        if (h_elem_off[0] != 0) return CVEC_ERR_BAD_ARGUMENT;
        total = h_elem_off[n];
    }
    if (total == 0) return CVEC_OK;
    if (!h_values) return CVEC_ERR_BAD_ARGUMENT;
    DeviceScope scope(a->device);
    hipStream_t st = (hipStream_t)stream;
    SER_TRY(a->up_vals.need(total * a->width));
    SER_TRY(hipMemcpyAsync(a->up_vals.p, h_values, total * a->width, hipMemcpyHostToDevice, st));
    const uint64_t *d_off = nullptr;
    if (h_elem_off) {
        SER_TRY(a->up_off.need((n + 1) * 8));
        SER_TRY(hipMemcpyAsync(a->up_off.p, h_elem_off, (n + 1) * 8, hipMemcpyHostToDevice, st));
        d_off = a->up_off.as<uint64_t>();
    }
    return append_any(a, a->up_vals.p, d_off, per_series, 0, stream); // synchronises on the stream before it returns
}

cvec_status cvec_series_appender_append_nulls_dev(cvec_series_appender *a, const uint64_t *d_elem_off, uint64_t per_series,
                                                  void *stream) {
    return append_any(a, nullptr, d_elem_off, per_series, 1, stream);
}

cvec_status cvec_series_appender_stats(const cvec_series_appender *a, uint64_t *n_segments, uint64_t *encoded_bytes,
                                       uint64_t *state_bytes) {
    if (!a) return CVEC_ERR_BAD_ARGUMENT;
    if (n_segments) *n_segments = a->segs.size();
    if (encoded_bytes) *encoded_bytes = a->resident_bytes;
    if (state_bytes) *state_bytes = a->n * ((uint64_t)FIXED_LEN * a->width + 16) + a->segs.size() * (a->n + 1) * 8; // (a null run's counts are in encoded_bytes)
    // (the slabs behind the segments may hold more than encoded_bytes: they are kept across finish() and reset())
    return CVEC_OK;
}

cvec_status cvec_series_appender_seal(cvec_series_appender *a, int32_t reader, uint64_t total_num_rows, uint64_t *n_bytes,
                                      uint64_t *n_sections, uint64_t detail[2], void *stream) {
    if (!a || a->sealed) return CVEC_ERR_BAD_ARGUMENT;
    if (detail) detail[0] = detail[1] = 0;
    DeviceScope scope(a->device);
    hipStream_t st = (hipStream_t)stream;
    const uint64_t n = a->n;
    // The closing segments belong to a seal that succeeded: every other way out of this call drops them, so that a
    // later seal never gathers segments of an attempt that failed half-way (their arena space comes back with the
    // next finish() or reset()).
    a->tail.clear();
    struct TailGuard {
        cvec_series_appender *a;
        bool keep;
        ~TailGuard() { if (!keep) a->tail.clear(); }
    } tail_guard{a, false};
    if (!reader) {
        // src/vector.rs:344-349: more values than rows, or more rows than the header's u32 can say
        unsigned long long max_len = 0;
        if (n) {
            SER_TRY(hipMemsetAsync(a->maxlen.p, 0, 8, st));
            k_ser_max_len<<<lane_blocks(n), SER_THREADS, 0, st>>>(n, a->len.as<uint64_t>(), a->maxlen.as<unsigned long long>());
            SER_TRY(hipMemcpyAsync(&max_len, a->maxlen.p, 8, hipMemcpyDeviceToHost, st));
            SER_TRY(hipStreamSynchronize(st));
        }
        if (max_len > total_num_rows) {
            if (detail) { detail[0] = total_num_rows; detail[1] = max_len; }
            return CVEC_INVALID_NUM_ROWS;
        }
        if (total_num_rows > 0xffffffffull) {
            if (detail) { detail[0] = total_num_rows; detail[1] = 0xffffffffull; }
            return CVEC_INVALID_NUM_ROWS;
        }
        if (n) {
            uint64_t *cnt = a->cnt.as<uint64_t>(), *off = a->off.as<uint64_t>();
            k_ser_tail_plan<<<lane_blocks(n), SER_THREADS, 0, st>>>(n, total_num_rows, a->len.as<uint64_t>(),
                                                                    a->fill.as<uint32_t>(), cnt, a->run.as<uint64_t>());
            SER_TRY(launch_exclusive_scan(cnt, n, off, a->scan_tmp.as<uint64_t>(), st));
            uint64_t total = 0;
            SER_TRY(hipMemcpyAsync(&total, off + n, 8, hipMemcpyDeviceToHost, st));
            SER_TRY(hipStreamSynchronize(st));
            if (total) {
                SER_TRY(a->vals.need(total * a->width));
                if (a->width == 8)
                    k_ser_flush<uint64_t><<<wave_blocks(n), SER_THREADS, 0, st>>>(n, nullptr, nullptr, 0, a->stage.as<uint64_t>(),
                                                                                  a->fill.as<uint32_t>(), off, a->vals.as<uint64_t>());
                else
                    k_ser_flush<uint32_t><<<wave_blocks(n), SER_THREADS, 0, st>>>(n, nullptr, nullptr, 0, a->stage.as<uint32_t>(),
                                                                                  a->fill.as<uint32_t>(), off, a->vals.as<uint32_t>());
                SER_TRY(hipGetLastError());
                Segment t;
                cvec_status cs = encode_segment(a, a->vals.p, off, total, &t, st);
                if (cs) return cs;
                a->tail.push_back(t);
            }
            cvec_status cs = null_run_segment(a, &a->tail, st);
            if (cs) return cs;
        }
    }
    // pointer tables of the segments, then sizes -> offsets
    uint32_t n_seg = 0;
    {
        cvec_status cs = upload_tables(a, a->segs.data(), a->segs.size(), &a->tail, &n_seg, st);
        if (cs) return cs;
    }
    uint64_t tot[2] = {0, 0};
    if (n) {
        k_ser_sizes<<<lane_blocks(n), SER_THREADS, 0, st>>>(n, reader ? 1 : 0, total_num_rows, a->len.as<uint64_t>(),
                                                            a->fill.as<uint32_t>(), a->tab_off.as<const uint64_t *>(),
                                                            a->tab_null.as<const uint64_t *>(), n_seg,
                                                            a->out_size.as<uint64_t>(), a->out_nsec.as<uint64_t>());
        SER_TRY(launch_exclusive_scan(a->out_size.as<uint64_t>(), n, a->out_off.as<uint64_t>(), a->scan_tmp.as<uint64_t>(), st));
        SER_TRY(launch_exclusive_scan(a->out_nsec.as<uint64_t>(), n, a->out_sec.as<uint64_t>(), a->scan_tmp.as<uint64_t>(), st));
        SER_TRY(hipMemcpyAsync(&tot[0], a->out_off.as<uint64_t>() + n, 8, hipMemcpyDeviceToHost, st));
        SER_TRY(hipMemcpyAsync(&tot[1], a->out_sec.as<uint64_t>() + n, 8, hipMemcpyDeviceToHost, st));
        SER_TRY(hipStreamSynchronize(st));
    } else {
        SER_TRY(hipMemsetAsync(a->out_off.p, 0, 8, st));
        SER_TRY(hipMemsetAsync(a->out_sec.p, 0, 8, st));
        SER_TRY(hipStreamSynchronize(st));
    }
    tail_guard.keep = true;
    a->sealed = reader ? 2 : 1;
    a->sealed_rows = total_num_rows;
    a->sealed_bytes = tot[0];
    a->sealed_sections = tot[1];
    if (n_bytes) *n_bytes = tot[0];
    if (n_sections) *n_sections = tot[1];
    return CVEC_OK;
}

cvec_status cvec_series_appender_emit_dev(cvec_series_appender *a, uint8_t *d_bytes, size_t bytes_cap, uint64_t *d_byte_off,
                                          uint64_t *d_sec_base, void *stream) {
    if (!a || !a->sealed || !d_byte_off || !d_sec_base) return CVEC_ERR_BAD_ARGUMENT;
    if (a->sealed_bytes && !d_bytes) return CVEC_ERR_BAD_ARGUMENT;
    if (bytes_cap < a->sealed_bytes) return CVEC_ERR_CAPACITY;
    DeviceScope scope(a->device);
    hipStream_t st = (hipStream_t)stream;
    const uint64_t n = a->n;
    const int reader = a->sealed == 2;
    const uint32_t n_seg = (uint32_t)(a->segs.size() + a->tail.size());
    SER_TRY(hipMemcpyAsync(d_byte_off, a->out_off.p, (n + 1) * 8, hipMemcpyDeviceToDevice, st));
    SER_TRY(hipMemcpyAsync(d_sec_base, a->out_sec.p, (n + 1) * 8, hipMemcpyDeviceToDevice, st));
    if (n) {
        k_ser_gather<<<wave_blocks(n), SER_THREADS, 0, st>>>(n, a->etype, reader, a->sealed_rows, a->len.as<uint64_t>(),
                                                             a->fill.as<uint32_t>(), a->nulls.as<uint32_t>(),
                                                             a->tab_off.as<const uint64_t *>(), a->tab_bytes.as<const uint8_t *>(),
                                                             a->tab_null.as<const uint64_t *>(), n_seg, a->out_off.as<uint64_t>(), d_bytes);
        SER_TRY(hipGetLastError());
    }
    SER_TRY(hipStreamSynchronize(st));
    if (reader) { // the appender goes on where it was
        a->sealed = 0;
        return CVEC_OK;
    }
    return clear_state(a, st); // src/vector.rs:368: finish() resets the appender
}

cvec_status cvec_series_appender_unseal(cvec_series_appender *a) {
    if (!a) return CVEC_ERR_BAD_ARGUMENT;
    DeviceScope scope(a->device);
    a->tail.clear(); // their arena space comes back with the next finish() or reset()
    a->sealed = 0;
    return CVEC_OK;
}

} // extern "C"
