// cvec_api.hip — the extern "C" entry points declared in include/cvec.h.
#include <stdio.h>
#include <string.h>

#include <stdlib.h>

#include <algorithm>
#include <atomic>
#include <vector>

#include "cvec_device.hpp"

namespace cvec {
// cvec_decode.hip
hipError_t launch_sec_base(const uint8_t *d_bytes, const uint64_t *d_byte_off, uint64_t n, uint64_t *d_sec_base,
                           uint64_t *d_scratch, hipStream_t stream);
uint64_t scan_scratch_words(uint64_t n);
hipError_t launch_read(int mode, int phase, const cvec_dbatch_view &b, const PredArgs &pa, const SinkArgs &sa,
                       uint64_t *desc, uint32_t *sec_res, int32_t *vstatus, uint64_t *d_result, uint64_t *d_per_vector,
                       int32_t *d_status, uint8_t *d_masks, void *d_out, const uint8_t *d_gate, uint64_t *tag,
                       uint32_t *work, hipStream_t stream);
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
                                 const int32_t *vstatus, const uint64_t *tag, uint32_t *valid, uint8_t *parse_failed,
                                 hipStream_t stream);
void set_read_path(int path);
#ifdef CVEC_PROBES
hipError_t debug_sections_times(uint64_t *h_out);
#endif
void set_tail_rule(int rule);
void set_call_options(int path, int tail);
void get_call_options(int *path, int *tail);
int effective_read_path();
int effective_tail_rule();
hipError_t launch_error_detail(int etype, const uint8_t *bytes, const uint64_t *byte_off, uint64_t n, const int32_t *status,
                               const uint32_t *valid, const uint8_t *parse_failed, uint64_t *detail, hipStream_t stream);
hipError_t launch_merge_filter_pass(uint64_t n_vectors, uint64_t n_sections, const uint64_t *tmp_result,
                                    const uint64_t *tmp_per, const uint8_t *tmp_masks, uint64_t *result, uint64_t *per,
                                    uint8_t *masks, hipStream_t stream);
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
// cvec_stream.hip: histograms
size_t hist_workspace_bytes(uint64_t n_hist, uint32_t num_buckets);
hipError_t launch_hist_delta_diff(const uint8_t *d_bytes, const uint64_t *d_byte_off, uint64_t n_bytes,
                                  const uint64_t *d_series_off, uint64_t n_series, uint64_t n_hist, uint32_t num_buckets,
                                  uint8_t *d_out, uint64_t *d_out_off, uint32_t *d_dropped, int32_t *d_status,
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

// Scratch of cvec_filter_count_static_dev: the workspace that the other read entry points take as an
// argument, as a file-scope device array.  Every GPU the library runs on holds its own copy until the
// process ends.  In host code the array's name is only a handle for hipGetSymbolAddress: the address
// that kernels, memsets and copies use is looked up once per device (with that device current) and
// kept in g_static_scratch[device].
static __device__ __align__(256) unsigned char scratch[CVEC_STATIC_SCRATCH_BYTES];
constexpr int MAX_STATIC_DEVICES = 64;
static std::atomic<void *> g_static_scratch[MAX_STATIC_DEVICES];

static cvec_status hip_fail(hipError_t e, const char *what) {
    snprintf(g_hip_err, sizeof(g_hip_err), "%s: %s", what, hipGetErrorString(e));
    return CVEC_ERR_HIP;
}
namespace cvec {
// for the other translation units: keeps the message that cvec_last_hip_error() returns
cvec_status record_hip_error(hipError_t e, const char *what) { return hip_fail(e, what); }
void record_message(const char *msg) { snprintf(g_hip_err, sizeof(g_hip_err), "%s", msg); }
} // namespace cvec
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
    size_t cap = 0;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t n) { // grow-only: a buffer that is large enough is kept
        n = n ? n : 16;
        if (p && cap >= n) return hipSuccess;
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        hipError_t e = hipMalloc(&p, n);
        if (e == hipSuccess) cap = n;
        return e;
    }
    template <class T> T *as() { return (T *)p; }
};

struct PinBuf { // page-locked host memory, grow-only like DevBuf
    void *p = nullptr;
    size_t cap = 0;
    ~PinBuf() { if (p) (void)hipHostFree(p); }
    hipError_t alloc(size_t n) {
        n = n ? n : 16;
        if (p && cap >= n) return hipSuccess;
        if (p) { (void)hipHostFree(p); p = nullptr; cap = 0; }
        hipError_t e = hipHostMalloc(&p, n, hipHostMallocDefault);
        if (e == hipSuccess) cap = n;
        return e;
    }
};
struct StreamBox {
    hipStream_t s = nullptr;
    StreamBox() = default;
    StreamBox(const StreamBox &) = delete;
    StreamBox &operator=(StreamBox &&o) { if (s) (void)hipStreamDestroy(s); s = o.s; o.s = nullptr; return *this; }
    ~StreamBox() { if (s) (void)hipStreamDestroy(s); }
    hipError_t make() { return s ? hipSuccess : hipStreamCreateWithFlags(&s, hipStreamNonBlocking); }
};

constexpr size_t PIPE_SLOTS = 3;
size_t pipe_chunk_bytes() {
    const char *e = getenv("CVEC_PIPE_CHUNK_BYTES");
    const unsigned long long v = e ? strtoull(e, nullptr, 10) : 0;
    return v ? (size_t)v : (size_t)64 << 20;
}

// Chunk c = vectors [first[c], first[c+1]): consecutive vectors whose weights add up to at most
// `target` (a vector heavier than that is a chunk of its own).
template <class Weight> std::vector<uint64_t> plan_chunks(uint64_t n, size_t target, Weight weight) {
    std::vector<uint64_t> first{0};
    uint64_t acc = 0;
    for (uint64_t v = 0; v < n; v++) {
        const uint64_t wv = weight(v);
        if (v > first.back() && acc + wv > target) { first.push_back(v); acc = 0; }
        acc += wv;
    }
    if (n) first.push_back(n);
    return first;
}

cvec_status summary_status(const uint64_t res[CVEC_RESULT_WORDS]) {
    return res[1] ? (cvec_status)res[2] : CVEC_OK;
}

struct EventBox {
    hipEvent_t e = nullptr;
    ~EventBox() { if (e) (void)hipEventDestroy(e); }
    hipError_t make() { return e ? hipSuccess : hipEventCreateWithFlags(&e, hipEventDisableTiming); }
};
struct ReadSlot {
    DevBuf bytes, boff, sbase, ws, res, per, stt, out;
    PinBuf hoff, hres, hper, hstt; // small arrays go through page-locked staging: a copy to pageable memory blocks the host
    EventBox ev_in, ev_run, ev_out; // copy-in done, kernels done, copy-out done
    bool busy = false;
    uint64_t v0 = 0, nv = 0;        // the vectors of the chunk in flight
};
struct EncSlot {
    DevBuf vals, eoff, bytes, boff, sbase, ws, res;
    PinBuf heoff, hres, hboff;
    EventBox ev_in, ev_run, ev_res, ev_out; // copy-in, encode, result block on the host, blob copied out
    size_t chunk = (size_t)-1;             // encoded, blob not yet requested
    size_t leaving = (size_t)-1;           // blob and offsets on their way to the host
};
// The pipeline of the calling thread on the current device (dropped when the device changes): one
// stream per role - copies in, kernels, copies out - because on this platform transfers in the two
// PCIe directions only run at the same time when they come from different streams that each copy
// one way; the slots hand a chunk from stream to stream through events.
template <class S> struct SlotCache {
    int device = -1;
    std::vector<S> slots;
    StreamBox s_in, s_run, s_out;
    S *get() {
        int d = -1;
        if (hipGetDevice(&d) != hipSuccess) return nullptr;
        if (d != device) { slots.clear(); s_in = StreamBox(); s_run = StreamBox(); s_out = StreamBox(); device = d; }
        if (slots.empty()) slots = std::vector<S>(PIPE_SLOTS);
        if (s_in.make() != hipSuccess || s_run.make() != hipSuccess || s_out.make() != hipSuccess) return nullptr;
        return slots.data();
    }
};

// filter + count (decode == false) or decode through the sink chain (decode == true) of a batch in
// host memory, chunk by chunk.  `r` receives the summed result block.
// Ends a pipelined call that failed half way: earlier chunks may still be copying into the caller's
// buffers and into the slots' staging on any of the three streams, and the next call reuses the slots
// without looking at their events.  Waits for all three streams; keeps the first error.
template <class Cache> cvec_status drain_pipeline(Cache *cache, cvec_status st) {
    if (st != CVEC_OK && cache) {
        for (hipStream_t s : {cache->s_in.s, cache->s_run.s, cache->s_out.s})
            if (s) (void)hipStreamSynchronize(s);
    }
    return st;
}

// Test hook (tests/test_gpu_parity.py): the chunk index at which the next pipelined call of this thread
// fails artificially after its copies have been queued, -1 = never.  Set through cvec_debug_fail_chunk.
thread_local int64_t g_fail_chunk = -1;

cvec_status read_pipeline_body(SlotCache<ReadSlot> *cache, bool decode, int32_t etype, const uint8_t *h_bytes,
                               const uint64_t *h_byte_off, uint64_t n, const cvec_pred *pred,
                               const cvec_sink_chain *chain, uint64_t r[CVEC_RESULT_WORDS], uint64_t *h_per_vector,
                               int32_t *h_status, void *h_out, uint64_t out_cap_elems, uint64_t *h_out_off);

cvec_status read_pipeline(bool decode, int32_t etype, const uint8_t *h_bytes, const uint64_t *h_byte_off, uint64_t n,
                          const cvec_pred *pred, const cvec_sink_chain *chain, uint64_t r[CVEC_RESULT_WORDS],
                          uint64_t *h_per_vector, int32_t *h_status, void *h_out, uint64_t out_cap_elems,
                          uint64_t *h_out_off) {
    // The slots (device buffers, pinned offsets, streams) are kept per host thread and device and only
    // grow: allocating them costs as much as moving tens of MiB over PCIe.  They are never freed (a
    // destructor at thread or process exit could run after the HIP runtime has shut down).
    static thread_local SlotCache<ReadSlot> *cache = new SlotCache<ReadSlot>();
    return drain_pipeline(cache, read_pipeline_body(cache, decode, etype, h_bytes, h_byte_off, n, pred, chain, r,
                                                    h_per_vector, h_status, h_out, out_cap_elems, h_out_off));
}

cvec_status read_pipeline_body(SlotCache<ReadSlot> *cache, bool decode, int32_t etype, const uint8_t *h_bytes,
                               const uint64_t *h_byte_off, uint64_t n, const cvec_pred *pred,
                               const cvec_sink_chain *chain, uint64_t r[CVEC_RESULT_WORDS], uint64_t *h_per_vector,
                               int32_t *h_status, void *h_out, uint64_t out_cap_elems, uint64_t *h_out_off) {
    if (!etype_ok(etype) || !h_byte_off || (n && !h_bytes)) return CVEC_ERR_BAD_ARGUMENT;
    for (uint64_t v = 0; v < n; v++)
        if (h_byte_off[v + 1] < h_byte_off[v]) return CVEC_ERR_BAD_ARGUMENT;
    const int w = elem_width(etype);
    // sections per vector from the headers (u32 num_elements at +8, reference src/vector.rs:165)
    std::vector<uint64_t> sec_base(n + 1, 0);
    for (uint64_t v = 0; v < n; v++) {
        uint64_t c = 0;
        if (h_byte_off[v + 1] - h_byte_off[v] >= (uint64_t)HEADER_BYTES) {
            uint32_t ne;
            memcpy(&ne, h_bytes + h_byte_off[v] + 8, 4);
            c = header_slots(ne, h_byte_off[v + 1] - h_byte_off[v]);
        }
        sec_base[v + 1] = sec_base[v] + c;
    }
    if (decode) {
        if (h_out_off)
            for (uint64_t v = 0; v <= n; v++) h_out_off[v] = sec_base[v] * FIXED_LEN;
        const uint64_t n_out = sec_base[n] * FIXED_LEN;
        if (n_out > out_cap_elems || (n_out && !h_out)) return CVEC_ERR_CAPACITY;
    }
    const std::vector<uint64_t> first = plan_chunks(n, pipe_chunk_bytes(), [&](uint64_t v) {
        return h_byte_off[v + 1] - h_byte_off[v] + 16 + (decode ? (sec_base[v + 1] - sec_base[v]) * FIXED_LEN * w : 0);
    });
    const size_t nchunks = first.size() - 1;
    uint64_t max_vec = 0, max_bytes = 0, max_sec = 0;
    for (size_t c = 0; c < nchunks; c++) {
        max_vec = std::max(max_vec, first[c + 1] - first[c]);
        max_bytes = std::max(max_bytes, h_byte_off[first[c + 1]] - h_byte_off[first[c]]);
        max_sec = std::max(max_sec, sec_base[first[c + 1]] - sec_base[first[c]]);
    }
    const size_t nslots = std::min<size_t>(PIPE_SLOTS, std::max<size_t>(nchunks, 1));
    ReadSlot *slots = cache->get();
    if (!slots) return hip_fail(hipGetLastError(), "pipeline streams");
    hipStream_t s_in = cache->s_in.s, s_run = cache->s_run.s, s_out = cache->s_out.s;
    const size_t wsb = cvec_workspace_bytes(max_vec, max_sec);
    for (size_t k = 0; k < nslots; k++) {
        ReadSlot &sl = slots[k];
        sl.busy = false;
        HIP_TRY(sl.bytes.alloc(align_up(max_bytes + 16, 256)));
        HIP_TRY(sl.boff.alloc((max_vec + 1) * 8));
        HIP_TRY(sl.sbase.alloc((max_vec + 1) * 8));
        HIP_TRY(sl.ws.alloc(wsb));
        HIP_TRY(sl.res.alloc(CVEC_RESULT_WORDS * 8));
        if (h_per_vector) { HIP_TRY(sl.per.alloc(max_vec * 8)); HIP_TRY(sl.hper.alloc(max_vec * 8)); }
        if (h_status) { HIP_TRY(sl.stt.alloc(max_vec * 4)); HIP_TRY(sl.hstt.alloc(max_vec * 4)); }
        if (decode) HIP_TRY(sl.out.alloc(max_sec * FIXED_LEN * w));
        HIP_TRY(sl.hoff.alloc(2 * (max_vec + 1) * 8));
        HIP_TRY(sl.hres.alloc(CVEC_RESULT_WORDS * 8));
        HIP_TRY(sl.ev_in.make());
        HIP_TRY(sl.ev_run.make());
        HIP_TRY(sl.ev_out.make());
    }
    auto collect = [&](ReadSlot &sl) -> cvec_status { // waits for the slot's chunk and folds its result block in
        if (!sl.busy) return CVEC_OK;
        HIP_TRY(hipEventSynchronize(sl.ev_out.e));
        sl.busy = false;
        if (h_per_vector && sl.nv) memcpy(h_per_vector + sl.v0, sl.hper.p, sl.nv * 8);
        if (h_status && sl.nv) memcpy(h_status + sl.v0, sl.hstt.p, sl.nv * 4);
        const uint64_t *q = (const uint64_t *)sl.hres.p;
        r[0] += q[0];
        r[1] += q[1];
        r[2] = std::max(r[2], q[2]);
        r[3] += q[3];
        return CVEC_OK;
    };
    for (size_t c = 0; c < nchunks; c++) {
        ReadSlot &sl = slots[c % nslots];
        cvec_status st = collect(sl);
        if (st) return st;
        const uint64_t v0 = first[c], nv = first[c + 1] - v0;
        const uint64_t b0 = h_byte_off[v0], nbytes = h_byte_off[first[c + 1]] - b0;
        const uint64_t s0 = sec_base[v0], nsec = sec_base[first[c + 1]] - s0;
        uint64_t *bo = (uint64_t *)sl.hoff.p, *sb = bo + (nv + 1);
        for (uint64_t i = 0; i <= nv; i++) { bo[i] = h_byte_off[v0 + i] - b0; sb[i] = sec_base[v0 + i] - s0; }
        if (nbytes) HIP_TRY(hipMemcpyAsync(sl.bytes.p, h_bytes + b0, nbytes, hipMemcpyHostToDevice, s_in));
        HIP_TRY(hipMemcpyAsync(sl.boff.p, bo, (nv + 1) * 8, hipMemcpyHostToDevice, s_in));
        HIP_TRY(hipMemcpyAsync(sl.sbase.p, sb, (nv + 1) * 8, hipMemcpyHostToDevice, s_in));
        HIP_TRY(hipEventRecord(sl.ev_in.e, s_in));
        HIP_TRY(hipStreamWaitEvent(s_run, sl.ev_in.e, 0));
        cvec_dbatch_view view;
        view.etype = etype;
        view._pad = 0;
        view.n_vectors = nv;
        view.n_sections = nsec;
        view.n_bytes = nbytes;
        view.d_bytes = sl.bytes.as<uint8_t>();
        view.d_byte_off = sl.boff.as<uint64_t>();
        view.d_sec_base = sl.sbase.as<uint64_t>();
        if (decode) {
            if (nsec) HIP_TRY(hipMemsetAsync(sl.out.p, 0, nsec * FIXED_LEN * w, s_run));
            st = cvec_decode_dev(&view, chain, sl.out.p, sl.res.as<uint64_t>(), h_status ? sl.stt.as<int32_t>() : nullptr,
                                 sl.ws.p, wsb, s_run);
        } else {
            st = cvec_filter_count_dev(&view, pred, sl.res.as<uint64_t>(), h_per_vector ? sl.per.as<uint64_t>() : nullptr,
                                       h_status ? sl.stt.as<int32_t>() : nullptr, nullptr, sl.ws.p, wsb, s_run);
        }
        if (st) return st;
        if (g_fail_chunk == (int64_t)c) { g_fail_chunk = -1; return CVEC_ERR_HIP; } // test hook: fail with work in flight
        HIP_TRY(hipEventRecord(sl.ev_run.e, s_run));
        HIP_TRY(hipStreamWaitEvent(s_out, sl.ev_run.e, 0));
        HIP_TRY(hipMemcpyAsync(sl.hres.p, sl.res.p, CVEC_RESULT_WORDS * 8, hipMemcpyDeviceToHost, s_out));
        if (h_per_vector && nv) HIP_TRY(hipMemcpyAsync(sl.hper.p, sl.per.p, nv * 8, hipMemcpyDeviceToHost, s_out));
        if (h_status && nv) HIP_TRY(hipMemcpyAsync(sl.hstt.p, sl.stt.p, nv * 4, hipMemcpyDeviceToHost, s_out));
        sl.v0 = v0;
        sl.nv = nv;
        if (decode && nsec)
            HIP_TRY(hipMemcpyAsync((uint8_t *)h_out + s0 * FIXED_LEN * w, sl.out.p, nsec * FIXED_LEN * w,
                                   hipMemcpyDeviceToHost, s_out));
        HIP_TRY(hipEventRecord(sl.ev_out.e, s_out));
        sl.busy = true;
    }
    for (size_t k = 0; k < nslots; k++) {
        cvec_status st = collect(slots[k]);
        if (st) return st;
    }
    return CVEC_OK;
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
    case CVEC_ERR_CAPACITY: return "OutputTooSmall";
    default: return "Unknown";
    }
}

// The current device's copy of `scratch`.  `resolve`: look the address up if this device has none yet (cvec_init);
// the entry point that must not allocate only reads what cvec_init stored, because the first symbol lookup on a
// device may load the code object there, which allocates and is not allowed under stream capture.
static cvec_status static_scratch(void **out, bool resolve) {
    int d = -1;
    HIP_TRY(hipGetDevice(&d));
    if (d < 0 || d >= MAX_STATIC_DEVICES) return CVEC_ERR_BAD_ARGUMENT;
    void *p = g_static_scratch[d].load(std::memory_order_acquire);
    if (!p && !resolve) {
        snprintf(g_hip_err, sizeof(g_hip_err), "cvec_filter_count_static_dev: call cvec_init(%d) first (it resolves the "
                                               "static scratch of the device; this entry point itself allocates and loads nothing)", d);
        return CVEC_ERR_BAD_ARGUMENT;
    }
    if (!p) {
        HIP_TRY(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        if (!p) return hip_fail(hipErrorInvalidSymbol, "hipGetSymbolAddress(scratch)");
        g_static_scratch[d].store(p, std::memory_order_release);
    }
    *out = p;
    return CVEC_OK;
}

cvec_status cvec_init(int device) {
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count == 0) return CVEC_ERR_NO_DEVICE;
    if (device < 0 || device >= count) return CVEC_ERR_BAD_ARGUMENT;
    HIP_TRY(hipSetDevice(device));
    (void)hipFree(nullptr);
    void *p = nullptr;
    return static_scratch(&p, true); // this device's address of the static scratch: the entry point that uses it only reads it
}

cvec_status cvec_set_read_path(int32_t path) {
    if (path != PATH_AUTO && path != PATH_SLOTS && path != PATH_VSTREAM) return CVEC_ERR_BAD_ARGUMENT;
    set_read_path(path);
    return CVEC_OK;
}

#ifdef CVEC_PROBES
// probe build only: 8192 x 2 wall-clock stamps (start, end) of the last k_sections launch, per wave
cvec_status cvec_debug_sections_times(uint64_t *h_out) {
    HIP_TRY(cvec::debug_sections_times(h_out));
    return CVEC_OK;
}
#endif

cvec_status cvec_set_tail_rule(int32_t rule) {
    if (rule != CVEC_TAIL_FORMAT && rule != CVEC_TAIL_REFERENCE) return CVEC_ERR_BAD_ARGUMENT;
    set_tail_rule(rule);
    return CVEC_OK;
}

// The options of one call: valid values or CVEC_OPT_DEFAULT.
static cvec_status check_options(const cvec_read_options *o) {
    if (!o) return CVEC_OK;
    if (o->read_path != CVEC_OPT_DEFAULT && o->read_path != PATH_AUTO && o->read_path != PATH_SLOTS && o->read_path != PATH_VSTREAM)
        return CVEC_ERR_BAD_ARGUMENT;
    if (o->tail_rule != CVEC_OPT_DEFAULT && o->tail_rule != CVEC_TAIL_FORMAT && o->tail_rule != CVEC_TAIL_REFERENCE)
        return CVEC_ERR_BAD_ARGUMENT;
    return CVEC_OK;
}

namespace {
// Puts a call's options in force on this thread until the call returns (and restores what an enclosing call had set).
struct CallOptions {
    int old_path, old_tail;
    explicit CallOptions(const cvec_read_options *o) {
        get_call_options(&old_path, &old_tail);
        if (o) set_call_options(o->read_path >= 0 ? o->read_path : old_path, o->tail_rule >= 0 ? o->tail_rule : old_tail);
    }
    ~CallOptions() { set_call_options(old_path, old_tail); }
};
} // namespace

cvec_status cvec_read_options_effective(const cvec_read_options *opts, int32_t *read_path, int32_t *tail_rule) {
    cvec_status st = check_options(opts);
    if (st) return st;
    CallOptions in_force(opts);
    if (read_path) *read_path = effective_read_path();
    if (tail_rule) *tail_rule = effective_tail_rule();
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
//   desc[n_sections] u64 | sec_res[n_sections] u32 | vstatus[n_vectors] i32 | scan scratch | range counters
struct ReadWs {
    uint64_t *desc;
    uint32_t *sec_res;
    int32_t *vstatus;
    uint64_t *scan; // 256 bytes or more of scratch: [0, 4) a spare result block, [8] and [9] the two tag words
    uint32_t *work; // range counters of the section kernel: WORK_COUNTERS lines of 128 bytes, cleared by every pass
    uint64_t *tag;  // [0] whose section index the workspace holds (written by the index pass, cleared by a
                    // vector-streaming pass), [1] VS_MAGIC when the per-vector array holds valid-section counts
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
    w.tag = w.scan + 8;
    w.work = (uint32_t *)(b + off); off += (size_t)(WORK_COUNTERS + 1) * WORK_STRIDE * 4;
    off += align_up(n_sections + 1, 256); // behind the counters: one byte per section, its distance to the vector's end
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
    uint64_t *scan_tmp = nullptr;
    HIP_TRY(hipMalloc((void **)&scan_tmp, scan_scratch_words(n_vectors) * 8));
    hipError_t e = launch_sec_base(d_bytes, d_byte_off, n_vectors, d_sec_base, scan_tmp, s);
    uint64_t total = 0;
    if (e == hipSuccess) e = hipMemcpyAsync(&total, d_sec_base + n_vectors, 8, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(scan_tmp);
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
    return cvec_filter_count_gated_dev(batch, pred, nullptr, d_result, d_per_vector, d_status, d_masks, d_workspace,
                                       workspace_bytes, stream);
}

static cvec_status filter_count_phase(int phase, const cvec_dbatch_view *batch, const cvec_pred *pred,
                                      const uint8_t *d_gate, uint64_t *d_result, uint64_t *d_per_vector,
                                      int32_t *d_status, uint8_t *d_masks, void *d_workspace, size_t workspace_bytes,
                                      void *stream);

cvec_status cvec_filter_count_gated_dev(const cvec_dbatch_view *batch, const cvec_pred *pred, const uint8_t *d_gate,
                                        uint64_t *d_result, uint64_t *d_per_vector, int32_t *d_status,
                                        uint8_t *d_masks, void *d_workspace, size_t workspace_bytes, void *stream) {
    return filter_count_phase(READ_ALL, batch, pred, d_gate, d_result, d_per_vector, d_status, d_masks, d_workspace,
                              workspace_bytes, stream);
}

cvec_status cvec_index_dev(const cvec_dbatch_view *batch, void *d_workspace, size_t workspace_bytes, void *stream) {
    cvec_status st = check_view(batch);
    if (st) return st;
    if (!d_workspace) return CVEC_ERR_BAD_ARGUMENT;
    ReadWs w = carve_read_ws(d_workspace, batch->n_vectors, batch->n_sections);
    if (w.total > workspace_bytes) return CVEC_ERR_WORKSPACE;
    PredArgs pa;
    memset(&pa, 0, sizeof(pa));
    SinkArgs sa;
    memset(&sa, 0, sizeof(sa));
    // the index pass clears a result block as its side job; here that block is scratch of the workspace
    static_assert(CVEC_RESULT_WORDS * 8 <= 256, "the scan scratch is at least 256 bytes");
    hipError_t e = launch_read(0, READ_INDEX_ONLY, *batch, pa, sa, w.desc, w.sec_res, w.vstatus, w.scan, nullptr,
                               nullptr, nullptr, nullptr, nullptr, w.tag, w.work, (hipStream_t)stream);
    if (e != hipSuccess) return hip_fail(e, "cvec_index_dev");
    return CVEC_OK;
}

cvec_status cvec_filter_count_indexed_dev(const cvec_dbatch_view *batch, const cvec_pred *pred, const uint8_t *d_gate,
                                          uint64_t *d_result, uint64_t *d_per_vector, int32_t *d_status,
                                          uint8_t *d_masks, void *d_workspace, size_t workspace_bytes, void *stream) {
    return filter_count_phase(READ_INDEXED, batch, pred, d_gate, d_result, d_per_vector, d_status, d_masks,
                              d_workspace, workspace_bytes, stream);
}

cvec_status cvec_filter_count_static_dev(const cvec_dbatch_view *batch, const cvec_pred *pred, uint64_t *d_result,
                                         uint64_t *d_per_vector, int32_t *d_status, uint8_t *d_masks, void *stream) {
    if (!batch) return CVEC_ERR_BAD_ARGUMENT;
    const size_t need = cvec_workspace_bytes(batch->n_vectors, batch->n_sections);
    if (need > (size_t)CVEC_STATIC_SCRATCH_BYTES) {
        snprintf(g_hip_err, sizeof(g_hip_err),
                 "cvec_filter_count_static_dev: a batch of %llu vectors / %llu sections needs %zu bytes of scratch, the "
                 "static scratch holds %d (use cvec_filter_count_dev with a workspace)",
                 (unsigned long long)batch->n_vectors, (unsigned long long)batch->n_sections, need,
                 (int)CVEC_STATIC_SCRATCH_BYTES);
        return CVEC_ERR_WORKSPACE;
    }
    void *ws = nullptr;
    cvec_status st = static_scratch(&ws, false);
    if (st) return st;
    // Every slot of the scratch that the kernels read is written earlier in the same call (descriptors,
    // per-section results, per-vector header statuses, the index tag), so nothing has to be cleared.
    return filter_count_phase(READ_ALL, batch, pred, nullptr, d_result, d_per_vector, d_status, d_masks, ws,
                              CVEC_STATIC_SCRATCH_BYTES, stream);
}

static cvec_status filter_count_phase(int phase, const cvec_dbatch_view *batch, const cvec_pred *pred,
                                      const uint8_t *d_gate, uint64_t *d_result, uint64_t *d_per_vector,
                                      int32_t *d_status, uint8_t *d_masks, void *d_workspace, size_t workspace_bytes,
                                      void *stream) {
    cvec_status st = check_view(batch);
    if (st) return st;
    if (!pred || !d_result || !d_workspace) return CVEC_ERR_BAD_ARGUMENT;
    if ((((uintptr_t)d_gate) & 15) != 0 || (((uintptr_t)d_masks) & 7) != 0) return CVEC_ERR_BAD_ARGUMENT;
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
    hipError_t e = launch_read(0, phase, *batch, pa, sa, w.desc, w.sec_res, w.vstatus, d_result, d_per_vector, d_status,
                               d_masks, nullptr, d_gate, w.tag, w.work, (hipStream_t)stream);
    if (e != hipSuccess) return hip_fail(e, "cvec_filter_count_dev");
    return CVEC_OK;
}

// Workspace of the large-set filter: the read workspace, then a result block, per-vector counts and masks for the
// passes after the first.
struct SetWs {
    size_t read_bytes, result_off, per_off, masks_off, total;
};
static SetWs carve_set_ws(uint64_t n_vectors, uint64_t n_sections) {
    SetWs w;
    w.read_bytes = align_up(cvec_workspace_bytes(n_vectors, n_sections), 256);
    w.result_off = w.read_bytes;
    w.per_off = w.result_off + 256;
    w.masks_off = w.per_off + align_up((n_vectors + 1) * 8, 256);
    w.total = w.masks_off + align_up((n_sections + 1) * 32, 256);
    return w;
}

size_t cvec_filter_set_workspace_bytes(uint64_t n_vectors, uint64_t n_sections) {
    return carve_set_ws(n_vectors, n_sections).total;
}

cvec_status cvec_filter_count_set_dev(const cvec_dbatch_view *batch, const uint64_t *values, uint32_t n_values,
                                      uint64_t *d_result, uint64_t *d_per_vector, int32_t *d_status, uint8_t *d_masks,
                                      void *d_workspace, size_t workspace_bytes, void *stream) {
    cvec_status st = check_view(batch);
    if (st) return st;
    if (!values || n_values == 0 || !d_result || !d_workspace) return CVEC_ERR_BAD_ARGUMENT;
    const SetWs w = carve_set_ws(batch->n_vectors, batch->n_sections);
    if (w.total > workspace_bytes) return CVEC_ERR_WORKSPACE;
    // Distinct members only (a value listed twice, or +0.0 beside -0.0, must not count its elements twice once the
    // set is split into passes); a NaN equals nothing (IEEE ==) and is dropped unless it is all there is.
    std::vector<uint64_t> set;
    set.reserve(n_values);
    const bool f32 = batch->etype == CVEC_F32;
    auto key = [&](uint64_t v) -> uint64_t {
        if (batch->etype != CVEC_U64) v &= 0xffffffffull;
        if (f32 && (v & 0x7fffffffull) == 0) v = 0;
        return v;
    };
    for (uint32_t i = 0; i < n_values; i++) {
        const uint64_t k = key(values[i]);
        if (f32 && (k & 0x7fffffffull) > 0x7f800000ull) continue;
        bool seen = false;
        for (uint64_t s : set) seen |= key(s) == k;
        if (!seen) set.push_back(batch->etype != CVEC_U64 ? (values[i] & 0xffffffffull) : values[i]);
    }
    if (set.empty()) set.push_back(values[0] & 0xffffffffull); // only NaNs: one pass that matches nothing
    uint8_t *base = (uint8_t *)d_workspace;
    uint64_t *tmp_result = (uint64_t *)(base + w.result_off);
    uint64_t *tmp_per = (uint64_t *)(base + w.per_off);
    uint8_t *tmp_masks = base + w.masks_off;
    for (size_t at = 0, pass = 0; at < set.size(); at += CVEC_MAX_PRED_VALUES, pass++) {
        cvec_pred p;
        memset(&p, 0, sizeof(p));
        p.kind = CVEC_PRED_ONEOF;
        p.count = (uint32_t)(set.size() - at < CVEC_MAX_PRED_VALUES ? set.size() - at : CVEC_MAX_PRED_VALUES);
        for (uint32_t i = 0; i < p.count; i++) p.values[i] = set[at + i];
        if (pass == 0) {
            st = filter_count_phase(READ_ALL, batch, &p, nullptr, d_result, d_per_vector, d_status, d_masks, d_workspace,
                                    w.read_bytes, stream);
            if (st) return st;
            continue;
        }
        st = filter_count_phase(READ_ALL, batch, &p, nullptr, tmp_result, d_per_vector ? tmp_per : nullptr, nullptr,
                                d_masks ? tmp_masks : nullptr, d_workspace, w.read_bytes, stream);
        if (st) return st;
        hipError_t e = launch_merge_filter_pass(batch->n_vectors, batch->n_sections, tmp_result, tmp_per, tmp_masks, d_result,
                                                d_per_vector, d_masks, (hipStream_t)stream);
        if (e != hipSuccess) return hip_fail(e, "cvec_filter_count_set_dev");
    }
    return CVEC_OK;
}

cvec_status cvec_filter_count_opt_dev(const cvec_dbatch_view *batch, const cvec_pred *pred, const cvec_read_options *opts,
                                      uint64_t *d_result, uint64_t *d_per_vector, int32_t *d_status, uint8_t *d_masks,
                                      void *d_workspace, size_t workspace_bytes, void *stream) {
    cvec_status st = check_options(opts);
    if (st) return st;
    CallOptions in_force(opts);
    return filter_count_phase(opts && opts->indexed ? READ_INDEXED : READ_ALL, batch, pred, opts ? opts->d_gate : nullptr,
                              d_result, d_per_vector, d_status, d_masks, d_workspace, workspace_bytes, stream);
}

static cvec_status decode_phase(int phase, const cvec_dbatch_view *batch, const cvec_sink_chain *chain, void *d_out,
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
    hipError_t e = launch_read(1, phase, *batch, pa, sa, w.desc, w.sec_res, w.vstatus, d_result, nullptr, d_status,
                               nullptr, d_out, nullptr, w.tag, w.work, (hipStream_t)stream);
    if (e != hipSuccess) return hip_fail(e, "cvec_decode_dev");
    return CVEC_OK;
}

cvec_status cvec_decode_dev(const cvec_dbatch_view *batch, const cvec_sink_chain *chain, void *d_out,
                            uint64_t *d_result, int32_t *d_status, void *d_workspace, size_t workspace_bytes,
                            void *stream) {
    return decode_phase(READ_ALL, batch, chain, d_out, d_result, d_status, d_workspace, workspace_bytes, stream);
}

cvec_status cvec_decode_indexed_dev(const cvec_dbatch_view *batch, const cvec_sink_chain *chain, void *d_out,
                                    uint64_t *d_result, int32_t *d_status, void *d_workspace,
                                    size_t workspace_bytes, void *stream) {
    return decode_phase(READ_INDEXED, batch, chain, d_out, d_result, d_status, d_workspace, workspace_bytes, stream);
}

cvec_status cvec_decode_opt_dev(const cvec_dbatch_view *batch, const cvec_sink_chain *chain, const cvec_read_options *opts,
                                void *d_out, uint64_t *d_result, int32_t *d_status, void *d_workspace,
                                size_t workspace_bytes, void *stream) {
    cvec_status st = check_options(opts);
    if (st) return st;
    if (opts && opts->d_gate) return CVEC_ERR_BAD_ARGUMENT; // a gate belongs to a filter pass
    CallOptions in_force(opts);
    return decode_phase(opts && opts->indexed ? READ_INDEXED : READ_ALL, batch, chain, d_out, d_result, d_status,
                        d_workspace, workspace_bytes, stream);
}

static_assert(CVEC_MAX_COLUMNS == CVEC_MAX_PRED_VALUES, "MaskColumns is sized by CVEC_MAX_PRED_VALUES");

cvec_status cvec_valid_sections_dev(const cvec_dbatch_view *batch, const void *d_workspace, size_t workspace_bytes,
                                    uint32_t *d_valid, uint8_t *d_parse_failed, void *stream) {
    cvec_status st = check_view(batch);
    if (st) return st;
    if (!d_workspace || (batch->n_vectors && !d_valid)) return CVEC_ERR_BAD_ARGUMENT;
    ReadWs w = carve_read_ws((void *)d_workspace, batch->n_vectors, batch->n_sections);
    if (w.total > workspace_bytes) return CVEC_ERR_WORKSPACE;
    hipError_t e = launch_valid_sections(batch->d_sec_base, batch->n_vectors, w.desc, w.sec_res, w.vstatus, w.tag, d_valid,
                                         d_parse_failed, (hipStream_t)stream);
    if (e != hipSuccess) return hip_fail(e, "cvec_valid_sections_dev");
    return CVEC_OK;
}

// multi workspace: popc[n] u16 | ended[n] u8
cvec_status cvec_error_detail_dev(const cvec_dbatch_view *batch, const int32_t *d_status, const uint32_t *d_valid,
                                  const uint8_t *d_parse_failed, uint64_t *d_detail, void *stream) {
    cvec_status st = check_view(batch);
    if (st) return st;
    if (batch->n_vectors && (!d_status || !d_valid || !d_parse_failed || !d_detail)) return CVEC_ERR_BAD_ARGUMENT;
    hipError_t e = launch_error_detail(batch->etype, batch->d_bytes, batch->d_byte_off, batch->n_vectors, d_status, d_valid,
                                       d_parse_failed, d_detail, (hipStream_t)stream);
    if (e != hipSuccess) return hip_fail(e, "cvec_error_detail_dev");
    return CVEC_OK;
}

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

static inline bool stream_kind_ok(int32_t k) {
    return k == CVEC_STREAM_U64_DELTA || k == CVEC_STREAM_F64_XOR || k == CVEC_STREAM_U64_PLAIN;
}

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
    if (bytes_cap < cvec_stream_max_packed_bytes(kind, n_streams, n_values_total)) return CVEC_ERR_CAPACITY;
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

size_t cvec_hist_max_packed_bytes(uint64_t n_hist, uint32_t num_buckets) {
    return (size_t)n_hist * (((size_t)num_buckets + 7) / 8 + 1) * 66 + 16;
}

size_t cvec_hist_workspace_bytes(uint64_t n_hist, uint32_t num_buckets) { return hist_workspace_bytes(n_hist, num_buckets); }

cvec_status cvec_hist_delta_diff_dev(const uint8_t *d_bytes, const uint64_t *d_byte_off, uint64_t n_bytes,
                                     const uint64_t *d_series_off, uint64_t n_series, uint64_t n_hist,
                                     uint32_t num_buckets, uint8_t *d_out, size_t out_cap, uint64_t *d_out_off,
                                     uint32_t *d_dropped, int32_t *d_status, uint64_t *d_result, void *d_workspace,
                                     size_t workspace_bytes, void *stream) {
    if (!d_byte_off || !d_series_off || !d_out || !d_out_off || !d_dropped || !d_status || !d_result || !d_workspace ||
        (n_bytes && !d_bytes))
        return CVEC_ERR_BAD_ARGUMENT;
    if (num_buckets > 0xffff) return CVEC_ERR_BAD_ARGUMENT; // BinHistogramHeader.num_buckets is a u16 (src/histogram.rs:21)
    if ((((uintptr_t)d_bytes) & 15) != 0 || (((uintptr_t)d_out) & 15) != 0) return CVEC_ERR_BAD_ARGUMENT;
    if (out_cap < cvec_hist_max_packed_bytes(n_hist, num_buckets)) return CVEC_ERR_CAPACITY;
    if (workspace_bytes < hist_workspace_bytes(n_hist, num_buckets)) return CVEC_ERR_WORKSPACE;
    hipError_t e = launch_hist_delta_diff(d_bytes, d_byte_off, n_bytes, d_series_off, n_series, n_hist, num_buckets, d_out,
                                          d_out_off, d_dropped, d_status, d_result, d_workspace, (hipStream_t)stream);
    if (e != hipSuccess) return hip_fail(e, "cvec_hist_delta_diff_dev");
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
    if (bytes_cap < cvec_max_encoded_bytes(etype, n_vectors, n_elements_total)) return CVEC_ERR_CAPACITY;
    if (workspace_bytes < encode_workspace_bytes(n_vectors, n_elements_total)) return CVEC_ERR_WORKSPACE;
    hipError_t e = launch_encode(etype, d_values, d_elem_off, n_vectors, n_elements_total, d_bytes, bytes_cap,
                                 d_byte_off, d_sec_base, d_result, d_workspace, workspace_bytes, (hipStream_t)stream);
    if (e != hipSuccess) return hip_fail(e, "cvec_encode_dev");
    return CVEC_OK;
}

// ---- host-buffer entry points --------------------------------------------------------------------
// The batch is cut into chunks of whole vectors (about 64 MiB each; CVEC_PIPE_CHUNK_BYTES overrides,
// read per call) that rotate through PIPE_SLOTS slots of device buffers and pass from a copy-in
// stream to a kernel stream to a copy-out stream: the copy-in of one chunk, the kernels of another
// and the copy-out of a third overlap, and the two PCIe directions run at the same time.  Copies from or to pageable host memory are staged by the
// runtime and overlap little; page-locked caller buffers (hipHostMalloc / hipHostRegister) get the
// full effect.  A batch of one chunk behaves like a plain upload - run - download.

cvec_status cvec_filter_count_batch(int32_t etype, const uint8_t *h_bytes, const uint64_t *h_byte_off,
                                    uint64_t n_vectors, const cvec_pred *pred, uint64_t *h_total,
                                    uint64_t *h_per_vector, int32_t *h_status) {
    if (!h_total || !pred) return CVEC_ERR_BAD_ARGUMENT;
    uint64_t r[CVEC_RESULT_WORDS] = {0, 0, 0, 0};
    cvec_status st = read_pipeline(false, etype, h_bytes, h_byte_off, n_vectors, pred, nullptr, r, h_per_vector, h_status,
                                   nullptr, 0, nullptr);
    if (st) return st;
    *h_total = r[0];
    return summary_status(r);
}

cvec_status cvec_decode_batch(int32_t etype, const uint8_t *h_bytes, const uint64_t *h_byte_off, uint64_t n_vectors,
                              const cvec_sink_chain *chain, void *h_out, uint64_t out_cap_elems, uint64_t *h_out_off,
                              int32_t *h_status) {
    uint64_t r[CVEC_RESULT_WORDS] = {0, 0, 0, 0};
    cvec_status st = read_pipeline(true, etype, h_bytes, h_byte_off, n_vectors, nullptr, chain, r, nullptr, h_status, h_out,
                                   out_cap_elems, h_out_off);
    if (st) return st;
    return summary_status(r);
}

// The host-buffer calls with per-call options: the pipeline's kernels are launched by the calling thread, so the
// options in force on it reach every chunk.  No gate and no resident index here (the batch lives in host memory).
static cvec_status check_batch_options(const cvec_read_options *opts) {
    cvec_status st = check_options(opts);
    if (st) return st;
    if (opts && (opts->d_gate || opts->indexed)) return CVEC_ERR_BAD_ARGUMENT;
    return CVEC_OK;
}

cvec_status cvec_filter_count_batch_opt(int32_t etype, const uint8_t *h_bytes, const uint64_t *h_byte_off,
                                        uint64_t n_vectors, const cvec_pred *pred, const cvec_read_options *opts,
                                        uint64_t *h_total, uint64_t *h_per_vector, int32_t *h_status) {
    cvec_status st = check_batch_options(opts);
    if (st) return st;
    CallOptions in_force(opts);
    return cvec_filter_count_batch(etype, h_bytes, h_byte_off, n_vectors, pred, h_total, h_per_vector, h_status);
}

cvec_status cvec_decode_batch_opt(int32_t etype, const uint8_t *h_bytes, const uint64_t *h_byte_off, uint64_t n_vectors,
                                  const cvec_sink_chain *chain, const cvec_read_options *opts, void *h_out,
                                  uint64_t out_cap_elems, uint64_t *h_out_off, int32_t *h_status) {
    cvec_status st = check_batch_options(opts);
    if (st) return st;
    CallOptions in_force(opts);
    return cvec_decode_batch(etype, h_bytes, h_byte_off, n_vectors, chain, h_out, out_cap_elems, h_out_off, h_status);
}

static cvec_status encode_pipeline_body(SlotCache<EncSlot> *cache, int32_t etype, const void *h_values,
                                        const uint64_t *h_elem_off, uint64_t n_vectors, uint8_t *h_bytes, size_t bytes_cap,
                                        uint64_t *h_byte_off, uint64_t *h_total_bytes);

cvec_status cvec_encode_batch(int32_t etype, const void *h_values, const uint64_t *h_elem_off, uint64_t n_vectors,
                              uint8_t *h_bytes, size_t bytes_cap, uint64_t *h_byte_off, uint64_t *h_total_bytes) {
    static thread_local SlotCache<EncSlot> *cache = new SlotCache<EncSlot>();
    return drain_pipeline(cache, encode_pipeline_body(cache, etype, h_values, h_elem_off, n_vectors, h_bytes, bytes_cap,
                                                      h_byte_off, h_total_bytes));
}

cvec_status cvec_debug_fail_chunk(int64_t chunk) {
    g_fail_chunk = chunk;
    return CVEC_OK;
}

static cvec_status encode_pipeline_body(SlotCache<EncSlot> *cache, int32_t etype, const void *h_values,
                                        const uint64_t *h_elem_off, uint64_t n_vectors, uint8_t *h_bytes, size_t bytes_cap,
                                        uint64_t *h_byte_off, uint64_t *h_total_bytes) {
    if (!etype_ok(etype) || !h_elem_off || !h_byte_off || !h_bytes) return CVEC_ERR_BAD_ARGUMENT;
    for (uint64_t v = 0; v < n_vectors; v++) {
        if (h_elem_off[v + 1] < h_elem_off[v]) return CVEC_ERR_BAD_ARGUMENT;
        // finish(): total_num_rows > u32::MAX is InvalidNumRows (reference src/vector.rs:347-349)
        if (h_elem_off[v + 1] - h_elem_off[v] > 0xffffffffull) return CVEC_INVALID_NUM_ROWS;
    }
    if (h_elem_off[n_vectors] > h_elem_off[0] && !h_values) return CVEC_ERR_BAD_ARGUMENT;
    const int w = elem_width(etype);
    // chunk weight: the values plus a little per vector (headers, offsets)
    const std::vector<uint64_t> first = plan_chunks(n_vectors, pipe_chunk_bytes(), [&](uint64_t v) {
        return (h_elem_off[v + 1] - h_elem_off[v]) * (uint64_t)w + 64;
    });
    const size_t nchunks = first.size() - 1;
    uint64_t max_vec = 0, max_el = 0;
    for (size_t c = 0; c < nchunks; c++) {
        max_vec = std::max(max_vec, first[c + 1] - first[c]);
        max_el = std::max(max_el, h_elem_off[first[c + 1]] - h_elem_off[first[c]]);
    }
    const size_t nslots = std::min<size_t>(PIPE_SLOTS, std::max<size_t>(nchunks, 1));
    EncSlot *slots = cache->get();
    if (!slots) return hip_fail(hipGetLastError(), "pipeline streams");
    hipStream_t s_in = cache->s_in.s, s_run = cache->s_run.s, s_out = cache->s_out.s;
    const size_t cap = cvec_max_encoded_bytes(etype, max_vec, max_el);
    const size_t wsb = cvec_encode_workspace_bytes(max_vec, max_el);
    for (size_t k = 0; k < nslots; k++) {
        EncSlot &sl = slots[k];
        sl.chunk = sl.leaving = (size_t)-1;
        HIP_TRY(sl.hboff.alloc((max_vec + 1) * 8));
        HIP_TRY(sl.vals.alloc(max_el * w + 16));
        HIP_TRY(sl.eoff.alloc((max_vec + 1) * 8));
        HIP_TRY(sl.bytes.alloc(align_up(cap, 256)));
        HIP_TRY(sl.boff.alloc((max_vec + 1) * 8));
        HIP_TRY(sl.sbase.alloc((max_vec + 1) * 8));
        HIP_TRY(sl.ws.alloc(wsb));
        HIP_TRY(sl.res.alloc(CVEC_RESULT_WORDS * 8));
        HIP_TRY(sl.heoff.alloc((max_vec + 1) * 8));
        HIP_TRY(sl.hres.alloc(CVEC_RESULT_WORDS * 8));
        HIP_TRY(sl.ev_in.make());
        HIP_TRY(sl.ev_run.make());
        HIP_TRY(sl.ev_res.make());
        HIP_TRY(sl.ev_out.make());
        HIP_TRY(hipEventRecord(sl.ev_out.e, s_out)); // nothing to wait for before the slot's first use
    }
    uint64_t total = 0;
    bool overflow = false;
    std::vector<uint64_t> base(nchunks, 0); // blob offset of each chunk's first byte
    // The blob offset of a chunk is the size of everything before it, which the host learns when the
    // earlier chunks have been encoded: a chunk's bytes are fetched when its slot comes up again,
    // PIPE_SLOTS - 1 chunks later; the slot's next encode waits for that copy.
    auto land = [&](EncSlot &sl) -> cvec_status { // the offsets of the chunk whose blob was requested last
        if (sl.leaving == (size_t)-1) return CVEC_OK;
        HIP_TRY(hipEventSynchronize(sl.ev_out.e));
        const size_t c = sl.leaving;
        sl.leaving = (size_t)-1;
        const uint64_t *bo = (const uint64_t *)sl.hboff.p;
        for (uint64_t v = first[c]; v < first[c + 1]; v++) h_byte_off[v] = base[c] + bo[v - first[c]];
        return CVEC_OK;
    };
    auto collect = [&](EncSlot &sl) -> cvec_status {
        cvec_status ls = land(sl);
        if (ls) return ls;
        if (sl.chunk == (size_t)-1) return CVEC_OK;
        HIP_TRY(hipEventSynchronize(sl.ev_res.e));
        const size_t c = sl.chunk;
        sl.chunk = (size_t)-1;
        const uint64_t size = ((const uint64_t *)sl.hres.p)[0];
        const uint64_t nv = first[c + 1] - first[c];
        base[c] = total;
        if (total + size > bytes_cap) overflow = true;
        else {
            if (size) HIP_TRY(hipMemcpyAsync(h_bytes + total, sl.bytes.p, size, hipMemcpyDeviceToHost, s_out));
            if (nv) HIP_TRY(hipMemcpyAsync(sl.hboff.p, sl.boff.p, nv * 8, hipMemcpyDeviceToHost, s_out));
            sl.leaving = c;
        }
        HIP_TRY(hipEventRecord(sl.ev_out.e, s_out));
        total += size;
        return CVEC_OK;
    };
    for (size_t c = 0; c < nchunks; c++) {
        EncSlot &sl = slots[c % nslots];
        cvec_status st = collect(sl);
        if (st) return st;
        // the slot's pinned offsets may still feed its previous copy-in only until that chunk ran: collect() synchronised
        const uint64_t v0 = first[c], nv = first[c + 1] - v0;
        const uint64_t e0 = h_elem_off[v0], n_el = h_elem_off[first[c + 1]] - e0;
        uint64_t *eo = (uint64_t *)sl.heoff.p;
        for (uint64_t i = 0; i <= nv; i++) eo[i] = h_elem_off[v0 + i] - e0;
        if (n_el) HIP_TRY(hipMemcpyAsync(sl.vals.p, (const uint8_t *)h_values + e0 * w, n_el * w, hipMemcpyHostToDevice, s_in));
        HIP_TRY(hipMemcpyAsync(sl.eoff.p, eo, (nv + 1) * 8, hipMemcpyHostToDevice, s_in));
        HIP_TRY(hipEventRecord(sl.ev_in.e, s_in));
        HIP_TRY(hipStreamWaitEvent(s_run, sl.ev_in.e, 0));
        HIP_TRY(hipStreamWaitEvent(s_run, sl.ev_out.e, 0)); // the slot's previous blob has left the device buffer
        st = cvec_encode_dev(etype, sl.vals.p, sl.eoff.as<uint64_t>(), nv, n_el, sl.bytes.as<uint8_t>(), cap,
                             sl.boff.as<uint64_t>(), sl.sbase.as<uint64_t>(), sl.res.as<uint64_t>(), sl.ws.p, wsb, s_run);
        if (st) return st;
        if (g_fail_chunk == (int64_t)c) { g_fail_chunk = -1; return CVEC_ERR_HIP; } // test hook: fail with work in flight
        HIP_TRY(hipEventRecord(sl.ev_run.e, s_run));
        HIP_TRY(hipStreamWaitEvent(s_out, sl.ev_run.e, 0));
        HIP_TRY(hipMemcpyAsync(sl.hres.p, sl.res.p, CVEC_RESULT_WORDS * 8, hipMemcpyDeviceToHost, s_out));
        HIP_TRY(hipEventRecord(sl.ev_res.e, s_out));
        sl.chunk = c;
    }
    // the chunks still in flight, oldest first
    for (size_t k = 0; k < nslots; k++) {
        cvec_status st = collect(slots[(nchunks + k) % nslots]);
        if (st) return st;
    }
    for (size_t k = 0; k < nslots; k++) {
        cvec_status st = land(slots[k]);
        if (st) return st;
    }
    HIP_TRY(hipStreamSynchronize(s_out));
    if (h_total_bytes) *h_total_bytes = total;
    if (overflow) return CVEC_ERR_CAPACITY;
    h_byte_off[n_vectors] = total;
    return CVEC_OK;
}

} // extern "C"
