/*
 * cvec.h — C ABI of the MI355X-native NibblePack / XOR / delta path of compressed_vec.
 *
 * The reference crate (velvia/compressed-vec 0.1.1) exports no FFI (its Cargo.toml:43 builds a
 * cdylib but no `extern "C"` item exists); its seams are Rust traits resolved at compile time.
 * This header is the boundary a maintainer's `ffi.rs` would bind so that the crate keeps its API
 * surface — VectorF32XorAppender::encode_all (src/vector.rs:238), VectorReader::decode_to_sink
 * (src/vector.rs:458), reader.filter_iter(..) + filter::count_hits (src/vector.rs:447,
 * src/filter.rs:296) and the sink stack (src/sink.rs:135-320) — while the work runs in HIP
 * kernels on gfx950.  One 8 KiB vector is far below GPU launch granularity, so every entry point is
 * batch-granular: a batch is n reference-format vectors stored back to back in one byte blob plus
 * `byte_off[n+1]` (vector v occupies bytes [byte_off[v], byte_off[v+1])).  The bytes of each vector
 * are exactly what the reference appender's finish() returns (16-byte header + 256-element
 * sections; src/vector.rs:51-58, :163-186, src/section.rs:32-38).
 *
 * Conventions
 *  - Plain pointers and sizes only.  `d_` parameters are device pointers, `h_` host pointers.
 *  - Every function returns a cvec_status: 0 = OK, 1..8 = the reference's CodingError discriminant
 *    (src/error.rs:2-11), >= 100 = errors of this library (HIP failures, bad arguments).  A call-level
 *    failure (bad argument, workspace or output capacity, HIP error) is always >= 101 and means that no
 *    result was produced; the *_batch entry points return 1..8 or CVEC_ERR_SECTION_COUNT only as "the
 *    largest per-vector status", with every vector's own status in h_status.
 *  - `stream` is a hipStream_t passed as void* (NULL = the default stream).  `_dev` functions only
 *    enqueue work; results are valid after the stream is synchronised.
 *  - No exceptions or panics cross this boundary.
 */
#ifndef CVEC_H
#define CVEC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int32_t cvec_status;

/* src/error.rs:2-11 */
enum {
    CVEC_OK = 0,
    CVEC_NOT_ENOUGH_SPACE = 1,
    CVEC_INPUT_TOO_SHORT = 2,
    CVEC_BAD_OFFSET = 3,
    CVEC_INVALID_SECTION_TYPE = 4,
    CVEC_INVALID_FORMAT = 5,
    CVEC_INVALID_NUM_ROWS = 6,
    CVEC_WRONG_VECTOR_TYPE = 7,
    CVEC_SCROLL_ERR = 8,
    /* library-level */
    CVEC_ERR_SECTION_COUNT = 100, /* sections found != ceil(header.num_elements / 256) */
    CVEC_ERR_BAD_ARGUMENT = 101,
    CVEC_ERR_HIP = 102,           /* a HIP runtime call failed; see cvec_last_hip_error() */
    CVEC_ERR_NO_DEVICE = 103,
    CVEC_ERR_WORKSPACE = 104,     /* workspace too small */
    CVEC_ERR_CAPACITY = 105       /* a caller-provided output buffer is too small for the whole call; nothing was produced
                                     (never a per-vector status: those are 1..8 and CVEC_ERR_SECTION_COUNT) */
};

/* Element type = the vector header's minor-type byte (src/vector.rs:94-96). */
enum { CVEC_U64 = 0x10, CVEC_U32 = 0x11, CVEC_F32 = 0x12 };

/* Predicates of the filter path: EqualsSink (src/filter.rs:129-139) and OneOfSink (:143-177).
 * Values are raw little-endian bits of the element type (f32 compares with IEEE ==). */
enum { CVEC_PRED_EQ = 0, CVEC_PRED_ONEOF = 1 };
#define CVEC_MAX_PRED_VALUES 16
typedef struct cvec_pred {
    int32_t kind;
    uint32_t count; /* 1 for EQ, 1..CVEC_MAX_PRED_VALUES for ONEOF */
    uint64_t values[CVEC_MAX_PRED_VALUES];
} cvec_pred;

/* The sink stack applied to every decoded octet before it is stored (src/sink.rs).  XorSink
 * (src/sink.rs:241-284) is implied by an Xor section and the Delta add by a Delta section, exactly
 * as src/section.rs:707 and :637 do; the caller only chooses whether an AddConstSink
 * (src/sink.rs:289-320) sits in front of the final VecSink (src/sink.rs:151-182). */
typedef struct cvec_sink_chain {
    int32_t add_const;        /* 0 = VecSink only; 1 = AddConstSink(add_const_bits) -> VecSink */
    int32_t _pad;
    uint64_t add_const_bits;  /* raw bits of the element type */
} cvec_sink_chain;

/* A batch resident in device memory.  `sec_base[v]` is the number of 256-element sections of all
 * vectors before v (exclusive scan of ceil(num_elements/256) over the headers), so section k of
 * vector v has the global slot sec_base[v] + k and its decoded elements live at
 * out[256 * (sec_base[v] + k) ...] — VecSink's "whole sections" layout (src/sink.rs:149), vectors
 * back to back.  `d_bytes` must be 16-byte aligned and its allocation readable up to the next
 * multiple of 16 bytes past n_bytes (the kernels stage the blob with 16-byte loads). */
typedef struct cvec_dbatch_view {
    int32_t etype;
    int32_t _pad;
    uint64_t n_vectors;
    uint64_t n_sections;
    uint64_t n_bytes;
    const uint8_t *d_bytes;
    const uint64_t *d_byte_off; /* n_vectors + 1 */
    const uint64_t *d_sec_base; /* n_vectors + 1 */
} cvec_dbatch_view;

/* Result block written by the _dev operations (device memory, 4 x u64):
 * [0] total match count (filter) / total elements stored (decode) / total bytes (encode)
 * [1] number of vectors whose status != OK    [2] largest status code seen    [3] reserved */
#define CVEC_RESULT_WORDS 4

/* ---- library ------------------------------------------------------------------------------ */
const char *cvec_version(void);
/* Selects the device for the calling thread and warms the code object. */
cvec_status cvec_init(int device);
const char *cvec_last_hip_error(void);
const char *cvec_status_name(cvec_status s);

/* Per-kernel timing of cvec_filter_count_dev / cvec_decode_dev for roofline accounting (no
 * counterpart in the reference, whose benches time whole calls with criterion, benches/bench.rs:184).
 * Between begin and end every such call records HIP events around its three kernels on its launch
 * stream (at most max_records calls); end waits for them and returns the summed milliseconds of
 * the section-index, section-decode and finalize kernels.  Not thread-safe; one user at a time. */
cvec_status cvec_timing_begin(int32_t max_records);
cvec_status cvec_timing_end(float *ms_index, float *ms_sections, float *ms_finalize, int32_t *n_records);

/* Which kernel serves cvec_filter_count_dev / cvec_decode_dev (and the *_batch calls built on them).  No
 * counterpart in the reference.  Two kernels produce identical results:
 *   CVEC_PATH_SLOTS    an index pass walks the section headers of every vector, then a kernel decodes 16
 *                      consecutive sections per wavefront pass, then a pass folds per-vector results:
 *                      parallel over sections, any batch shape;
 *   CVEC_PATH_VSTREAM  one kernel; 16 vectors per wavefront side by side, the section chain followed in LDS:
 *                      parallel over vectors, no index, each byte read once (u32 / f32, blob < 4 GiB);
 * CVEC_PATH_AUTO (default) picks VSTREAM for sparse batches (at least 16384 vectors, at most 64 sections per
 * vector and 48 bytes per section on average: mostly Null / Const sections), SLOTS otherwise.  Forcing
 * VSTREAM applies wherever the batch is eligible at all (gated and indexed passes, u64 batches and blobs of 4 GiB or more always take SLOTS).  The setter changes the
 * process-wide default; a single call chooses for itself through cvec_read_options. */
enum { CVEC_PATH_AUTO = 0, CVEC_PATH_SLOTS = 1, CVEC_PATH_VSTREAM = 2 };
cvec_status cvec_set_read_path(int32_t path);

/* Which values the last octets of a vector decode to.  CVEC_TAIL_FORMAT (default): every octet by the
 * bitstream rule of the format, i.e. the values that were encoded.  CVEC_TAIL_REFERENCE: what the reference
 * crate returns: with fewer than 34 bytes of the vector slice left, its 32-bit unpacker
 * (src/nibblepack_simd.rs:511-585; 64-bit octets whose bits fit 32 take it too, src/nibblepacking.rs:431-434)
 * falls back to preload_u32x8_nibbles (:433-450, :546-554), which yields zeros for the fields of octets of
 * 5 to 8 nibbles, so such octets decode as if all their fields were zero (an Xor section repeats its previous
 * values from there on).  Under this rule decode, counts, per-vector counts and masks of the _dev and _batch
 * read operations equal the crate's on well-formed vectors; passes take the section-slot kernels, which clear
 * the masks of such octets as they decode (no extra pass, no measurable cost).  The legacy stream codecs are not
 * affected.  The setter changes the process-wide default; a single call chooses for itself through
 * cvec_read_options. */
enum { CVEC_TAIL_FORMAT = 0, CVEC_TAIL_REFERENCE = 1 };
cvec_status cvec_set_tail_rule(int32_t rule);

/* Options of ONE read call.  The reference's readers are shared by reference between threads (&self,
 * src/vector.rs:392-464) and take everything that shapes a pass as arguments of the call; the two setters above are
 * process-wide defaults, which two threads that want different rules cannot share.  The _opt entry points below take
 * this struct instead: CVEC_OPT_DEFAULT in a field means "the process-wide default", a NULL struct means defaults
 * throughout.  The options hold for that call alone, on the calling thread; concurrent calls with different options
 * do not affect each other. */
#define CVEC_OPT_DEFAULT (-1)
typedef struct cvec_read_options {
    int32_t read_path;     /* CVEC_PATH_* or CVEC_OPT_DEFAULT */
    int32_t tail_rule;     /* CVEC_TAIL_* or CVEC_OPT_DEFAULT */
    int32_t indexed;       /* nonzero: over the resident section index of the workspace (cvec_index_dev), as
                              cvec_filter_count_indexed_dev / cvec_decode_indexed_dev */
    int32_t _pad;
    const uint8_t *d_gate; /* filter passes only, nullable: as cvec_filter_count_gated_dev */
} cvec_read_options;
/* What a call with these options would run with: the fields resolved against the process-wide defaults.
 * CVEC_ERR_BAD_ARGUMENT for a field that is neither a valid value nor CVEC_OPT_DEFAULT.  Touches no device. */
cvec_status cvec_read_options_effective(const cvec_read_options *opts, int32_t *read_path, int32_t *tail_rule);

/* ---- device-resident operations ----------------------------------------------------------- */

/* Bytes of scratch the _dev operations need for a batch of this shape. */
size_t cvec_workspace_bytes(uint64_t n_vectors, uint64_t n_sections);

/* Computes sec_base[n_vectors+1] from the vector headers (u32 num_elements at +8,
 * src/vector.rs:165) on the device; also returns the section total through h_n_sections (this
 * call synchronises the stream).  Done once per batch, like building byte_off.  A vector gets
 * ceil(num_elements / 256) slots, except that a header claiming more sections than the vector has
 * bytes behind its header (a section takes at least one byte) gets that many plus one: such a
 * vector reports CVEC_ERR_SECTION_COUNT, and no buffer is ever sized by a damaged header alone.
 * The *_batch entry points and the decode output layout (256 elements per slot) use the same rule. */
cvec_status cvec_sec_base_dev(const uint8_t *d_bytes, const uint64_t *d_byte_off, uint64_t n_vectors,
                              uint64_t *d_sec_base, uint64_t *h_n_sections, void *stream);

/* reader.filter_iter(pred) + filter::count_hits for every vector of the batch, with the predicate
 * fused into the decode (no decoded vector is written).  Replaces src/vector.rs:447 +
 * src/filter.rs:189-231, :296-299.  d_result: CVEC_RESULT_WORDS u64.  d_per_vector (nullable):
 * n_vectors u64 match counts (counts include the zero padding of the last section, as the
 * reference's do).  d_status (nullable): n_vectors int32 status codes.  d_masks (nullable):
 * n_sections * 32 bytes, the 256-bit mask the reference's VectorFilter yields per section
 * (byte = octet, LSB = first element; src/filter.rs:46, :95-101). */
cvec_status cvec_filter_count_dev(const cvec_dbatch_view *batch, const cvec_pred *pred, uint64_t *d_result,
                                  uint64_t *d_per_vector, int32_t *d_status, uint8_t *d_masks,
                                  void *d_workspace, size_t workspace_bytes, void *stream);

/* OneOfSink over a set of any size: the reference's MembershipPredicate holds a Vec of values
 * (src/filter.rs:143-160) and OneOfIF ORs one comparison per member (:162-175).  values: n_values raw
 * element bits in HOST memory (duplicates, and -0.0 beside +0.0, count once; NaNs equal nothing).  Sets of
 * up to CVEC_MAX_PRED_VALUES distinct values take one pass, like cvec_filter_count_dev; larger sets take
 * one pass per CVEC_MAX_PRED_VALUES values, with counts added and masks OR-ed on the device.  Outputs as
 * for cvec_filter_count_dev; the workspace is cvec_filter_set_workspace_bytes() bytes. */
size_t cvec_filter_set_workspace_bytes(uint64_t n_vectors, uint64_t n_sections);
cvec_status cvec_filter_count_set_dev(const cvec_dbatch_view *batch, const uint64_t *values, uint32_t n_values,
                                      uint64_t *d_result, uint64_t *d_per_vector, int32_t *d_status, uint8_t *d_masks,
                                      void *d_workspace, size_t workspace_bytes, void *stream);

/* cvec_filter_count_dev without a workspace argument and without any allocation: the scratch is a
 * static device array inside the library (CVEC_STATIC_SCRATCH_BYTES per GPU, held until the process
 * ends).  No counterpart in the reference; it exists for callers that cannot allocate or size device
 * memory: code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc.  The
 * call makes no hipMalloc / hipFree and calls no allocating routine.  A batch whose
 * cvec_workspace_bytes(n_vectors, n_sections) exceeds CVEC_STATIC_SCRATCH_BYTES (about 170 000
 * sections) is rejected with CVEC_ERR_WORKSPACE and a message in cvec_last_hip_error().  All calls on
 * one GPU share that GPU's scratch: calls that overlap in time on one GPU (two streams, two threads)
 * race and must be serialised by the caller.  cvec_init(device) must have run for the device (it resolves the
 * address of that GPU's scratch; CVEC_ERR_BAD_ARGUMENT otherwise), so that the call itself touches no loader or
 * allocator and can be recorded by hipStreamBeginCapture.  Results are identical to cvec_filter_count_dev's. */
#define CVEC_STATIC_SCRATCH_BYTES (2 * 1024 * 1024)
cvec_status cvec_filter_count_static_dev(const cvec_dbatch_view *batch, const cvec_pred *pred, uint64_t *d_result,
                                         uint64_t *d_per_vector, int32_t *d_status, uint8_t *d_masks, void *stream);

/* cvec_filter_count_dev / _gated_dev / _indexed_dev with the options of this call (see cvec_read_options). */
cvec_status cvec_filter_count_opt_dev(const cvec_dbatch_view *batch, const cvec_pred *pred, const cvec_read_options *opts,
                                      uint64_t *d_result, uint64_t *d_per_vector, int32_t *d_status, uint8_t *d_masks,
                                      void *d_workspace, size_t workspace_bytes, void *stream);

/* The same, behind a gate: d_gate (nullable, 16-byte aligned) holds n_sections * 32 bytes, the masks
 * an earlier filter pass wrote for a batch with the same sec_base (e.g. the previous column of a
 * MultiVectorFilter).  A section whose gate mask is empty is parsed but not decoded, like the
 * short-circuit of src/filter.rs:277-282 (advance(), :204-209): it yields an empty mask, count 0 and
 * no decode error.  d_masks must be 8-byte aligned. */
cvec_status cvec_filter_count_gated_dev(const cvec_dbatch_view *batch, const cvec_pred *pred, const uint8_t *d_gate,
                                        uint64_t *d_result, uint64_t *d_per_vector, int32_t *d_status,
                                        uint8_t *d_masks, void *d_workspace, size_t workspace_bytes, void *stream);

/* A resident section index.  The reference walks the section headers of a vector on every pass
 * (FixedSectIterator, src/section.rs:866-876) and keeps nothing; the _dev read operations above do
 * the same walk as their first kernel, for the whole batch, into the workspace (one descriptor per
 * section slot + one header status per vector).  These two calls split that: cvec_index_dev runs the
 * walk alone, and cvec_filter_count_indexed_dev is cvec_filter_count_dev without it, over the index
 * that the workspace still holds.  The caller guarantees that the workspace was last written by
 * cvec_index_dev, cvec_filter_count_indexed_dev or cvec_decode_indexed_dev on this same batch and that the
 * batch has not changed since (any other call that uses the workspace may leave it without an index).  Results are identical to cvec_filter_count_dev's; with d_gate (may be NULL)
 * they are cvec_filter_count_gated_dev's: the section kernel itself looks at the gate masks, the
 * index stays the ungated one and remains valid.  The index pass tags the
 * workspace with the batch it belongs to (pointers and shape); an indexed call over a workspace that
 * carries another tag (never indexed, indexed for another batch, last used by a gated pass) follows
 * no descriptor: every vector reports CVEC_ERR_WORKSPACE and the total is 0.  Changed blob contents
 * behind the same pointers are not detected. */
cvec_status cvec_index_dev(const cvec_dbatch_view *batch, void *d_workspace, size_t workspace_bytes, void *stream);
cvec_status cvec_filter_count_indexed_dev(const cvec_dbatch_view *batch, const cvec_pred *pred, const uint8_t *d_gate,
                                          uint64_t *d_result, uint64_t *d_per_vector, int32_t *d_status,
                                          uint8_t *d_masks, void *d_workspace, size_t workspace_bytes, void *stream);
/* cvec_decode_dev over the resident index, under the same conditions. */
cvec_status cvec_decode_indexed_dev(const cvec_dbatch_view *batch, const cvec_sink_chain *chain, void *d_out,
                                    uint64_t *d_result, int32_t *d_status, void *d_workspace,
                                    size_t workspace_bytes, void *stream);

/* VectorReader::decode_to_sink through [AddConstSink ->] VecSink for every vector of the batch.
 * Replaces src/vector.rs:458-464 with the sinks of src/sink.rs.  d_out (16-byte aligned) holds
 * 256 * n_sections elements of the batch's type (layout: see cvec_dbatch_view).
 *
 * Values are decoded by the bitstream rule of the format, i.e. they are the values that were encoded.
 * The reference crate itself does not always return those: when fewer than 34 bytes of the vector
 * slice remain, its 32-bit unpacker falls back to preload_u32x8_nibbles (src/nibblepack_simd.rs:433-450,
 * reached from :546-554, and for 64-bit vectors through src/nibblepacking.rs:431-434 when the octet's
 * bits fit 32), which yields zeros for octets of 5 to 8 nibbles.  Well-formed vectors that end on a
 * section boundary with such an octet (e.g. 2048-element f32 vectors of high-entropy data) therefore
 * decode and filter differently here than in the crate, in those last lanes only; the filter pass
 * follows the same rule as this decode.  cvec_set_tail_rule(CVEC_TAIL_REFERENCE) switches both to the
 * crate's behaviour. */
cvec_status cvec_decode_dev(const cvec_dbatch_view *batch, const cvec_sink_chain *chain, void *d_out,
                            uint64_t *d_result, int32_t *d_status, void *d_workspace, size_t workspace_bytes,
                            void *stream);

/* cvec_decode_dev / cvec_decode_indexed_dev with the options of this call (opts->d_gate must be NULL). */
cvec_status cvec_decode_opt_dev(const cvec_dbatch_view *batch, const cvec_sink_chain *chain, const cvec_read_options *opts,
                                void *d_out, uint64_t *d_result, int32_t *d_status, void *d_workspace,
                                size_t workspace_bytes, void *stream);

/* ---- consumers of the per-section masks ----------------------------------------------------- */

/* The number of leading sections of every vector that the last cvec_filter_count_dev /
 * cvec_decode_dev on `batch` with this workspace produced: the reference's section iteration ends
 * at the first section that fails to decode (src/filter.rs:226) or to parse.  d_valid: n_vectors
 * u32.  Masks / decoded data of later sections are unspecified.  d_parse_failed (nullable): n_vectors
 * bytes, 1 where the iteration ended at the vector's header checks or at a section header that does not
 * parse (where the reference's VectorFilter panics, src/filter.rs:221, instead of just ending). */
cvec_status cvec_valid_sections_dev(const cvec_dbatch_view *batch, const void *d_workspace, size_t workspace_bytes,
                                    uint32_t *d_valid, uint8_t *d_parse_failed, void *stream);

/* The payload of the reference's CodingError (src/error.rs:2-11) for every failed vector of the pass
 * before, and where its iteration stopped.  d_status: the statuses that pass wrote; d_valid /
 * d_parse_failed: what cvec_valid_sections_dev made of its workspace.  d_detail: 2 * n_vectors u64;
 * for vector v, d_detail[2v] = index of the section at which the iteration stopped (all ones: the vector
 * header, VectorReader::try_new, src/vector.rs:404-414) and d_detail[2v+1] = the offset of BadOffset(usize),
 * the type byte of InvalidSectionType(u8), the subtype byte of WrongVectorType(u8); for every other code
 * the byte offset, inside the vector, of the section where it stopped (0 for the header).  Both words
 * are 0 for a vector whose status is CVEC_OK.  (InvalidFormat and ScrollErr carry strings in the
 * reference; the section index and offset stand in for them.) */
cvec_status cvec_error_detail_dev(const cvec_dbatch_view *batch, const int32_t *d_status, const uint32_t *d_valid,
                                  const uint8_t *d_parse_failed, uint64_t *d_detail, void *stream);

/* One filtered vector column: the masks cvec_filter_count_dev wrote for a batch, that batch's
 * sec_base and the valid-section counts from cvec_valid_sections_dev. */
#define CVEC_MAX_COLUMNS 16
typedef struct cvec_mask_column {
    const uint8_t *d_masks;      /* n_sections * 32 bytes */
    const uint64_t *d_sec_base;  /* n_vectors + 1 */
    const uint32_t *d_valid;     /* n_vectors */
    uint64_t n_sections;
} cvec_mask_column;

/* MultiVectorFilter + count_hits (src/filter.rs:239-286, :296-299) for n_rows rows of k columns:
 * row r ANDs the masks of vector r of every column section by section, consulting column c only
 * while the running mask is non-zero, and ends where column 0 or a consulted column has no section
 * left.  d_result[0] = total; d_per_row (nullable) n_rows u64; d_row_sections (nullable) n_rows u32
 * = masks yielded per row; d_masks_out (nullable) the ANDed masks in column 0's slot layout. */
size_t cvec_multi_workspace_bytes(uint64_t n_sections_col0);
cvec_status cvec_multi_filter_count_dev(const cvec_mask_column *cols, uint32_t k, uint64_t n_rows,
                                        uint64_t *d_result, uint64_t *d_per_row, uint32_t *d_row_sections,
                                        uint8_t *d_masks_out, void *d_workspace, size_t workspace_bytes,
                                        void *stream);

/* filter::match_positions (src/filter.rs:302-321) for every vector of a column: the element
 * positions (relative to the vector) of the set mask bits, ascending, vectors back to back:
 * vector v's positions are d_positions[d_pos_off[v] .. d_pos_off[v+1]).  Unlike the reference's
 * loop these are true positions (its `pos` is not advanced over all-zero 32-bit mask words).
 * Positions beyond positions_cap are dropped; d_pos_off[n_vectors] is the number needed. */
size_t cvec_positions_workspace_bytes(uint64_t n_sections);
cvec_status cvec_match_positions_dev(const cvec_mask_column *col, uint64_t n_vectors, uint64_t *d_pos_off,
                                     uint32_t *d_positions, uint64_t positions_cap, void *d_workspace,
                                     size_t workspace_bytes, void *stream);

/* Vector{U32,U64,F32Xor}Appender::encode_all for n_vectors vectors: vector v's elements are
 * d_values[elem_off[v] .. elem_off[v+1]).  Replaces src/vector.rs:238-246 (append → encode_section
 * → finish) with AutoEncoder (src/section.rs:819-846) for u32/u64 and XorNPMedFixedSect
 * (src/section.rs:716-754) for f32.  Writes the blob to d_bytes (capacity bytes_cap, must be
 * >= cvec_max_encoded_bytes, else CVEC_ERR_CAPACITY), d_byte_off[n+1] and d_sec_base[n+1]; d_result[0] = blob size. */
size_t cvec_max_encoded_bytes(int32_t etype, uint64_t n_vectors, uint64_t n_elements_total);
size_t cvec_encode_workspace_bytes(uint64_t n_vectors, uint64_t n_elements_total);
cvec_status cvec_encode_dev(int32_t etype, const void *d_values, const uint64_t *d_elem_off, uint64_t n_vectors,
                            uint64_t n_elements_total, uint8_t *d_bytes, size_t bytes_cap, uint64_t *d_byte_off,
                            uint64_t *d_sec_base, uint64_t *d_result, void *d_workspace, size_t workspace_bytes,
                            void *stream);

/* ---- legacy NibblePack stream codecs (batches of streams) -------------------------------------
 * CVEC_STREAM_U64_PLAIN is pack_u64 / unpack themselves (src/nibblepacking.rs:68-93, :364-377):
 * the values as they are, no delta or XOR. */

/* pack_u64_delta (src/nibblepacking.rs:15-23: deltas clipped at 0, then pack_u64 :68-93) or
 * pack_f64_xor (:29-45: first value raw, then XOR with the predecessor) for n_streams streams;
 * stream s's values are d_values[elem_off[s] .. elem_off[s+1]) as 64-bit patterns.  Writes the
 * streams back to back into d_bytes (16-byte aligned, capacity >= cvec_stream_max_packed_bytes) and
 * d_byte_off[n_streams+1]; d_result[0] = bytes written.  d_status (nullable): per stream, only
 * InputTooShort for an empty XOR stream (src/nibblepacking.rs:37). */
enum { CVEC_STREAM_U64_DELTA = 1, CVEC_STREAM_F64_XOR = 2, CVEC_STREAM_U64_PLAIN = 3 };
size_t cvec_stream_max_packed_bytes(int32_t kind, uint64_t n_streams, uint64_t n_values_total);
size_t cvec_stream_pack_workspace_bytes(uint64_t n_streams, uint64_t n_values_total);
cvec_status cvec_stream_pack_dev(int32_t kind, const uint64_t *d_values, const uint64_t *d_elem_off,
                                 uint64_t n_streams, uint64_t n_values_total, uint8_t *d_bytes, size_t bytes_cap,
                                 uint64_t *d_byte_off, int32_t *d_status, uint64_t *d_result, void *d_workspace,
                                 size_t workspace_bytes, void *stream);

/* unpack into a DeltaSink (src/nibblepacking.rs:239-279, :364-377: running wrapping sum) or
 * unpack_f64_xor into a DoubleXorSink (:284-325, :392-400: running XOR seeded with the raw first
 * value): stream s decodes elem_off[s+1] - elem_off[s] values into d_out[elem_off[s] ...] (the
 * zero padding of the last octet is not written).  d_bytes as for a batch (16-byte aligned, padded
 * allocation).  d_status: n_streams int32; d_result[0] = values written, [1] = failed streams. */
size_t cvec_stream_unpack_workspace_bytes(uint64_t n_streams, uint64_t n_values_total);
cvec_status cvec_stream_unpack_dev(int32_t kind, const uint8_t *d_bytes, const uint64_t *d_byte_off, uint64_t n_bytes,
                                   const uint64_t *d_elem_off, uint64_t n_streams, uint64_t n_values_total,
                                   uint64_t *d_out, int32_t *d_status, uint64_t *d_result, void *d_workspace,
                                   size_t workspace_bytes, void *stream);

/* ---- histograms ------------------------------------------------------------------------------ */

/* DeltaDiffPackSink (src/histogram.rs:92-171) for batches of histogram series, driven the way the
 * reference drives it (examples/2d_delta_bench.rs:53-57): for every histogram
 * unpack(bytes, sink, num_buckets) then sink.finish(), with sink.reset() at the start of a series.
 * Input: n_hist NibblePack streams of num_buckets u64 each (pack_u64 / pack_u64_delta output:
 * histogram h is d_bytes[d_byte_off[h] .. d_byte_off[h+1])); series s owns the consecutive
 * histograms d_series_off[s] .. d_series_off[s+1] (d_series_off[0] = 0, [n_series] = n_hist).
 * Output: for every histogram the NibblePack stream of its bucket-wise difference to the previous
 * histogram of the series (to zero for the first); a bucket below its predecessor is passed through
 * and sets d_dropped[h] (the sink's value_dropped before finish() clears it).  As in the reference,
 * finish() packs the last octet a second time, so every stream holds ceil(num_buckets/8) + 1 octets.
 * A histogram that fails to unpack gets its status in d_status[h] and an empty stream, and so does
 * every later histogram of its series.  d_result[0] = bytes written, [1] = failed histograms.
 * d_bytes and d_out 16-byte aligned (d_bytes padded as for a batch). */
size_t cvec_hist_max_packed_bytes(uint64_t n_hist, uint32_t num_buckets);
size_t cvec_hist_workspace_bytes(uint64_t n_hist, uint32_t num_buckets);
cvec_status cvec_hist_delta_diff_dev(const uint8_t *d_bytes, const uint64_t *d_byte_off, uint64_t n_bytes,
                                     const uint64_t *d_series_off, uint64_t n_series, uint64_t n_hist,
                                     uint32_t num_buckets, uint8_t *d_out, size_t out_cap, uint64_t *d_out_off,
                                     uint32_t *d_dropped, int32_t *d_status, uint64_t *d_result, void *d_workspace,
                                     size_t workspace_bytes, void *stream);

/* ---- host-buffer entry points (what a Rust ffi.rs shim behind the unchanged crate API calls) -- */
/* The batch is cut into chunks of whole vectors (about 64 MiB; the environment variable
 * CVEC_PIPE_CHUNK_BYTES, read per call, overrides) that rotate through three slots of device
 * buffers and pass from a copy-in stream to a kernel stream to a copy-out stream, so the upload
 * of one chunk, the kernels of another and the download of a third overlap and both PCIe
 * directions are busy at once.  Page-locked caller buffers (hipHostMalloc / hipHostRegister) get
 * the full overlap; copies from or to pageable memory are staged by the runtime and block.
 * The slots, their pinned staging and the three streams are kept per calling thread and device
 * and only grow.  Results equal those of one _dev call over the whole batch.  These replace the
 * per-vector calls of the crate API (src/vector.rs:238-246 encode_all, :447 filter_iter +
 * src/filter.rs:296 count_hits, src/vector.rs:458-464 decode_to_sink) for n_vectors at a time. */

cvec_status cvec_filter_count_batch(int32_t etype, const uint8_t *h_bytes, const uint64_t *h_byte_off,
                                    uint64_t n_vectors, const cvec_pred *pred, uint64_t *h_total,
                                    uint64_t *h_per_vector /*nullable*/, int32_t *h_status /*nullable*/);

/* h_out: 256 * (sum of ceil(num_elements/256)) elements; h_out_off (nullable, n+1) receives each
 * vector's first element index in h_out. */
cvec_status cvec_decode_batch(int32_t etype, const uint8_t *h_bytes, const uint64_t *h_byte_off,
                              uint64_t n_vectors, const cvec_sink_chain *chain, void *h_out, uint64_t out_cap_elems,
                              uint64_t *h_out_off /*nullable*/, int32_t *h_status /*nullable*/);

cvec_status cvec_encode_batch(int32_t etype, const void *h_values, const uint64_t *h_elem_off, uint64_t n_vectors,
                              uint8_t *h_bytes, size_t bytes_cap, uint64_t *h_byte_off /* n+1 */,
                              uint64_t *h_total_bytes);

/* The two read calls with the options of this call (read path and tail rule; opts->d_gate and opts->indexed must be
 * zero: the batch lives in host memory). */
cvec_status cvec_filter_count_batch_opt(int32_t etype, const uint8_t *h_bytes, const uint64_t *h_byte_off,
                                        uint64_t n_vectors, const cvec_pred *pred, const cvec_read_options *opts,
                                        uint64_t *h_total, uint64_t *h_per_vector /*nullable*/, int32_t *h_status /*nullable*/);
cvec_status cvec_decode_batch_opt(int32_t etype, const uint8_t *h_bytes, const uint64_t *h_byte_off, uint64_t n_vectors,
                                  const cvec_sink_chain *chain, const cvec_read_options *opts, void *h_out,
                                  uint64_t out_cap_elems, uint64_t *h_out_off /*nullable*/, int32_t *h_status /*nullable*/);

/* ---- appender handle: VectorAppender<T, W> (src/vector.rs:207-379) ------------------------------ */
/* What the crate's VectorU32Appender / VectorU64Appender / VectorF32XorAppender do for one vector:
 * values and nulls are appended, finish() returns the vector's bytes.  The handle lives on the host, one vector
 * being far below GPU launch granularity, and like the reference's appender (a section is encoded when it fills,
 * src/vector.rs:270-281) it holds encoded sections, not the values that went into them: complete sections wait as
 * raw values until 1024 of them have gathered, then that run is encoded on the GPU in one call and only its bytes
 * stay.  Beside them the handle keeps the open section (the reference's write_buf) and the runs of explicit null
 * sections as counts (append_nulls(n) costs no memory per null section); finish() and reader() encode what is
 * still raw and write the null-section count into the header (num_null_sections, src/vector.rs:160-169, :321-327).
 * If the encoder is not available when a run is due (no device), the run stays raw and append still succeeds; the
 * error surfaces in finish() / reader().  Not thread-safe: one handle per thread, like the reference's &mut self. */
typedef struct cvec_appender cvec_appender;
/* try_new (src/vector.rs:223-234): initial_capacity < 8 is CVEC_NOT_ENOUGH_SPACE, as the header write
 * into a buffer that small fails in the reference. */
cvec_status cvec_appender_new(int32_t etype, size_t initial_capacity, cvec_appender **out);
void cvec_appender_free(cvec_appender *a);
cvec_status cvec_appender_reset(cvec_appender *a);                                  /* :254-262 */
uint64_t cvec_appender_num_elements(const cvec_appender *a);                        /* :381-384 */
/* append (src/vector.rs:300-307) for n values in host memory (element type of the appender). */
cvec_status cvec_appender_append(cvec_appender *a, const void *values, size_t n);
cvec_status cvec_appender_append_nulls(cvec_appender *a, size_t num_nulls);         /* :311-335 */
/* finish (src/vector.rs:344-370): pads with nulls to total_num_rows, writes the vector into out and
 * resets the appender.  *len receives the vector's size; with out == NULL or cap too small the call
 * returns CVEC_ERR_CAPACITY, *len says what is needed and the appender keeps its contents.
 * CVEC_INVALID_NUM_ROWS when more values were appended than total_num_rows or total_num_rows exceeds
 * u32; detail (nullable) then receives the two numbers of the reference's InvalidNumRows(usize, usize). */
cvec_status cvec_appender_finish(cvec_appender *a, uint64_t total_num_rows, uint8_t *out, size_t cap, size_t *len,
                                 uint64_t detail[2]);
/* reader (src/vector.rs:372-378): the bytes of the vector as written so far, i.e. its complete
 * sections; values in the open section are not visible yet.  Same out / cap / len rules as finish. */
cvec_status cvec_appender_reader(cvec_appender *a, uint8_t *out, size_t cap, size_t *len);
/* Host memory the handle holds for the vector: bytes of the sections encoded so far, and bytes of values that are
 * still raw (complete sections waiting for their run, and the open section). */
cvec_status cvec_appender_stats(const cvec_appender *a, uint64_t *encoded_bytes, uint64_t *raw_value_bytes);

/* ---- series appender: one VectorAppender<T, W> per series, all of them resident on the device ------- */
/* What n_series appenders of the crate (src/vector.rs:207-379) do when they are driven side by side, as a
 * time-series store does on every scrape: each series keeps an open section of up to 255 raw values
 * (write_buf, :300-307) and the encoded bytes of its complete sections.  The handle holds both in device
 * memory for all series: an append call that completes sections encodes exactly those sections, for every
 * series at once, and keeps the result at its encoded size; the raw values of a call are not kept.  The
 * handle belongs to the device that was current at _new; the calls below are synchronous on `stream`
 * (they read a few counters back) and, like the reference's &mut self, not thread-safe per handle.
 * Bytes, element counts and null-section counts equal those of n_series reference appenders that receive
 * the same calls.  A call that fails with CVEC_ERR_HIP may have applied part of its effect: reset the handle. */
typedef struct cvec_series_appender cvec_series_appender;
cvec_status cvec_series_appender_new(int32_t etype, uint64_t n_series, cvec_series_appender **out);
void cvec_series_appender_free(cvec_series_appender *a);
cvec_status cvec_series_appender_reset(cvec_series_appender *a);                    /* :254-262 */
/* append (src/vector.rs:300-307), many values per call: series v receives d_values[d_elem_off[v] ..
 * d_elem_off[v+1]); with d_elem_off == NULL every series receives per_series values, d_values being
 * n_series x per_series row-major.  Element type of the handle, device memory. */
cvec_status cvec_series_appender_append_dev(cvec_series_appender *a, const void *d_values,
                                            const uint64_t *d_elem_off /*nullable*/, uint64_t per_series, void *stream);
/* The same with values (and offsets, starting at 0) in host memory: copied to the device, then appended.
 * Page-locked memory makes the copy asynchronous up to the call's own synchronisation. */
cvec_status cvec_series_appender_append_host(cvec_series_appender *a, const void *h_values,
                                             const uint64_t *h_elem_off /*nullable*/, uint64_t per_series, void *stream);
/* append_nulls (src/vector.rs:311-335): series v receives d_elem_off[v+1] - d_elem_off[v] nulls, or
 * per_series with d_elem_off == NULL.  Zeros first fill the open section; whole runs of 256 behind an empty
 * open section are explicit null sections, counted in the header. */
cvec_status cvec_series_appender_append_nulls_dev(cvec_series_appender *a, const uint64_t *d_elem_off /*nullable*/,
                                                  uint64_t per_series, void *stream);
/* finish (src/vector.rs:344-370, reader == 0) or reader (:372-378, reader != 0) in two steps, because the
 * caller owns the output.  _seal fixes the contents and reports the size of the resulting batch: for
 * finish every series is padded with nulls to total_num_rows (CVEC_INVALID_NUM_ROWS with the reference's
 * two numbers in detail when a series holds more, or total_num_rows exceeds u32; the handle then stays
 * open); for reader only complete sections are visible.  _emit_dev writes the blob, byte_off[n+1] and
 * sec_base[n+1] of a batch that the read entry points take as it is; after a finish the handle is empty
 * again (:368), after a reader it goes on where it was.  No append between the two calls; _unseal gives
 * up a seal without emitting. */
cvec_status cvec_series_appender_seal(cvec_series_appender *a, int32_t reader, uint64_t total_num_rows,
                                      uint64_t *n_bytes, uint64_t *n_sections, uint64_t detail[2] /*nullable*/,
                                      void *stream);
cvec_status cvec_series_appender_emit_dev(cvec_series_appender *a, uint8_t *d_bytes, size_t bytes_cap,
                                          uint64_t *d_byte_off, uint64_t *d_sec_base, void *stream);
cvec_status cvec_series_appender_unseal(cvec_series_appender *a);
/* Device memory held: number of encoded segments (an append call that completes sections adds one; sixteen of a
 * kind are merged into one, so the number stays small however many calls there were), their bytes, and the bytes of
 * open sections and offsets. */
cvec_status cvec_series_appender_stats(const cvec_series_appender *a, uint64_t *n_segments, uint64_t *encoded_bytes,
                                       uint64_t *state_bytes);

/* ---- across GPUs: the all-reduce of result blocks -------------------------------------------------- */
/* Vectors are independent (each rank filters its own shard; nothing of the data path crosses GPUs), so the only
 * exchange of a sharded query is the sum of the ranks' result blocks: count_hits (src/filter.rs:296-299) over a
 * batch that is spread over several GPUs.  d_result: n_blocks consecutive result blocks (CVEC_RESULT_WORDS u64 each)
 * in device memory, reduced in place over the ranks of `comm`: word 0 (total) and word 1 (failed vectors) are
 * summed, word 2 (largest status) takes the maximum, word 3 is summed.  `comm` is an RCCL communicator
 * (ncclComm_t) that the caller owns - its own, or one from cvec_comm_init_rank; the collective is enqueued on
 * `stream` like any _dev operation (one grouped launch for all blocks).  The library binds librccl at the first of
 * these calls (the copy the process has loaded already, else the ROCm installation's; CVEC_RCCL_PATH overrides) and
 * does not depend on it otherwise.  RCCL failures come back as CVEC_ERR_HIP with the text in cvec_last_hip_error(). */
cvec_status cvec_allreduce_counts(uint64_t *d_result, uint32_t n_blocks, void *comm, void *stream);
/* For callers without RCCL bindings of their own: ncclGetUniqueId (one rank calls it and hands the 128 bytes to the
 * others by any means), ncclCommInitRank (every rank, with its device current; blocks until all ranks have joined)
 * and ncclCommDestroy. */
typedef struct cvec_comm_id { char bytes[128]; } cvec_comm_id;
cvec_status cvec_comm_unique_id(cvec_comm_id *id);
cvec_status cvec_comm_init_rank(const cvec_comm_id *id, int32_t n_ranks, int32_t rank, void **comm);
cvec_status cvec_comm_destroy(void *comm);

/* Test hook, no counterpart in the reference: makes the next *_batch call of the calling thread fail
 * with CVEC_ERR_HIP at chunk `chunk` (0-based) after that chunk's copies and kernels have been queued,
 * to exercise the error exit of the pipeline (which must wait for everything in flight before it
 * returns).  -1 disarms.  The hook disarms itself when it fires. */
cvec_status cvec_debug_fail_chunk(int64_t chunk);

#ifdef __cplusplus
}
#endif
#endif /* CVEC_H */
