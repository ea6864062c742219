/*
 * cvec_oracle.h — CPU restatement of compressed_vec's NibblePack / XOR / delta path.
 *
 * TEST INFRASTRUCTURE ONLY.  This library is the parity checker and the CPU baseline;
 * nothing in the shipped GPU path may call it.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg load it.
 *
 * It is a plain-C, scalar, bounds-checked restatement of the algorithm in the reference
 * crate (velvia/compressed-vec 0.1.1); each function cites the reference file:line it
 * follows.  The reference itself cannot be compiled here (no Rust toolchain), so the
 * oracle is pinned by the known-answer tests, size/choice asserts and round-trip inputs
 * that the reference's own in-source tests hold (transcribed as data under tests/golden/).
 *
 * Deliberate differences from the reference's *behaviour on malformed input* (never on
 * well-formed vectors):
 *   - u32 octets with 5..8 nibbles decoded near the end of a slice use the plain bitstream
 *     rule; the reference's tail fallback (nibblepack_simd.rs:433-450) drops the data.
 *     ocv_set_ref_tail_quirk(1) reproduces the reference behaviour for documentation.
 *   - truncated octets return NOT_ENOUGH_SPACE where the reference can panic
 *     (nibblepacking.rs:474, nibblepack_simd.rs:565).
 *   - section length checks compare in 64 bits (reference truncates the slice length to
 *     u16: section.rs:537, :616, :694).
 *   - an octet must end inside its section's declared length.
 *   - an Xor section whose declared total is < 3 is INVALID_FORMAT (the reference would
 *     never advance).
 */
#ifndef CVEC_ORACLE_H
#define CVEC_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* CodingError discriminants — error.rs:2-11 (0 = Ok). */
enum {
    OCV_OK = 0,
    OCV_NOT_ENOUGH_SPACE = 1,
    OCV_INPUT_TOO_SHORT = 2,
    OCV_BAD_OFFSET = 3,
    OCV_INVALID_SECTION_TYPE = 4,
    OCV_INVALID_FORMAT = 5,
    OCV_INVALID_NUM_ROWS = 6,
    OCV_WRONG_VECTOR_TYPE = 7,
    OCV_SCROLL_ERR = 8
};

/* Element types; values are the vector header's minor type byte — vector.rs:94-96. */
enum { OCV_U64 = 0x10, OCV_U32 = 0x11, OCV_F32 = 0x12 };

/* Section type bytes — section.rs:32-38. */
enum { OCV_SECT_NULL = 0, OCV_SECT_NP = 1, OCV_SECT_DELTA = 3, OCV_SECT_CONST = 5, OCV_SECT_XOR = 6 };

/* Predicate kinds for the filter path — filter.rs:129-177. */
enum { OCV_PRED_EQ = 0, OCV_PRED_ONEOF = 1 };

/* ---- L1: octet codec ---------------------------------------------------------------- */

/* nibble_pack8 (nibblepacking.rs:108-153).  Returns the new offset, or -(error code). */
int64_t ocv_nibble_pack8(const uint64_t in[8], uint8_t *out, size_t cap, size_t off);
/* pack_u64 (nibblepacking.rs:68-93): n values, zero-padded to a multiple of 8. */
int64_t ocv_pack_u64(const uint64_t *vals, size_t n, uint8_t *out, size_t cap, size_t off);
/* nibble_unpack8 (nibblepacking.rs:414-476).  Returns error code; *consumed = bytes used. */
int ocv_nibble_unpack8(const uint8_t *in, size_t len, uint64_t out[8], size_t *consumed);
/* unpack8_u32_simd (nibblepack_simd.rs:511-567), as the plain bitstream rule. *zeroes=1 when
 * the octet was the single 0x00 byte (process_zeroes call). */
int ocv_unpack8_u32(const uint8_t *in, size_t len, uint32_t out[8], int *zeroes, size_t *consumed);
/* unpack (nibblepacking.rs:364-377) into a plain array of round-up-to-8 values. */
int ocv_unpack_u64(const uint8_t *in, size_t len, uint64_t *out, size_t nvalues, size_t *consumed);
/* Legacy stream codecs: pack_u64_delta (:15-23) + DeltaSink (:239-279); pack_f64_xor (:29-45)
 * + DoubleXorSink / unpack_f64_xor (:284-325, :392-400). */
int64_t ocv_pack_u64_delta(const uint64_t *vals, size_t n, uint8_t *out, size_t cap);
int ocv_unpack_u64_delta(const uint8_t *in, size_t len, uint64_t *out, size_t nvalues, size_t *consumed);
int64_t ocv_pack_f64_xor(const double *vals, size_t n, uint8_t *out, size_t cap);
int ocv_unpack_f64_xor(const uint8_t *in, size_t len, double *out, size_t nvalues, size_t *consumed);

/* Histograms: compress_geom_nonincreasing (histogram.rs:55-76) and DeltaDiffPackSink (:92-171) fed by
 * unpack(in, sink, num_buckets) + finish().  `last` is the sink's last_hist_deltas. */
int64_t ocv_compress_geom_nonincreasing(uint16_t num_buckets, double initial_bucket, double multiplier,
                                        uint8_t format_code, const uint64_t *values, size_t n_values, uint8_t *out,
                                        size_t cap);
int ocv_delta_diff_pack(const uint8_t *in, size_t len, size_t num_buckets, uint64_t *last, uint8_t *out, size_t cap,
                        size_t *out_off, int *value_dropped, size_t *consumed);

void ocv_set_ref_tail_quirk(int on);

/* ---- L3: 256-element sections --------------------------------------------------------- */

/* FixedSectionWriter::gen_stats_and_write for the default writers: AutoEncoder for u32/u64
 * (section.rs:821-846), XorNPMedFixedSect for f32 (:716-754).  `vals` holds 256 elements of
 * the given type.  Returns new offset or -(error). */
int64_t ocv_section_write(int etype, const void *vals, uint8_t *out, size_t cap, size_t off);
/* Explicit writers (section.rs:570-586, :650-668, :795-803, :440-443). kind = section type byte. */
int64_t ocv_section_write_kind(int etype, int kind, const void *vals, uint8_t *out, size_t cap, size_t off);
/* FixedSectEnum::try_from (section.rs:252-271): parse one section header at in[0..len).
 * Outputs type byte and num_bytes(). */
int ocv_section_parse(int etype, const uint8_t *in, size_t len, int *kind, size_t *num_bytes);
/* FixedSectEnum::decode into a Section256Sink: writes 256 elements. */
int ocv_section_decode(int etype, const uint8_t *in, size_t len, void *out256);
/* DeltaNPMedFixedSect header fields (section.rs:598-628). */
int ocv_section_delta_info(int etype, const uint8_t *in, size_t len, uint64_t *base, int *numbits);

/* ---- L4: vectors ---------------------------------------------------------------------- */

typedef struct ocv_appender ocv_appender;
ocv_appender *ocv_appender_new(int etype, size_t initial_capacity);   /* vector.rs:223 */
void ocv_appender_free(ocv_appender *a);
int ocv_appender_append(ocv_appender *a, const void *value);           /* vector.rs:300 */
int ocv_appender_append_many(ocv_appender *a, const void *values, size_t n);
int ocv_appender_append_nulls(ocv_appender *a, size_t n);              /* vector.rs:311 */
size_t ocv_appender_num_elements(const ocv_appender *a);               /* vector.rs:249 */
/* reader(): bytes written so far (vector.rs:375-378). Pointer valid until next mutation. */
const uint8_t *ocv_appender_bytes(const ocv_appender *a, size_t *len);
/* finish (vector.rs:344-370): returns length of the finished vector (or -(error)); bytes are
 * copied to out (cap checked). The appender is reset. err_detail[2] receives InvalidNumRows args. */
int64_t ocv_appender_finish(ocv_appender *a, size_t total_rows, uint8_t *out, size_t cap);
/* encode_all (vector.rs:238-246) with a fresh appender. */
int64_t ocv_encode_all(int etype, const void *values, size_t n, uint8_t *out, size_t cap);
/* Upper bound on encoded size for n elements of etype. */
size_t ocv_max_encoded_size(int etype, size_t n);

/* VectorReader::try_new (vector.rs:404-414).  detail receives the WrongVectorType byte. */
int ocv_reader_check(int etype, const uint8_t *bytes, size_t len, uint64_t *detail);
int ocv_error_detail(int etype, const uint8_t *bytes, size_t len, uint64_t detail[2]);
int ocv_reader_num_elements(const uint8_t *bytes, size_t len, uint32_t *n, uint16_t *null_sections);
/* sect_iter().count() and per-section kinds (up to max_kinds) — vector.rs:442, section.rs:864-876. */
int ocv_reader_sections(int etype, const uint8_t *bytes, size_t len, size_t *count,
                        uint8_t *kinds, size_t max_kinds);

/* decode_to_sink (vector.rs:458-464) through [AddConstSink(add_base) →] VecSink.
 * use_add = 0: VecSink only.  out receives the padded element count (multiple of 256). */
int ocv_decode_to_vec(int etype, const uint8_t *bytes, size_t len, int use_add, const void *add_base,
                      void *out, size_t out_cap_elems, size_t *n_out);
/* iterate() (vector.rs:452, :512-556): num_elements values. */
int ocv_iterate(int etype, const uint8_t *bytes, size_t len, void *out, size_t out_cap_elems, size_t *n_out);

/* filter_iter + count_hits (vector.rs:447, filter.rs:189-231, :296-299).
 * pred_vals: npred values of etype (1 for EQ). masks (optional): 32 bytes per section yielded. */
int ocv_filter_count(int etype, const uint8_t *bytes, size_t len, int pred_kind, const void *pred_vals,
                     size_t npred, uint64_t *count, uint8_t *masks, size_t max_sections, size_t *n_sections);
/* match_positions (filter.rs:302-321).  ref_quirk=1 reproduces the reference's skipped-word
 * position bug; 0 gives true positions. */
int ocv_match_positions(const uint8_t *masks, size_t n_sections, int ref_quirk, uint64_t *pos,
                        size_t max_pos, size_t *n_pos);
/* MultiVectorFilter (filter.rs:239-286) over k vectors of one type with one predicate each
 * (EQ values in pred_vals[k]).  Yields ANDed masks + count. */
int ocv_multi_filter_count(int etype, const uint8_t *const *bytes, const size_t *lens, size_t k,
                           const void *pred_vals, uint64_t *count, uint8_t *masks, size_t max_sections,
                           size_t *n_sections);

/* ---- batch helpers used by bench.py's cpu_baseline leg and the parity tests ----------- */
/* Runs ocv_filter_count over n concatenated vectors (byte_off[n+1]); total count returned.
 * per_vec (optional) receives each vector's count, status (optional) each vector's code. */
int ocv_batch_filter_count(int etype, const uint8_t *blob, const uint64_t *byte_off, size_t n,
                           int pred_kind, const void *pred_vals, size_t npred,
                           uint64_t *total, uint64_t *per_vec, int32_t *status);
int ocv_batch_decode(int etype, const uint8_t *blob, const uint64_t *byte_off, size_t n, int use_add,
                     const void *add_base, void *out, const uint64_t *out_off, int32_t *status);
/* Encodes n vectors (elem_off[n+1] into values) back to back; writes byte_off[n+1]. */
int64_t ocv_batch_encode(int etype, const void *values, const uint64_t *elem_off, size_t n,
                         uint8_t *out, size_t cap, uint64_t *byte_off);

#ifdef __cplusplus
}
#endif
#endif
