/*
 * cvec_oracle.c — CPU restatement (plain C, scalar, bounds-checked) of the reference crate's
 * NibblePack codec, fixed 256-element sections, vector container, sink stack and filter path.
 *
 * TEST INFRASTRUCTURE ONLY (see cvec_oracle.h).  Citations are file:line in the reference
 * crate velvia/compressed-vec 0.1.1 (src/...).
 */
#include "cvec_oracle.h"

#include <stdlib.h>
#include <string.h>

#define FIXED_LEN 256            /* section.rs:184 */
#define HEADER_BYTES 16          /* vector.rs:110 NUM_HEADER_BYTES_TOTAL */
#define DELTA_HDR 12             /* section.rs:607 */
#define GROW_BYTES 4096          /* vector.rs:188 */

/* Documentation mode: decode the way the reference's tail fallback does (nibblepack_simd.rs:433-450
 * returns zeros for a 5..8-nibble octet when fewer than 34 bytes are left in the slice it is handed,
 * :546-554).  That slice is the rest of the whole VECTOR slice, not of the section: the section
 * readers keep `sect_bytes` = everything from the section's type byte to the end of the bytes given to
 * VectorReader::try_new (section.rs:548, :636, :706; vector.rs:443, :448).  t_slice_tail = bytes of
 * that slice behind the section being decoded (0 for calls that are handed a bare section or octet). */
static int g_ref_tail_quirk = 0;
static _Thread_local size_t t_slice_tail = 0;
void ocv_set_ref_tail_quirk(int on) { g_ref_tail_quirk = on; }

/* ---------- L0: little-endian byte IO (byteutils.rs:9-48) ---------- */

static inline uint64_t rd_le(const uint8_t *p, size_t nbytes) {
    uint64_t v = 0;
    for (size_t i = 0; i < nbytes; i++) v |= (uint64_t)p[i] << (8 * i);
    return v;
}
static inline void wr_le(uint8_t *p, uint64_t v, size_t nbytes) {
    for (size_t i = 0; i < nbytes; i++) p[i] = (uint8_t)(v >> (8 * i));
}
static inline int elem_width(int etype) { return etype == OCV_U64 ? 8 : 4; }
static inline int popcount8(unsigned m) { return __builtin_popcount(m & 0xffu); }

/* ---------- L1: octet pack (nibblepacking.rs:108-232; nibblepack_simd.rs:22-57) ---------- */

/* Appends the low 4n bits of (x >> 4t) of every nonzero lane, LSB first, as a byte stream.
 * pack_to_even_nibbles (:163-183) and pack_universal (:190-232) both produce exactly this. */
static int64_t pack8_body(const uint64_t in[8], unsigned mask, uint8_t *out, size_t cap, size_t off) {
    uint64_t ored = 0;
    for (int i = 0; i < 8; i++) ored |= in[i];
    unsigned lz = (unsigned)__builtin_clzll(ored), tz = (unsigned)__builtin_ctzll(ored);
    unsigned t = tz / 4;                       /* :140 */
    unsigned n = 16 - lz / 4 - t;              /* :141 */
    if (off >= cap) return -OCV_NOT_ENOUGH_SPACE;
    out[off++] = (uint8_t)(((n - 1) << 4) | t); /* :142 */
    unsigned cnt = (unsigned)popcount8(mask);
    size_t nbytes = ((size_t)n * cnt + 1) / 2;
    if (off + nbytes > cap) return -OCV_NOT_ENOUGH_SPACE;
    unsigned __int128 acc = 0;
    unsigned accbits = 0;
    for (int i = 0; i < 8; i++) {
        if (!in[i]) continue;
        uint64_t f = in[i] >> (4 * t);
        acc |= (unsigned __int128)f << accbits;
        accbits += 4 * n;
        while (accbits >= 8) { out[off++] = (uint8_t)acc; acc >>= 8; accbits -= 8; }
    }
    if (accbits) out[off++] = (uint8_t)acc;
    return (int64_t)off;
}

int64_t ocv_nibble_pack8(const uint64_t in[8], uint8_t *out, size_t cap, size_t off) {
    unsigned mask = 0;
    for (int i = 0; i < 8; i++) if (in[i]) mask |= 1u << i;       /* :114-118 */
    if (off + 1 >= cap) return -OCV_NOT_ENOUGH_SPACE;             /* :120 */
    out[off++] = (uint8_t)mask;
    if (!mask) return (int64_t)off;                               /* :127 */
    return pack8_body(in, mask, out, cap, off);
}

/* pack8_u64_simd (nibblepack_simd.rs:22-57): same bytes, slightly different space check. */
static int64_t pack8_simd(const uint64_t in[8], uint8_t *out, size_t cap, size_t off) {
    if (off + 2 >= cap) return -OCV_NOT_ENOUGH_SPACE;             /* :23 */
    unsigned mask = 0;
    for (int i = 0; i < 8; i++) if (in[i]) mask |= 1u << i;
    out[off++] = (uint8_t)mask;
    if (!mask) return (int64_t)off;
    return pack8_body(in, mask, out, cap, off);
}

int64_t ocv_pack_u64(const uint64_t *vals, size_t n, uint8_t *out, size_t cap, size_t off) {
    uint64_t buf[8];
    size_t i = 0;
    int64_t o = (int64_t)off;
    while (i < n) {                                               /* :75-91 */
        size_t k = n - i < 8 ? n - i : 8;
        for (size_t j = 0; j < 8; j++) buf[j] = j < k ? vals[i + j] : 0;
        o = ocv_nibble_pack8(buf, out, cap, (size_t)o);
        if (o < 0) return o;
        i += k;
    }
    return o;
}

/* ---------- L1: octet unpack ---------- */

/* Field i (width nbits) of a little-endian bitstream starting at p. Caller guarantees bounds. */
static inline uint64_t bits_at(const uint8_t *p, size_t bitpos, unsigned nbits) {
    size_t byte = bitpos >> 3;
    unsigned sh = (unsigned)(bitpos & 7);
    unsigned need = (sh + nbits + 7) / 8;      /* <= 9 */
    unsigned __int128 w = 0;
    for (unsigned i = 0; i < need; i++) w |= (unsigned __int128)p[byte + i] << (8 * i);
    w >>= sh;
    uint64_t v = (uint64_t)w;
    if (nbits < 64) v &= ((uint64_t)1 << nbits) - 1;
    return v;
}

int ocv_unpack8_u32(const uint8_t *in, size_t len, uint32_t out[8], int *zeroes, size_t *consumed) {
    if (zeroes) *zeroes = 0;
    if (len == 0) return OCV_NOT_ENOUGH_SPACE;                    /* nibblepack_simd.rs:515 */
    unsigned mask = in[0];
    if (mask == 0) {                                              /* :518-520 */
        for (int i = 0; i < 8; i++) out[i] = 0;
        if (zeroes) *zeroes = 1;
        *consumed = 1;
        return OCV_OK;
    }
    if (len < 3) return OCV_NOT_ENOUGH_SPACE;                     /* :523 */
    unsigned n = (in[1] >> 4) + 1u;                               /* :524 */
    unsigned tz = (in[1] & 0x0fu) * 4u;                           /* :525 */
    if (n > 8) return OCV_INVALID_FORMAT;                         /* :556 */
    unsigned cnt = (unsigned)popcount8(mask);
    size_t nbytes = ((size_t)n * cnt + 1) / 2;                    /* :534,:543,:547 */
    if (2 + nbytes > len) return OCV_NOT_ENOUGH_SPACE;            /* reference: panic at :565 */
    int drop = g_ref_tail_quirk && n >= 5 && len + t_slice_tail < 34;   /* :546-554 + :433-450 */
    unsigned idx = 0;
    for (int j = 0; j < 8; j++) {
        if (mask & (1u << j)) {
            uint64_t f = drop ? 0 : bits_at(in + 2, (size_t)idx * 4 * n, 4 * n);
            /* u32 lane shift (:581): bits shifted past bit 31 are lost; a count >= 32 gives 0 */
            out[j] = tz >= 32 ? 0u : (uint32_t)((uint32_t)f << tz);
            idx++;
        } else {
            out[j] = 0;
        }
    }
    *consumed = 2 + nbytes;
    return OCV_OK;
}

int ocv_nibble_unpack8(const uint8_t *in, size_t len, uint64_t out[8], size_t *consumed) {
    if (len == 0) return OCV_NOT_ENOUGH_SPACE;                    /* nibblepacking.rs:418 */
    unsigned mask = in[0];
    if (mask == 0) {                                              /* :420-423 */
        for (int i = 0; i < 8; i++) out[i] = 0;
        *consumed = 1;
        return OCV_OK;
    }
    if (len < 2) return OCV_NOT_ENOUGH_SPACE;                     /* :425 */
    unsigned nbits = ((in[1] >> 4) + 1u) * 4u;                    /* :426 */
    unsigned tz = (in[1] & 0x0fu) * 4u;                           /* :427 */
    if (nbits + tz <= 32) {                                       /* :431-434 → u32 path, zero-extended */
        uint32_t o32[8];
        int rc = ocv_unpack8_u32(in, len, o32, NULL, consumed);
        if (rc) return rc;
        for (int i = 0; i < 8; i++) out[i] = o32[i];
        return OCV_OK;
    }
    unsigned cnt = (unsigned)popcount8(mask);
    size_t total = 2 + ((size_t)nbits * cnt + 7) / 8;             /* :436 */
    if (total > len) return OCV_NOT_ENOUGH_SPACE;                 /* reference: error or panic (:457,:474) */
    unsigned idx = 0;
    for (int j = 0; j < 8; j++) {                                 /* :445-470 */
        if (mask & (1u << j)) {
            uint64_t f = bits_at(in + 2, (size_t)idx * nbits, nbits);
            out[j] = f << tz;                                     /* :465 */
            idx++;
        } else {
            out[j] = 0;
        }
    }
    *consumed = total;
    return OCV_OK;
}

int ocv_unpack_u64(const uint8_t *in, size_t len, uint64_t *out, size_t nvalues, size_t *consumed) {
    size_t pos = 0, produced = 0;
    while (produced < nvalues) {                                  /* :370-375 */
        size_t c = 0;
        int rc = ocv_nibble_unpack8(in + pos, len - pos, out + produced, &c);
        if (rc) return rc;
        pos += c;
        produced += 8;
    }
    if (consumed) *consumed = pos;
    return OCV_OK;
}

int64_t ocv_pack_u64_delta(const uint64_t *vals, size_t n, uint8_t *out, size_t cap) {
    uint64_t *d = (uint64_t *)malloc((n ? n : 1) * sizeof(uint64_t));
    uint64_t last = 0;
    for (size_t i = 0; i < n; i++) {                              /* :16-21: saturating_sub */
        d[i] = vals[i] >= last ? vals[i] - last : 0;
        last = vals[i];
    }
    int64_t r = ocv_pack_u64(d, n, out, cap, 0);
    free(d);
    return r;
}

int ocv_unpack_u64_delta(const uint8_t *in, size_t len, uint64_t *out, size_t nvalues, size_t *consumed) {
    int rc = ocv_unpack_u64(in, len, out, nvalues, consumed);
    if (rc) return rc;
    size_t padded = (nvalues + 7) / 8 * 8;
    uint64_t acc = 0;                                             /* DeltaSink :260-269 */
    for (size_t i = 0; i < padded; i++) { acc += out[i]; out[i] = acc; }
    return OCV_OK;
}

int64_t ocv_pack_f64_xor(const double *vals, size_t n, uint8_t *out, size_t cap) {
    if (n == 0) return -OCV_INPUT_TOO_SHORT;                      /* :37 */
    if (cap < 8) return -OCV_NOT_ENOUGH_SPACE;
    uint64_t last;
    memcpy(&last, &vals[0], 8);
    wr_le(out, last, 8);                                          /* :34 */
    uint64_t *d = (uint64_t *)malloc(n * sizeof(uint64_t));
    for (size_t i = 1; i < n; i++) {                              /* :39-44 */
        uint64_t b;
        memcpy(&b, &vals[i], 8);
        d[i - 1] = last ^ b;
        last = b;
    }
    int64_t r = ocv_pack_u64(d, n - 1, out, cap, 8);
    free(d);
    return r;
}

int ocv_unpack_f64_xor(const uint8_t *in, size_t len, double *out, size_t nvalues, size_t *consumed) {
    if (nvalues < 1) return OCV_INPUT_TOO_SHORT;                  /* :395 assert */
    if (len == 0) return OCV_NOT_ENOUGH_SPACE;                    /* direct_read_uint_le :396 */
    uint64_t last = rd_le(in, len < 8 ? len : 8);
    memcpy(&out[0], &last, 8);                                    /* DoubleXorSink::reset :295-299 */
    if (len < 8) return OCV_NOT_ENOUGH_SPACE;                     /* &encoded[8..] */
    size_t rest = nvalues - 1, padded = (rest + 7) / 8 * 8;
    uint64_t *tmp = (uint64_t *)malloc((padded ? padded : 1) * sizeof(uint64_t));
    size_t c = 0;
    int rc = ocv_unpack_u64(in + 8, len - 8, tmp, rest, &c);
    if (!rc) {
        for (size_t i = 0; i < padded; i++) {                     /* :304-316 */
            last ^= tmp[i];
            memcpy(&out[1 + i], &last, 8);
        }
        if (consumed) *consumed = 8 + c;
    }
    free(tmp);
    return rc;
}

/* ---------- histograms (histogram.rs) ---------- */

/* compress_geom_nonincreasing (histogram.rs:55-76): BinHistogramHeader {format_code u8, bucket_def_len
 * u16 = 18, num_buckets u16} (packed, :16-22), PackedGeometricBuckets {initial_bucket f64, multiplier
 * f64} (:40-45), then pack_u64 of the values at offset bucket_def_len + 3 = 21 (:75). */
int64_t ocv_compress_geom_nonincreasing(uint16_t num_buckets, double initial_bucket, double multiplier,
                                        uint8_t format_code, const uint64_t *values, size_t n_values, uint8_t *out,
                                        size_t cap) {
    if (cap < 21) return -OCV_NOT_ENOUGH_SPACE;
    const uint16_t bucket_def_len = 16 + 2;                       /* :62 */
    out[0] = format_code;                                         /* :64 */
    wr_le(out + 1, bucket_def_len, 2);                            /* :65 */
    wr_le(out + 3, num_buckets, 2);                               /* :66 */
    uint64_t b;
    memcpy(&b, &initial_bucket, 8);
    wr_le(out + 5, b, 8);                                         /* :71 */
    memcpy(&b, &multiplier, 8);
    wr_le(out + 13, b, 8);                                        /* :72 */
    return ocv_pack_u64(values, n_values, out, cap, (size_t)bucket_def_len + 3);
}

/* DeltaDiffPackSink (histogram.rs:92-171) driven the way examples/2d_delta_bench.rs:53-57 and the
 * crate's own test (:213-231) drive it: unpack(in, sink, num_buckets) then sink.finish().
 * `last` is last_hist_deltas (num_buckets values, the caller zeroes it where the reference calls
 * reset()).  The packed differences are appended to out at *out_off.  *value_dropped is the flag as
 * it stands after the unpack, before finish() clears it (:124).  finish() packs the last octet a
 * second time whenever at least one octet was processed (:117-122: `i` is already past it, so the
 * zero-fill loop is empty and pack_array still holds the last octet); that duplicate is reproduced. */
int ocv_delta_diff_pack(const uint8_t *in, size_t len, size_t num_buckets, uint64_t *last, uint8_t *out, size_t cap,
                        size_t *out_off, int *value_dropped, size_t *consumed) {
    size_t pos = 0, i = 0;
    int dropped = 0;
    uint64_t pack_array[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int64_t off = (int64_t)*out_off;
    for (long left = (long)num_buckets; left > 0; left -= 8) {    /* unpack, nibblepacking.rs:370-375 */
        uint64_t data[8];
        size_t c = 0;
        int rc = ocv_nibble_unpack8(in + pos, len - pos, data, &c);
        if (rc) return rc;
        pos += c;
        const size_t looplen = i + 8 <= num_buckets ? 8 : num_buckets - i;     /* :132 */
        for (size_t n = 0; n < looplen; n++) {                    /* :133-144 */
            const uint64_t lastv = last[i + n];
            if (data[n] < lastv) { dropped = 1; pack_array[n] = data[n]; }
            else pack_array[n] = data[n] - lastv;
        }
        for (size_t n = 0; n < looplen; n++) last[i + n] = data[n];           /* :146-148 */
        for (size_t n = looplen; n < 8; n++) pack_array[n] = 0;               /* :150-152 */
        off = ocv_nibble_pack8(pack_array, out, cap, (size_t)off);            /* :153 */
        if (off < 0) return (int)-off;
        i += 8;
    }
    if (value_dropped) *value_dropped = dropped;
    if (i != 0) {                                                 /* finish, :115-125 */
        for (size_t j = i; j < 8; j++) pack_array[j] = 0;
        off = ocv_nibble_pack8(pack_array, out, cap, (size_t)off);
        if (off < 0) return (int)-off;
    }
    *out_off = (size_t)off;
    if (consumed) *consumed = pos;
    return OCV_OK;
}

/* ---------- L2: sinks (sink.rs:135-320, filter.rs:53-107) ---------- */

typedef struct sink sink;
struct sink {
    void (*process)(sink *, const void *octet);   /* 8 × T */
    void (*process_zeroes)(sink *);
    void (*reset)(sink *);
};

/* VecSink (sink.rs:151-182) over a caller buffer. */
typedef struct { sink s; int w; uint8_t *out; size_t cap, len; int overflow; } vec_sink;
static void vec_process(sink *s_, const void *oct) {
    vec_sink *s = (vec_sink *)s_;
    if (s->len + 8 > s->cap) { s->overflow = 1; s->len += 8; return; }
    memcpy(s->out + s->len * s->w, oct, 8 * (size_t)s->w);
    s->len += 8;
}
static void vec_zeroes(sink *s_) {
    vec_sink *s = (vec_sink *)s_;
    if (s->len + 8 > s->cap) { s->overflow = 1; s->len += 8; return; }
    memset(s->out + s->len * s->w, 0, 8 * (size_t)s->w);
    s->len += 8;
}
static void vec_reset(sink *s_) { ((vec_sink *)s_)->len = 0; }
static void vec_sink_init(vec_sink *s, int w, void *out, size_t cap) {
    s->s.process = vec_process; s->s.process_zeroes = vec_zeroes; s->s.reset = vec_reset;
    s->w = w; s->out = (uint8_t *)out; s->cap = cap; s->len = 0; s->overflow = 0;
}

/* Section256Sink (sink.rs:195-232). */
typedef struct { sink s; int w; uint8_t values[FIXED_LEN * 8]; size_t i; } s256_sink;
static void s256_process(sink *s_, const void *oct) {
    s256_sink *s = (s256_sink *)s_;
    if (s->i < FIXED_LEN) { memcpy(s->values + s->i * s->w, oct, 8 * (size_t)s->w); s->i += 8; }
}
static void s256_zeroes(sink *s_) {
    s256_sink *s = (s256_sink *)s_;
    if (s->i < FIXED_LEN) { memset(s->values + s->i * s->w, 0, 8 * (size_t)s->w); s->i += 8; }
}
static void s256_reset(sink *s_) { ((s256_sink *)s_)->i = 0; }
static void s256_init(s256_sink *s, int w) {
    s->s.process = s256_process; s->s.process_zeroes = s256_zeroes; s->s.reset = s256_reset;
    s->w = w; s->i = 0; memset(s->values, 0, sizeof(s->values));
}

/* AddConstSink (sink.rs:289-320): u32/u64 wrapping add, f32 IEEE add. */
typedef struct { sink s; int etype; uint8_t base[8]; sink *inner; } add_sink;
static void add_process(sink *s_, const void *oct) {
    add_sink *s = (add_sink *)s_;
    if (s->etype == OCV_U64) {
        uint64_t o[8], b; memcpy(&b, s->base, 8); memcpy(o, oct, 64);
        for (int i = 0; i < 8; i++) o[i] += b;
        s->inner->process(s->inner, o);
    } else if (s->etype == OCV_U32) {
        uint32_t o[8], b; memcpy(&b, s->base, 4); memcpy(o, oct, 32);
        for (int i = 0; i < 8; i++) o[i] += b;
        s->inner->process(s->inner, o);
    } else {
        float o[8], b; memcpy(&b, s->base, 4); memcpy(o, oct, 32);
        for (int i = 0; i < 8; i++) o[i] = o[i] + b;
        s->inner->process(s->inner, o);
    }
}
static void add_zeroes(sink *s_) {                                /* :314-317: forwards the base itself */
    add_sink *s = (add_sink *)s_;
    uint8_t o[64];
    int w = elem_width(s->etype);
    for (int i = 0; i < 8; i++) memcpy(o + i * w, s->base, (size_t)w);
    s->inner->process(s->inner, o);
}
static void noop_reset(sink *s_) { (void)s_; }
static void add_init(add_sink *s, int etype, const void *base, sink *inner) {
    s->s.process = add_process; s->s.process_zeroes = add_zeroes; s->s.reset = noop_reset;
    s->etype = etype; memset(s->base, 0, 8); memcpy(s->base, base, (size_t)elem_width(etype)); s->inner = inner;
}

/* XorSink<f32,u32> (sink.rs:241-284): state is fresh per section (section.rs:707). */
typedef struct { sink s; uint32_t last[8]; sink *inner; } xor_sink;
static void xor_process(sink *s_, const void *oct) {
    xor_sink *s = (xor_sink *)s_;
    uint32_t x[8]; memcpy(x, oct, 32);
    for (int i = 0; i < 8; i++) s->last[i] ^= x[i];
    s->inner->process(s->inner, s->last);                         /* bits reinterpreted as f32 */
}
static void xor_zeroes(sink *s_) {                                /* :278-281: never forwards process_zeroes */
    xor_sink *s = (xor_sink *)s_;
    s->inner->process(s->inner, s->last);
}
static void xor_init(xor_sink *s, sink *inner) {
    s->s.process = xor_process; s->s.process_zeroes = xor_zeroes; s->s.reset = noop_reset;
    memset(s->last, 0, sizeof(s->last)); s->inner = inner;
}

/* GenericFilterSink with EqualsIF / OneOfIF (filter.rs:53-177). */
#define MAX_PRED 64
typedef struct {
    sink s; int etype; size_t npred; uint64_t pred[MAX_PRED]; /* raw bits */
    uint8_t mask[32]; size_t i; int match_zero;
} filt_sink;
static inline int pred_is_zero(int etype, uint64_t bits) {        /* num::Zero::is_zero */
    if (etype == OCV_F32) { float f; uint32_t b = (uint32_t)bits; memcpy(&f, &b, 4); return f == 0.0f; }
    if (etype == OCV_U32) return (uint32_t)bits == 0;
    return bits == 0;
}
static void filt_process(sink *s_, const void *oct) {
    filt_sink *s = (filt_sink *)s_;
    unsigned m = 0;
    if (s->etype == OCV_U64) {
        uint64_t o[8]; memcpy(o, oct, 64);
        for (size_t p = 0; p < s->npred; p++)
            for (int i = 0; i < 8; i++) if (o[i] == s->pred[p]) m |= 1u << i;
    } else if (s->etype == OCV_U32) {
        uint32_t o[8]; memcpy(o, oct, 32);
        for (size_t p = 0; p < s->npred; p++)
            for (int i = 0; i < 8; i++) if (o[i] == (uint32_t)s->pred[p]) m |= 1u << i;
    } else {                                                      /* IEEE == : sink.rs:122-124 */
        float o[8]; memcpy(o, oct, 32);
        for (size_t p = 0; p < s->npred; p++) {
            float pv; uint32_t b = (uint32_t)s->pred[p]; memcpy(&pv, &b, 4);
            for (int i = 0; i < 8; i++) if (o[i] == pv) m |= 1u << i;
        }
    }
    if (s->i < 32) s->mask[s->i] = (uint8_t)m;
    s->i++;
}
static void filt_zeroes(sink *s_) {                               /* filter.rs:73-76 */
    filt_sink *s = (filt_sink *)s_;
    if (s->i < 32) s->mask[s->i] = s->match_zero ? 0xff : 0;
    s->i++;
}
static void filt_reset(sink *s_) { ((filt_sink *)s_)->i = 0; }    /* :85-87: mask is NOT cleared */
static int filt_init(filt_sink *s, int etype, const void *pred_vals, size_t npred) {
    if (npred > MAX_PRED) return OCV_INVALID_FORMAT;
    s->s.process = filt_process; s->s.process_zeroes = filt_zeroes; s->s.reset = filt_reset;
    s->etype = etype; s->npred = npred; s->i = 0; s->match_zero = 0;
    memset(s->mask, 0, 32);
    int w = elem_width(etype);
    for (size_t p = 0; p < npred; p++) {
        s->pred[p] = rd_le((const uint8_t *)pred_vals + p * w, (size_t)w);
        if (pred_is_zero(etype, s->pred[p])) s->match_zero = 1;   /* :119-121, :151-154 */
    }
    return OCV_OK;
}

/* ---------- L3: sections ---------- */

/* SectionWriterStats::from_vect (section.rs:473-480): min_by keeps x unless x > y; max_by keeps
 * x only if x > y (so the last of equals); partial_cmp None counts as Equal. */
static void stats_u64(const uint64_t *v, uint64_t *mn, uint64_t *mx) {
    uint64_t a = v[0], b = v[0];
    for (int i = 1; i < FIXED_LEN; i++) { if (a > v[i]) a = v[i]; if (!(b > v[i])) b = v[i]; }
    *mn = a; *mx = b;
}
static void stats_f32(const float *v, float *mn, float *mx) {
    float a = v[0], b = v[0];
    for (int i = 1; i < FIXED_LEN; i++) { if (a > v[i]) a = v[i]; if (!(b > v[i])) b = v[i]; }
    *mn = a; *mx = b;
}
static inline unsigned bitlen64(uint64_t x) { return x ? 64u - (unsigned)__builtin_clzll(x) : 0u; }

static int64_t write_null(uint8_t *out, size_t cap, size_t off) {       /* section.rs:440-443 */
    if (off >= cap) return -OCV_BAD_OFFSET;
    out[off] = OCV_SECT_NULL;
    return (int64_t)off + 1;
}
static int64_t write_const(int w, uint64_t v0bits, uint8_t *out, size_t cap, size_t off) { /* :795-803 */
    if (off >= cap) return -OCV_BAD_OFFSET;
    if (off + 1 + (size_t)w > cap) return -OCV_NOT_ENOUGH_SPACE;
    out[off] = OCV_SECT_CONST;
    wr_le(out + off + 1, v0bits, (size_t)w);
    return (int64_t)(off + 1 + (size_t)w);
}
static int64_t write_np(const uint64_t *v, uint8_t *out, size_t cap, size_t off) {        /* :570-586 */
    if (off >= cap) return -OCV_BAD_OFFSET;
    out[off] = OCV_SECT_NP;
    int64_t e = ocv_pack_u64(v, FIXED_LEN, out, cap, off + 3);
    if (e < 0) return e;
    size_t nb = (size_t)e - off - 3;
    if (nb > 65535) return -OCV_NOT_ENOUGH_SPACE;
    wr_le(out + off + 1, nb, 2);
    return e;
}
static int64_t write_delta(int w, const uint64_t *v, uint64_t mn, uint64_t mx, uint8_t *out, size_t cap,
                           size_t off) {                                                    /* :650-668 */
    if (off >= cap) return -OCV_BAD_OFFSET;
    if (off + DELTA_HDR > cap) return -OCV_NOT_ENOUGH_SPACE;
    uint64_t d[FIXED_LEN];
    for (int i = 0; i < FIXED_LEN; i++) d[i] = v[i] - mn;
    memset(out + off, 0, DELTA_HDR);       /* u32 writes only 4 base bytes; +8..+11 stay 0 */
    out[off] = OCV_SECT_DELTA;
    int64_t e = ocv_pack_u64(d, FIXED_LEN, out, cap, off + DELTA_HDR);
    if (e < 0) return e;
    size_t nb = (size_t)e - off - DELTA_HDR;
    if (nb > 65535) return -OCV_NOT_ENOUGH_SPACE;
    wr_le(out + off + 1, nb, 2);
    wr_le(out + off + 4, mn, (size_t)w);                          /* :662 */
    out[off + 3] = (uint8_t)bitlen64(mx - mn);                    /* :663, :490-492 */
    return e;
}
static int64_t write_xor_f32(const float *v, uint8_t *out, size_t cap, size_t off) {      /* :733-751 */
    if (off >= cap) return -OCV_BAD_OFFSET;
    out[off] = OCV_SECT_XOR;
    uint64_t last[8] = {0}, x[8];
    int64_t o = (int64_t)off + 3;
    for (int k = 0; k < FIXED_LEN / 8; k++) {
        for (int j = 0; j < 8; j++) {
            uint32_t b; memcpy(&b, &v[8 * k + j], 4);             /* sink.rs:127-130: zero-extended bits */
            x[j] = (uint64_t)b ^ last[j];
            last[j] = b;
        }
        o = pack8_simd(x, out, cap, (size_t)o);
        if (o < 0) return o;
    }
    size_t total = (size_t)o - off;
    if (total > 65535) return -OCV_NOT_ENOUGH_SPACE;
    wr_le(out + off + 1, total, 2);                               /* :745-748: includes the 3 header bytes */
    return o;
}

static void load_u64_vals(int etype, const void *vals, uint64_t *v) {
    if (etype == OCV_U64) memcpy(v, vals, FIXED_LEN * 8);
    else { const uint32_t *p = (const uint32_t *)vals; for (int i = 0; i < FIXED_LEN; i++) v[i] = p[i]; }
}

int64_t ocv_section_write(int etype, const void *vals, uint8_t *out, size_t cap, size_t off) {
    if (etype == OCV_F32) {                                       /* section.rs:719-753 */
        const float *f = (const float *)vals;
        float mn, mx;
        stats_f32(f, &mn, &mx);
        if (mn == mx) {
            if (mn == 0.0f) return write_null(out, cap, off);
            uint32_t b; memcpy(&b, &f[0], 4);
            return write_const(4, b, out, cap, off);
        }
        return write_xor_f32(f, out, cap, off);
    }
    if (etype != OCV_U32 && etype != OCV_U64) return -OCV_INVALID_FORMAT;
    uint64_t v[FIXED_LEN], mn, mx;                                /* AutoEncoder :823-845 */
    load_u64_vals(etype, vals, v);
    stats_u64(v, &mn, &mx);
    int w = elem_width(etype);
    if (mn == mx) {
        if (mn == 0) return write_null(out, cap, off);
        return write_const(w, v[0], out, cap, off);
    }
    unsigned reg = (bitlen64(mx) + 3) / 4, rng = (bitlen64(mx - mn) + 3) / 4;  /* :836-837 */
    if (rng < reg) return write_delta(w, v, mn, mx, out, cap, off);
    return write_np(v, out, cap, off);
}

int64_t ocv_section_write_kind(int etype, int kind, const void *vals, uint8_t *out, size_t cap, size_t off) {
    int w = elem_width(etype);
    if (kind == OCV_SECT_NULL) return write_null(out, cap, off);
    if (kind == OCV_SECT_CONST) return write_const(w, rd_le((const uint8_t *)vals, (size_t)w), out, cap, off);
    if (kind == OCV_SECT_XOR) {
        if (etype != OCV_F32) return -OCV_INVALID_FORMAT;
        return write_xor_f32((const float *)vals, out, cap, off);
    }
    if (etype == OCV_F32) return -OCV_INVALID_FORMAT;
    uint64_t v[FIXED_LEN], mn, mx;
    load_u64_vals(etype, vals, v);
    stats_u64(v, &mn, &mx);
    if (kind == OCV_SECT_NP) return write_np(v, out, cap, off);
    if (kind == OCV_SECT_DELTA) return write_delta(w, v, mn, mx, out, cap, off);
    return -OCV_INVALID_SECTION_TYPE;
}

/* scroll pread_with::<u16>(1, LE) error mapping (error.rs:13-21). */
static int read_u16_at1(const uint8_t *in, size_t len, unsigned *n) {
    if (len <= 1) return OCV_BAD_OFFSET;
    if (len < 3) return OCV_NOT_ENOUGH_SPACE;
    *n = (unsigned)rd_le(in + 1, 2);
    return OCV_OK;
}

int ocv_section_parse(int etype, const uint8_t *in, size_t len, int *kind, size_t *num_bytes) {
    if (len == 0) return OCV_INPUT_TOO_SHORT;                     /* section.rs:257 */
    unsigned n = 0;
    int rc;
    int w = elem_width(etype);
    *kind = in[0];
    switch (in[0]) {
    case OCV_SECT_NULL: *num_bytes = 1; return OCV_OK;            /* :447 */
    case OCV_SECT_NP:                                             /* :534-541, :559 */
        if ((rc = read_u16_at1(in, len, &n))) return rc;
        if ((size_t)n + 3 > len) return OCV_SCROLL_ERR;
        *num_bytes = (size_t)n + 3; return OCV_OK;
    case OCV_SECT_DELTA:                                          /* :613-622, :673 */
        if ((rc = read_u16_at1(in, len, &n))) return rc;
        if ((size_t)n + DELTA_HDR > len) return OCV_SCROLL_ERR;
        *num_bytes = (size_t)n + DELTA_HDR; return OCV_OK;
    case OCV_SECT_CONST:                                          /* :770-777, :807 */
        if (len < 1 + (size_t)w) return OCV_INPUT_TOO_SHORT;
        *num_bytes = 1 + (size_t)w; return OCV_OK;
    case OCV_SECT_XOR:                                            /* :691-698, :757 */
        if ((rc = read_u16_at1(in, len, &n))) return rc;
        if ((size_t)n > len) return OCV_SCROLL_ERR;
        if (n < 3) return OCV_INVALID_FORMAT;
        *num_bytes = n; return OCV_OK;
    default: return OCV_INVALID_SECTION_TYPE;                     /* :258, :40-44 */
    }
}

int ocv_section_delta_info(int etype, const uint8_t *in, size_t len, uint64_t *base, int *numbits) {
    int kind; size_t nb;
    int rc = ocv_section_parse(etype, in, len, &kind, &nb);
    if (rc) return rc;
    if (kind != OCV_SECT_DELTA) return OCV_INVALID_FORMAT;
    *base = rd_le(in + 4, (size_t)elem_width(etype));
    *numbits = in[3];
    return OCV_OK;
}

/* 32 octets of an int section into `out` (FSUtils::nibblepack_decode, section.rs:339-341, :370-372). */
static int decode_octets_int(int etype, const uint8_t *p, size_t len, sink *out) {
    size_t pos = 0;
    for (int k = 0; k < FIXED_LEN / 8; k++) {                     /* :548-553 */
        size_t c = 0;
        if (etype == OCV_U32) {
            uint32_t o[8]; int z = 0;
            int rc = ocv_unpack8_u32(p + pos, len - pos, o, &z, &c);
            if (rc) return rc;
            if (z) out->process_zeroes(out); else out->process(out, o);
        } else {
            uint64_t o[8];
            int rc = ocv_nibble_unpack8(p + pos, len - pos, o, &c);
            if (rc) return rc;
            out->process(out, o);
        }
        pos += c;
    }
    return OCV_OK;
}

/* FixedSectEnum::decode → FSUtils::decode_to_sink (section.rs:237-240, :317-326, :348-357, :379-387).
 * `in` points at the section type byte, nb = its num_bytes() (already parsed). */
static int section_decode_body(int etype, int kind, const uint8_t *in, size_t nb, sink *out);
/* slice_tail: bytes of the vector slice behind this section (only the tail-quirk mode looks at it). */
static int section_decode_to(int etype, int kind, const uint8_t *in, size_t nb, size_t slice_tail, sink *out) {
    t_slice_tail = slice_tail;
    int rc = section_decode_body(etype, kind, in, nb, out);
    t_slice_tail = 0;
    return rc;
}
static int section_decode_body(int etype, int kind, const uint8_t *in, size_t nb, sink *out) {
    int w = elem_width(etype);
    if (kind == OCV_SECT_NULL) {                                  /* :454-460 */
        for (int k = 0; k < FIXED_LEN / 8; k++) out->process_zeroes(out);
        return OCV_OK;
    }
    if (kind == OCV_SECT_CONST) {                                 /* :784-791 */
        uint8_t oct[64];
        for (int i = 0; i < 8; i++) memcpy(oct + i * w, in + 1, (size_t)w);
        for (int k = 0; k < FIXED_LEN / 8; k++) out->process(out, oct);
        return OCV_OK;
    }
    if (etype == OCV_F32) {
        if (kind != OCV_SECT_XOR) return OCV_INVALID_FORMAT;      /* :385 */
        xor_sink xs;                                              /* :703-713 */
        xor_init(&xs, out);
        size_t pos = 3;
        for (int k = 0; k < FIXED_LEN / 8; k++) {
            uint32_t o[8]; int z = 0; size_t c = 0;
            int rc = ocv_unpack8_u32(in + pos, nb - pos, o, &z, &c);
            if (rc) return rc;
            if (z) xs.s.process_zeroes(&xs.s); else xs.s.process(&xs.s, o);
            pos += c;
        }
        return OCV_OK;
    }
    if (kind == OCV_SECT_NP) return decode_octets_int(etype, in + 3, nb - 3, out);
    if (kind == OCV_SECT_DELTA) {                                 /* :633-643 */
        add_sink as;
        add_init(&as, etype, in + 4, out);
        return decode_octets_int(etype, in + DELTA_HDR, nb - DELTA_HDR, &as.s);
    }
    return OCV_INVALID_FORMAT;                                    /* Xor in an int vector: :324, :355 */
}

int ocv_section_decode(int etype, const uint8_t *in, size_t len, void *out256) {
    int kind; size_t nb;
    int rc = ocv_section_parse(etype, in, len, &kind, &nb);
    if (rc) return rc;
    s256_sink s;
    s256_init(&s, elem_width(etype));
    rc = section_decode_to(etype, kind, in, nb, len - nb, &s.s);
    if (rc) return rc;
    memcpy(out256, s.values, FIXED_LEN * (size_t)elem_width(etype));
    return OCV_OK;
}

/* ---------- L4: appender (vector.rs:207-379) ---------- */

struct ocv_appender {
    int etype, w;
    uint8_t *buf; size_t cap;          /* vect_buf (len == capacity, zero-filled) */
    size_t offset;
    uint8_t wbuf[FIXED_LEN * 8]; size_t wlen;   /* write_buf */
    uint32_t num_elements; uint16_t num_null_sections;
};

static void app_write_header(ocv_appender *a) {                   /* vector.rs:113-131, :265 */
    wr_le(a->buf, HEADER_BYTES - 4, 4);
    a->buf[4] = 0x10;                                             /* VectorType::FixedSection256 */
    a->buf[5] = (uint8_t)a->etype;
    a->buf[6] = 0; a->buf[7] = 0;
}
static void app_write_stats(ocv_appender *a) {                    /* :181-185 */
    wr_le(a->buf + 8, a->num_elements, 4);
    wr_le(a->buf + 12, a->num_null_sections, 2);
}
static void app_update_num_bytes(ocv_appender *a) {               /* :136-142 */
    wr_le(a->buf, (uint64_t)(a->offset - HEADER_BYTES) + (HEADER_BYTES - 4), 4);
}
static int app_grow(ocv_appender *a) {                            /* :288-290 */
    size_t ncap = a->cap * 2 > a->cap + GROW_BYTES ? a->cap * 2 : a->cap + GROW_BYTES;
    uint8_t *nb = (uint8_t *)realloc(a->buf, ncap);
    if (!nb) return OCV_NOT_ENOUGH_SPACE;
    memset(nb + a->cap, 0, ncap - a->cap);
    a->buf = nb; a->cap = ncap;
    return OCV_OK;
}

ocv_appender *ocv_appender_new(int etype, size_t initial_capacity) {
    if (etype != OCV_U32 && etype != OCV_U64 && etype != OCV_F32) return NULL;
    if (initial_capacity < HEADER_BYTES) return NULL;
    ocv_appender *a = (ocv_appender *)calloc(1, sizeof(*a));
    a->etype = etype; a->w = elem_width(etype);
    a->buf = (uint8_t *)calloc(initial_capacity, 1); a->cap = initial_capacity;
    a->offset = HEADER_BYTES;
    app_write_header(a);
    return a;
}
void ocv_appender_free(ocv_appender *a) { if (a) { free(a->buf); free(a); } }

static int app_encode_section(ocv_appender *a) {                  /* :270-279 */
    int64_t e = ocv_section_write(a->etype, a->wbuf, a->buf, a->cap, a->offset);
    if (e == -OCV_NOT_ENOUGH_SPACE || e == -OCV_BAD_OFFSET) {     /* retry_grow :283-296: once */
        int rc = app_grow(a);
        if (rc) return rc;
        e = ocv_section_write(a->etype, a->wbuf, a->buf, a->cap, a->offset);
    }
    if (e < 0) return (int)-e;
    a->offset = (size_t)e;
    a->wlen = 0;
    a->num_elements += FIXED_LEN;
    app_write_stats(a);
    app_update_num_bytes(a);
    return OCV_OK;
}

int ocv_appender_append(ocv_appender *a, const void *value) {     /* :300-307 */
    memcpy(a->wbuf + a->wlen * a->w, value, (size_t)a->w);
    a->wlen++;
    if (a->wlen >= FIXED_LEN) return app_encode_section(a);
    return OCV_OK;
}
int ocv_appender_append_many(ocv_appender *a, const void *values, size_t n) {
    for (size_t i = 0; i < n; i++) {
        int rc = ocv_appender_append(a, (const uint8_t *)values + i * a->w);
        if (rc) return rc;
    }
    return OCV_OK;
}

int ocv_appender_append_nulls(ocv_appender *a, size_t num_nulls) { /* :311-335 */
    size_t left = num_nulls;
    while (left > 0) {
        if (a->wlen > 0) {
            size_t fill = left < FIXED_LEN - a->wlen ? left : FIXED_LEN - a->wlen;
            memset(a->wbuf + a->wlen * a->w, 0, fill * a->w);
            a->wlen += fill;
            left -= fill;
            if (a->wlen >= FIXED_LEN) { int rc = app_encode_section(a); if (rc) return rc; }
        } else if (left >= FIXED_LEN) {
            int64_t e = write_null(a->buf, a->cap, a->offset);
            if (e < 0) { int rc = app_grow(a); if (rc) return rc; e = write_null(a->buf, a->cap, a->offset); }
            if (e < 0) return (int)-e;
            a->offset = (size_t)e;
            a->num_null_sections += 1;                            /* :323 */
            a->num_elements += FIXED_LEN;
            app_write_stats(a);
            app_update_num_bytes(a);
            left -= FIXED_LEN;
        } else {
            memset(a->wbuf, 0, left * a->w);
            a->wlen = left;
            left = 0;
        }
    }
    return OCV_OK;
}

size_t ocv_appender_num_elements(const ocv_appender *a) { return (size_t)a->num_elements + a->wlen; }
const uint8_t *ocv_appender_bytes(const ocv_appender *a, size_t *len) { *len = a->offset; return a->buf; }

static void app_reset(ocv_appender *a) {                          /* :254-262 */
    memset(a->buf, 0, a->cap);
    a->offset = HEADER_BYTES; a->wlen = 0;
    a->num_elements = 0; a->num_null_sections = 0;
    app_write_stats(a);
    app_write_header(a);
}

int64_t ocv_appender_finish(ocv_appender *a, size_t total_rows, uint8_t *out, size_t cap) { /* :344-370 */
    size_t so_far = (size_t)a->num_elements + a->wlen;
    if (so_far > total_rows) return -OCV_INVALID_NUM_ROWS;
    if (total_rows > 0xffffffffull) return -OCV_INVALID_NUM_ROWS;
    if (a->wlen > 0) { int rc = ocv_appender_append_nulls(a, FIXED_LEN - a->wlen); if (rc) return -rc; }
    while ((uint64_t)a->num_elements < total_rows) {
        int rc = ocv_appender_append_nulls(a, FIXED_LEN);
        if (rc) return -rc;
    }
    a->num_elements = (uint32_t)total_rows;
    app_write_stats(a);
    size_t n = a->offset;
    if (n > cap) { app_reset(a); return -OCV_NOT_ENOUGH_SPACE; }
    memcpy(out, a->buf, n);
    app_reset(a);
    return (int64_t)n;
}

size_t ocv_max_encoded_size(int etype, size_t n) {
    size_t sections = (n + FIXED_LEN - 1) / FIXED_LEN;
    size_t per = etype == OCV_U64 ? DELTA_HDR + 32 * 66 : DELTA_HDR + 32 * 34;
    return HEADER_BYTES + sections * per + 16;
}

int64_t ocv_encode_all(int etype, const void *values, size_t n, uint8_t *out, size_t cap) {  /* :238-246 */
    ocv_appender *a = ocv_appender_new(etype, 4096);
    if (!a) return -OCV_INVALID_FORMAT;
    int rc = ocv_appender_append_many(a, values, n);
    int64_t r = rc ? -rc : ocv_appender_finish(a, n, out, cap);
    ocv_appender_free(a);
    return r;
}

/* ---------- L4: reader (vector.rs:395-465) ---------- */

int ocv_reader_check(int etype, const uint8_t *bytes, size_t len, uint64_t *detail) {  /* :404-414 */
    if (detail) *detail = 0;
    if (len == 0) return OCV_BAD_OFFSET;                          /* scroll pread u32 @0 */
    if (len < 4) return OCV_NOT_ENOUGH_SPACE;
    uint64_t from_header = rd_le(bytes, 4);
    if (len <= 5) { if (detail) *detail = 5; return OCV_BAD_OFFSET; }
    unsigned subtype = bytes[5];
    if ((uint64_t)len < from_header + 4) return OCV_INPUT_TOO_SHORT;
    if (subtype != (unsigned)etype) { if (detail) *detail = subtype; return OCV_WRONG_VECTOR_TYPE; }
    if (len < HEADER_BYTES) return OCV_INPUT_TOO_SHORT;           /* reference: slice panic at :443 */
    return OCV_OK;
}

int ocv_reader_num_elements(const uint8_t *bytes, size_t len, uint32_t *n, uint16_t *null_sections) {
    if (len < HEADER_BYTES) return OCV_INPUT_TOO_SHORT;
    *n = (uint32_t)rd_le(bytes + 8, 4);                           /* :416-419, :437-439 */
    if (null_sections) *null_sections = (uint16_t)rd_le(bytes + 12, 2);
    return OCV_OK;
}

int ocv_reader_sections(int etype, const uint8_t *bytes, size_t len, size_t *count, uint8_t *kinds,
                        size_t max_kinds) {
    int rc = ocv_reader_check(etype, bytes, len, NULL);
    if (rc) return rc;
    size_t pos = HEADER_BYTES, c = 0;
    while (pos < len) {                                           /* section.rs:866-876 */
        int kind; size_t nb;
        rc = ocv_section_parse(etype, bytes + pos, len - pos, &kind, &nb);
        if (rc) { *count = c; return rc; }
        if (kinds && c < max_kinds) kinds[c] = (uint8_t)kind;
        c++; pos += nb;
    }
    *count = c;
    return OCV_OK;
}

/* Where and why the iteration over a vector stops, in two words: detail[0] = index of the section at which it
 * stops (all ones: the vector header itself, VectorReader::try_new), detail[1] = the payload the reference's
 * CodingError carries (error.rs:2-11): the offset of BadOffset, the type byte of InvalidSectionType, the subtype
 * byte of WrongVectorType; for every other code the byte offset, inside the vector, of the section where it
 * stops (0 for the header).  Returns the code, OCV_OK (detail 0, 0) for a vector that iterates to its end. */
int ocv_error_detail(int etype, const uint8_t *bytes, size_t len, uint64_t detail[2]) {
    uint64_t d = 0;
    detail[0] = detail[1] = 0;
    int rc = ocv_reader_check(etype, bytes, len, &d);
    if (rc) { detail[0] = ~0ull; detail[1] = d; return rc; }
    s256_sink s;
    s256_init(&s, elem_width(etype));
    size_t pos = HEADER_BYTES, k = 0;
    while (pos < len) {
        int kind; size_t nb;
        rc = ocv_section_parse(etype, bytes + pos, len - pos, &kind, &nb);
        if (rc) {
            detail[0] = k;
            detail[1] = rc == OCV_BAD_OFFSET ? 1 : rc == OCV_INVALID_SECTION_TYPE ? bytes[pos] : pos;
            return rc;
        }
        s.s.reset(&s.s);
        rc = section_decode_to(etype, kind, bytes + pos, nb, len - pos - nb, &s.s);
        if (rc) { detail[0] = k; detail[1] = pos; return rc; }
        pos += nb; k++;
    }
    return OCV_OK;
}

static int decode_vector_to(int etype, const uint8_t *bytes, size_t len, sink *out) {  /* vector.rs:458-464 */
    int rc = ocv_reader_check(etype, bytes, len, NULL);
    if (rc) return rc;
    size_t pos = HEADER_BYTES;
    while (pos < len) {
        int kind; size_t nb;
        rc = ocv_section_parse(etype, bytes + pos, len - pos, &kind, &nb);
        if (rc) return rc;
        rc = section_decode_to(etype, kind, bytes + pos, nb, len - pos - nb, out);
        if (rc) return rc;
        pos += nb;
    }
    return OCV_OK;
}

int ocv_decode_to_vec(int etype, const uint8_t *bytes, size_t len, int use_add, const void *add_base,
                      void *out, size_t out_cap_elems, size_t *n_out) {
    vec_sink vs;
    vec_sink_init(&vs, elem_width(etype), out, out_cap_elems);
    add_sink as;
    sink *top = &vs.s;
    if (use_add) { add_init(&as, etype, add_base, &vs.s); top = &as.s; }
    int rc = decode_vector_to(etype, bytes, len, top);
    if (n_out) *n_out = vs.len;
    if (rc) return rc;
    return vs.overflow ? OCV_NOT_ENOUGH_SPACE : OCV_OK;
}

int ocv_iterate(int etype, const uint8_t *bytes, size_t len, void *out, size_t out_cap_elems, size_t *n_out) {
    int rc = ocv_reader_check(etype, bytes, len, NULL);           /* vector.rs:452, :512-556 */
    if (rc) return rc;
    uint32_t n;
    ocv_reader_num_elements(bytes, len, &n, NULL);
    if (n > out_cap_elems) return OCV_NOT_ENOUGH_SPACE;
    int w = elem_width(etype);
    s256_sink s;
    s256_init(&s, w);
    size_t pos = HEADER_BYTES;
    for (size_t i = 0; i < n; i++) {
        if (i % FIXED_LEN == 0) {                                 /* next_section :533-538 */
            s.s.reset(&s.s);
            if (pos < len) {
                int kind; size_t nb;
                if (ocv_section_parse(etype, bytes + pos, len - pos, &kind, &nb) == OCV_OK) {
                    rc = section_decode_to(etype, kind, bytes + pos, nb, len - pos - nb, &s.s);
                    if (rc) return rc;                            /* reference: expect() panic */
                    pos += nb;
                }
            }
        }
        memcpy((uint8_t *)out + i * w, s.values + (i % FIXED_LEN) * w, (size_t)w);
    }
    if (n_out) *n_out = n;
    return OCV_OK;
}

/* ---------- L4: filter (filter.rs:189-321) ---------- */

static unsigned popcount_mask(const uint8_t m[32]) {
    unsigned c = 0;
    for (int i = 0; i < 32; i++) c += (unsigned)popcount8(m[i]);
    return c;
}

/* One VectorFilter::next step (filter.rs:218-230). Returns 1 = yielded a mask, 0 = end of iteration;
 * *err = parse error code (reference panics) or swallowed decode error (iteration just ends). */
typedef struct { int etype; const uint8_t *bytes; size_t len, pos; filt_sink fs; } vfilter;
static int vfilter_next(vfilter *f, uint8_t mask[32], int *err) {
    *err = OCV_OK;
    if (f->pos >= f->len) return 0;
    int kind; size_t nb;
    int rc = ocv_section_parse(f->etype, f->bytes + f->pos, f->len - f->pos, &kind, &nb);
    if (rc) { *err = rc; return 0; }
    const uint8_t *sect = f->bytes + f->pos;
    f->pos += nb;
    if (kind == OCV_SECT_NULL) {                                  /* :222-223, :104-106 */
        memset(mask, f->fs.match_zero ? 0xff : 0, 32);
        return 1;
    }
    f->fs.s.reset(&f->fs.s);
    rc = section_decode_to(f->etype, kind, sect, nb, f->len - f->pos, &f->fs.s);
    if (rc) { *err = rc; return 0; }                              /* :226 .ok()? */
    memcpy(mask, f->fs.mask, 32);
    return 1;
}
static void vfilter_advance(vfilter *f) {                         /* :207-209 */
    if (f->pos >= f->len) return;
    int kind; size_t nb;
    if (ocv_section_parse(f->etype, f->bytes + f->pos, f->len - f->pos, &kind, &nb) == OCV_OK) f->pos += nb;
}

static int filter_count_ex(int etype, const uint8_t *bytes, size_t len, int pred_kind, const void *pred_vals,
                           size_t npred, uint64_t *count, uint8_t *masks, size_t max_sections,
                           size_t *n_sections, int *stop_err) {
    *count = 0;
    if (n_sections) *n_sections = 0;
    if (stop_err) *stop_err = OCV_OK;
    int rc = ocv_reader_check(etype, bytes, len, NULL);
    if (rc) return rc;
    if (pred_kind == OCV_PRED_EQ) npred = 1;
    vfilter f;
    f.etype = etype; f.bytes = bytes; f.len = len; f.pos = HEADER_BYTES;
    rc = filt_init(&f.fs, etype, pred_vals, npred);
    if (rc) return rc;
    uint32_t total = 0;                                           /* filter.rs:298: u32 sum */
    size_t ns = 0;
    uint8_t m[32];
    int err;
    while (vfilter_next(&f, m, &err)) {
        total += popcount_mask(m);
        if (masks && ns < max_sections) memcpy(masks + 32 * ns, m, 32);
        ns++;
    }
    *count = total;
    if (n_sections) *n_sections = ns;
    if (stop_err) *stop_err = err;
    return OCV_OK;
}

int ocv_filter_count(int etype, const uint8_t *bytes, size_t len, int pred_kind, const void *pred_vals,
                     size_t npred, uint64_t *count, uint8_t *masks, size_t max_sections, size_t *n_sections) {
    return filter_count_ex(etype, bytes, len, pred_kind, pred_vals, npred, count, masks, max_sections,
                           n_sections, NULL);
}

int ocv_match_positions(const uint8_t *masks, size_t n_sections, int ref_quirk, uint64_t *pos_out,
                        size_t max_pos, size_t *n_pos) {
    uint64_t pos = 0;
    size_t np = 0;
    for (size_t s = 0; s < n_sections; s++) {
        for (int word = 0; word < 8; word++) {                    /* filter.rs:307-317 */
            uint32_t m = (uint32_t)rd_le(masks + 32 * s + 4 * word, 4);
            if (m != 0) {
                for (int bit = 0; bit < 32; bit++) {
                    if (m & (1u << bit)) { if (np < max_pos) pos_out[np] = pos; np++; }
                    pos++;
                }
            } else if (!ref_quirk) {
                pos += 32;                                        /* the reference forgets this */
            }
        }
    }
    *n_pos = np;
    return np > max_pos ? OCV_NOT_ENOUGH_SPACE : OCV_OK;
}

int ocv_multi_filter_count(int etype, const uint8_t *const *bytes, const size_t *lens, size_t k,
                           const void *pred_vals, uint64_t *count, uint8_t *masks, size_t max_sections,
                           size_t *n_sections) {
    *count = 0;
    if (n_sections) *n_sections = 0;
    if (k == 0 || k > 16) return OCV_INVALID_FORMAT;              /* filter.rs:249 panics on empty */
    vfilter f[16];
    int w = elem_width(etype);
    for (size_t i = 0; i < k; i++) {
        int rc = ocv_reader_check(etype, bytes[i], lens[i], NULL);
        if (rc) return rc;
        f[i].etype = etype; f[i].bytes = bytes[i]; f[i].len = lens[i]; f[i].pos = HEADER_BYTES;
        rc = filt_init(&f[i].fs, etype, (const uint8_t *)pred_vals + i * w, 1);
        if (rc) return rc;
    }
    uint32_t total = 0;
    size_t ns = 0;
    for (;;) {                                                    /* filter.rs:260-285 */
        uint8_t m[32], m2[32];
        int err;
        if (!vfilter_next(&f[0], m, &err)) break;
        size_t n = 1;
        int ended = 0;
        while (n < k) {
            int nonzero = 0;
            for (int i = 0; i < 32; i++) nonzero |= m[i];
            if (!nonzero) break;
            if (!vfilter_next(&f[n], m2, &err)) { ended = 1; break; }
            for (int i = 0; i < 32; i++) m[i] &= m2[i];
            n++;
        }
        if (ended) break;
        while (n < k) { vfilter_advance(&f[n]); n++; }
        total += popcount_mask(m);
        if (masks && ns < max_sections) memcpy(masks + 32 * ns, m, 32);
        ns++;
    }
    *count = total;
    if (n_sections) *n_sections = ns;
    return OCV_OK;
}

/* ---------- batch helpers ---------- */

int ocv_batch_filter_count(int etype, const uint8_t *blob, const uint64_t *byte_off, size_t n,
                           int pred_kind, const void *pred_vals, size_t npred,
                           uint64_t *total, uint64_t *per_vec, int32_t *status) {
    uint64_t t = 0;
    int first_err = OCV_OK;
    for (size_t v = 0; v < n; v++) {
        uint64_t c = 0;
        int stop = OCV_OK;
        int rc = filter_count_ex(etype, blob + byte_off[v], (size_t)(byte_off[v + 1] - byte_off[v]), pred_kind,
                                 pred_vals, npred, &c, NULL, 0, NULL, &stop);
        if (!rc) rc = stop;
        if (rc && !first_err) first_err = rc;
        if (status) status[v] = rc;
        if (per_vec) per_vec[v] = c;
        t += c;
    }
    *total = t;
    return first_err;
}

int ocv_batch_decode(int etype, const uint8_t *blob, const uint64_t *byte_off, size_t n, int use_add,
                     const void *add_base, void *out, const uint64_t *out_off, int32_t *status) {
    int first_err = OCV_OK;
    int w = elem_width(etype);
    for (size_t v = 0; v < n; v++) {
        size_t no = 0;
        int rc = ocv_decode_to_vec(etype, blob + byte_off[v], (size_t)(byte_off[v + 1] - byte_off[v]), use_add,
                                   add_base, (uint8_t *)out + out_off[v] * w,
                                   (size_t)(out_off[v + 1] - out_off[v]), &no);
        if (rc && !first_err) first_err = rc;
        if (status) status[v] = rc;
    }
    return first_err;
}

int64_t ocv_batch_encode(int etype, const void *values, const uint64_t *elem_off, size_t n,
                         uint8_t *out, size_t cap, uint64_t *byte_off) {
    int w = elem_width(etype);
    ocv_appender *a = ocv_appender_new(etype, 8192);
    if (!a) return -OCV_INVALID_FORMAT;
    size_t pos = 0;
    for (size_t v = 0; v < n; v++) {
        byte_off[v] = pos;
        size_t cnt = (size_t)(elem_off[v + 1] - elem_off[v]);
        int rc = ocv_appender_append_many(a, (const uint8_t *)values + elem_off[v] * w, cnt);
        if (rc) { ocv_appender_free(a); return -rc; }
        int64_t r = ocv_appender_finish(a, cnt, out + pos, cap - pos);
        if (r < 0) { ocv_appender_free(a); return r; }
        pos += (size_t)r;
    }
    byte_off[n] = pos;
    ocv_appender_free(a);
    return (int64_t)pos;
}
