/* sanitize_driver.c - drives the CPU oracle under AddressSanitizer + UndefinedBehaviorSanitizer.
 *
 * Test infrastructure, like the oracle itself.  Python loads the sanitizer build of the oracle too late for
 * ASan to own the heap, so heap overruns of the decoders would go unseen there; this program is linked with the
 * sanitizers from the start.  Every input of a decoder lives in a heap block of exactly its size, so one byte read
 * past a (possibly damaged or truncated) vector, section or stream ends the run with a report.
 *
 *   sanitize_driver SEED ROUNDS
 *
 * Per round and element type: random values (dense small, full width, sparse with Null / Const sections, a Delta
 * section, NaN and signed zeros for f32) -> encode_all -> reader checks, section walk, decode (with and without
 * AddConst), iterate, filter (Equals and OneOf) with masks, match positions, the appender with explicit nulls, a
 * two-column MultiVectorFilter; then the same read calls on damaged copies (flipped bytes, every kind of
 * truncation) where only "no sanitizer report, no crash" is asked; the legacy stream codecs and the histogram
 * sink on round trips and on damaged streams.  Exit 0 = every self-check held and the sanitizers saw nothing. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "cvec_oracle.h"

static int g_quirk; /* the reference's tail decoder is on: the last wide octets of a slice decode to zeros, so round trips are not the identity */
static uint64_t rng_state;
static uint64_t rnd(void) { /* SplitMix64 */
    uint64_t z = (rng_state += 0x9e3779b97f4a7c15ull);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}
static uint64_t below(uint64_t n) { return n ? rnd() % n : 0; }

#define CHECK(cond)                                                                       \
    do {                                                                                  \
        if (!(cond)) {                                                                    \
            fprintf(stderr, "self-check failed at %s:%d: %s\n", __FILE__, __LINE__, #cond); \
            exit(1);                                                                      \
        }                                                                                 \
    } while (0)

static void *exact(const void *src, size_t n) { /* a heap block of exactly n bytes */
    uint8_t *p = malloc(n ? n : 1);
    CHECK(p);
    if (n) memcpy(p, src, n);
    return p;
}

static size_t width(int etype) { return etype == OCV_U64 ? 8 : 4; }

static void fill_values(int etype, void *vals, size_t n, int shape) {
    for (size_t i = 0; i < n; i++) {
        uint64_t x;
        const size_t sec = i / 256;
        switch (shape) {
        case 0: x = below(16); break;                                 /* dense, low cardinality */
        case 1: x = rnd() >> below(64); break;                        /* every width */
        case 2: x = (sec % 3 == 0) ? 0 : (sec % 3 == 1 ? 77 : below(1000)); break; /* Null / Const / NP sections */
        case 3: x = 100000 + below(64); break;                        /* Delta */
        default: x = (i % 50 == 0) ? below(1u << 20) : 0; break;      /* sparse */
        }
        if (etype == OCV_U64) ((uint64_t *)vals)[i] = x;
        else if (etype == OCV_U32) ((uint32_t *)vals)[i] = (uint32_t)x;
        else {
            float f = (float)(x & 0xffffff) / 1.4f;
            if (shape == 1 && i % 97 == 0) f = NAN;
            if (shape == 1 && i % 89 == 0) f = -0.0f;
            if (shape == 1 && i % 83 == 0) { uint32_t b = (uint32_t)x; memcpy(&f, &b, 4); }
            ((float *)vals)[i] = f;
        }
    }
}

/* every read entry point on one vector; `expect_ok`: the vector is well formed */
static void read_all(int etype, const uint8_t *bytes, size_t len, const void *vals, size_t n, int expect_ok) {
    const size_t w = width(etype), nsec = (n + 255) / 256, cap = nsec * 256 + 512;
    uint64_t detail[2] = {0, 0};
    int rc = ocv_reader_check(etype, bytes, len, detail);
    if (expect_ok) CHECK(rc == 0);
    (void)ocv_error_detail(etype, bytes, len, detail);
    uint32_t ne = 0;
    uint16_t nulls = 0;
    (void)ocv_reader_num_elements(bytes, len, &ne, &nulls);
    size_t count = 0;
    uint8_t kinds[64];
    rc = ocv_reader_sections(etype, bytes, len, &count, kinds, 64);
    if (expect_ok) CHECK(rc == 0 && count == nsec);
    void *out = malloc(cap * w);
    size_t n_out = 0;
    rc = ocv_decode_to_vec(etype, bytes, len, 0, NULL, out, cap, &n_out);
    if (expect_ok) {
        CHECK(rc == 0 && n_out == nsec * 256);
        /* (f32 sections that hold a NaN are lossy by the format's stats fold: no identity check there) */
        if (etype != OCV_F32 && !g_quirk) CHECK(memcmp(out, vals, n * w) == 0);
    }
    uint64_t add = 7;
    float addf = 2.5f;
    (void)ocv_decode_to_vec(etype, bytes, len, 1, etype == OCV_F32 ? (void *)&addf : (void *)&add, out, cap, &n_out);
    (void)ocv_iterate(etype, bytes, len, out, cap, &n_out);
    /* Equals on the first value, OneOf on three */
    uint8_t pv[3 * 8];
    memset(pv, 0, sizeof(pv));
    for (int j = 0; j < 3; j++) memcpy(pv + j * w, (const uint8_t *)vals + (n ? (size_t)below(n) : 0) * w, n ? w : 0);
    uint8_t *masks = malloc((nsec + 2) * 32);
    uint64_t cnt = 0;
    size_t ns = 0;
    rc = ocv_filter_count(etype, bytes, len, OCV_PRED_EQ, pv, 1, &cnt, masks, nsec + 2, &ns);
    if (expect_ok && etype != OCV_F32 && !g_quirk) {
        uint64_t want = 0, p0 = 0, x;
        memcpy(&p0, pv, w);
        for (size_t i = 0; i < n; i++) { x = 0; memcpy(&x, (const uint8_t *)vals + i * w, w); want += x == p0; }
        if (p0 == 0) want += nsec * 256 - n; /* the padding of the last section matches zero */
        CHECK(rc == 0 && ns == nsec && cnt == want);
        uint64_t *pos = malloc((nsec * 256 + 1) * 8);
        size_t n_pos = 0;
        CHECK(ocv_match_positions(masks, ns, 0, pos, nsec * 256, &n_pos) == 0 && n_pos == cnt);
        (void)ocv_match_positions(masks, ns, 1, pos, nsec * 256, &n_pos);
        free(pos);
    }
    (void)ocv_filter_count(etype, bytes, len, OCV_PRED_ONEOF, pv, 3, &cnt, masks, nsec + 2, &ns);
    (void)ocv_filter_count(etype, bytes, len, OCV_PRED_EQ, pv, 1, &cnt, NULL, 0, &ns);
    const uint8_t *two[2] = {bytes, bytes};
    const size_t lens[2] = {len, len};
    uint8_t pv2[2 * 8];
    memcpy(pv2, pv, w);
    memcpy(pv2 + w, pv, w);
    (void)ocv_multi_filter_count(etype, two, lens, 2, pv2, &cnt, masks, nsec + 2, &ns);
    free(masks);
    free(out);
}

static void vector_round(int etype) {
    const size_t w = width(etype);
    const size_t n = 1 + below(1500);
    const int shape = (int)below(5);
    void *vals = malloc(n * w);
    fill_values(etype, vals, n, shape);
    const size_t cap = ocv_max_encoded_size(etype, n);
    uint8_t *big = malloc(cap);
    const int64_t len = ocv_encode_all(etype, vals, n, big, cap);
    CHECK(len > 0 && (size_t)len <= cap);
    uint8_t *enc = exact(big, (size_t)len);
    read_all(etype, enc, (size_t)len, vals, n, 1);
    /* the appender, value by value with explicit nulls in front, against encode_all of the same rows */
    {
        ocv_appender *a = ocv_appender_new(etype, 64);
        CHECK(a);
        const size_t nulls = below(700);
        CHECK(ocv_appender_append_nulls(a, nulls) == 0);
        const size_t half = n / 2;
        CHECK(ocv_appender_append_many(a, vals, half) == 0);
        for (size_t i = half; i < n; i++) CHECK(ocv_appender_append(a, (const uint8_t *)vals + i * w) == 0);
        CHECK(ocv_appender_num_elements(a) == nulls + n);
        size_t rl = 0;
        const uint8_t *rb = ocv_appender_bytes(a, &rl);
        uint8_t *rcopy = exact(rb, rl);
        size_t cnt = 0;
        uint8_t kinds[16];
        (void)ocv_reader_sections(etype, rcopy, rl, &cnt, kinds, 16);
        free(rcopy);
        const size_t rows = nulls + n + below(600);
        const size_t fcap = ocv_max_encoded_size(etype, rows) + 64;
        uint8_t *fin = malloc(fcap);
        const int64_t fl = ocv_appender_finish(a, rows, fin, fcap);
        CHECK(fl > 0);
        uint8_t *fe = exact(fin, (size_t)fl);
        void *all = calloc(rows, w);
        memcpy((uint8_t *)all + nulls * w, vals, n * w);
        read_all(etype, fe, (size_t)fl, all, rows, 1);
        CHECK(ocv_appender_finish(a, 0, fin, fcap) <= 0 || 1); /* an empty appender finishes too */
        free(all); free(fe); free(fin);
        ocv_appender_free(a);
    }
    /* damaged copies: flipped bytes, then every truncation class */
    for (int d = 0; d < 24; d++) {
        uint8_t *bad = exact(enc, (size_t)len);
        const int flips = 1 + (int)below(4);
        for (int f = 0; f < flips; f++) bad[below((uint64_t)len)] ^= (uint8_t)(1u << below(8));
        if (d % 3 == 0) bad[16 + below((uint64_t)len - 16 ? (uint64_t)len - 16 : 1) % (size_t)len] = (uint8_t)below(8); /* a section type byte, perhaps */
        read_all(etype, bad, (size_t)len, vals, n, 0);
        free(bad);
    }
    for (int d = 0; d < 24; d++) {
        const size_t cut = d < 20 ? below((uint64_t)len) : (size_t)len - 1 - below(3) % (size_t)len;
        uint8_t *bad = exact(enc, cut);
        read_all(etype, bad, cut, vals, n, 0);
        free(bad);
    }
    /* single sections: every writer kind that accepts the values, parsed and decoded from an exact block */
    if (n >= 256) {
        uint8_t sbuf[4096];
        const int kinds[] = {0, 1, 3, 5, 6};
        for (size_t k = 0; k < sizeof(kinds) / sizeof(kinds[0]); k++) {
            const int64_t sl = ocv_section_write_kind(etype, kinds[k], vals, sbuf, sizeof(sbuf), 0);
            if (sl <= 0) continue;
            uint8_t *sec = exact(sbuf, (size_t)sl);
            int kind = -1;
            size_t nb = 0;
            uint8_t out256[256 * 8];
            uint64_t base = 0;
            int bits = 0;
            (void)ocv_section_parse(etype, sec, (size_t)sl, &kind, &nb);
            (void)ocv_section_decode(etype, sec, (size_t)sl, out256);
            (void)ocv_section_delta_info(etype, sec, (size_t)sl, &base, &bits);
            for (size_t cut = 0; cut < (size_t)sl; cut += 1 + (size_t)sl / 40) {
                uint8_t *s2 = exact(sec, cut);
                (void)ocv_section_parse(etype, s2, cut, &kind, &nb);
                (void)ocv_section_decode(etype, s2, cut, out256);
                free(s2);
            }
            free(sec);
        }
        const int64_t sl = ocv_section_write(etype, vals, sbuf, sizeof(sbuf), 0);
        CHECK(sl > 0);
        CHECK(ocv_section_write(etype, vals, sbuf, 0, 0) < 0); /* no room */
    }
    free(enc); free(big); free(vals);
}

static void stream_round(void) {
    const size_t n = 1 + below(300), cap = (n / 8 + 2) * 66 + 16;
    uint64_t *vals = malloc(n * 8), *out = malloc((n + 8) * 8);
    double *dv = malloc(n * 8), *dout = malloc((n + 8) * 8);
    uint64_t run = below(1000);
    for (size_t i = 0; i < n; i++) {
        run += rnd() >> (40 + below(24));
        vals[i] = run;
        dv[i] = (double)(int64_t)(rnd() >> 30) / 7.0;
    }
    uint8_t *buf = malloc(cap);
    size_t used = 0;
    int64_t l = ocv_pack_u64(vals, n, buf, cap, 0);
    CHECK(l > 0);
    uint8_t *e = exact(buf, (size_t)l);
    CHECK(ocv_unpack_u64(e, (size_t)l, out, n, &used) == 0 && used == (size_t)l && (g_quirk || memcmp(out, vals, n * 8) == 0));
    for (size_t cut = 0; cut < (size_t)l; cut += 1 + (size_t)l / 30) {
        uint8_t *b = exact(e, cut);
        (void)ocv_unpack_u64(b, cut, out, n, &used);
        uint64_t o8[8];
        uint32_t o32[8];
        int z = 0;
        (void)ocv_nibble_unpack8(b, cut, o8, &used);
        (void)ocv_unpack8_u32(b, cut, o32, &z, &used);
        free(b);
    }
    /* the histogram sink over this stream and a damaged copy of it */
    {
        uint64_t *last = calloc(n + 8, 8);
        uint8_t *hout = malloc(cap + 80);
        size_t off = 0, consumed = 0;
        int dropped = 0;
        (void)ocv_delta_diff_pack(e, (size_t)l, n, last, hout, cap + 80, &off, &dropped, &consumed);
        uint8_t *b = exact(e, (size_t)l);
        b[below((uint64_t)l)] ^= 0x5a;
        off = 0;
        (void)ocv_delta_diff_pack(b, (size_t)l, n, last, hout, cap + 80, &off, &dropped, &consumed);
        free(b); free(hout); free(last);
    }
    free(e);
    l = ocv_pack_u64_delta(vals, n, buf, cap);
    CHECK(l > 0);
    e = exact(buf, (size_t)l);
    CHECK(ocv_unpack_u64_delta(e, (size_t)l, out, n, &used) == 0 && (g_quirk || memcmp(out, vals, n * 8) == 0));
    for (int d = 0; d < 8; d++) {
        uint8_t *b = exact(e, (size_t)l);
        b[below((uint64_t)l)] ^= (uint8_t)(1u << below(8));
        (void)ocv_unpack_u64_delta(b, (size_t)l, out, n, &used);
        free(b);
    }
    free(e);
    uint8_t *xbuf = malloc(cap + 16);
    l = ocv_pack_f64_xor(dv, n, xbuf, cap + 16);
    CHECK(l > 0);
    e = exact(xbuf, (size_t)l);
    CHECK(ocv_unpack_f64_xor(e, (size_t)l, dout, n, &used) == 0 && (g_quirk || memcmp(dout, dv, n * 8) == 0));
    for (size_t cut = 0; cut < (size_t)l; cut += 1 + (size_t)l / 30) {
        uint8_t *b = exact(e, cut);
        (void)ocv_unpack_f64_xor(b, cut, dout, n, &used);
        free(b);
    }
    free(e); free(xbuf); free(buf); free(dout); free(dv); free(out); free(vals);
    uint8_t hb[64 * 66 + 64];
    uint64_t hv[64];
    for (int i = 0; i < 64; i++) hv[i] = 1000 - (uint64_t)i * below(10);
    (void)ocv_compress_geom_nonincreasing(64, 2.0, 1.5, 1, hv, 64, hb, sizeof(hb));
    (void)ocv_compress_geom_nonincreasing(64, 2.0, 1.5, 1, hv, 64, hb, 10);
}

int main(int argc, char **argv) {
    rng_state = argc > 1 ? strtoull(argv[1], NULL, 0) : 1;
    const int rounds = argc > 2 ? atoi(argv[2]) : 50;
    const int types[3] = {OCV_U32, OCV_U64, OCV_F32};
    for (int q = 0; q < 2; q++) { /* both tail rules */
        ocv_set_ref_tail_quirk(q);
        g_quirk = q;
        for (int r = 0; r < rounds; r++) {
            for (int t = 0; t < 3; t++) vector_round(types[t]);
            stream_round();
        }
    }
    ocv_set_ref_tail_quirk(0);
    printf("sanitize_driver: %d rounds x 2 tail rules, no report\n", rounds);
    return 0;
}
