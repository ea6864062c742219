// cvec_tailfix.hip — the reference crate's tail decoder, as an optional rule (cvec_set_tail_rule).
//
// The reference unpacks 32-bit octets with unpack8_u32_simd (src/nibblepack_simd.rs:511-585; 64-bit octets whose
// bits fit 32 take the same function, src/nibblepacking.rs:431-434).  With fewer than 34 bytes of its input slice
// left it does not use the SIMD preload but preload_u32x8_3_4_nibble / preload_u32x8_nibbles (:546-554), and the
// latter (:433-450) returns zeros for every field of an octet of 5 to 8 nibbles.  The slice a section's decoder
// gets runs to the end of the vector (src/section.rs:706, src/vector.rs:443), so the octets within the last 33
// bytes of a VECTOR decode to "all fields zero" in the crate, although the encoder stored their values.  The read
// kernels decode every octet by the bitstream rule of the format.  Under CVEC_TAIL_REFERENCE this pass runs behind
// the section kernel and makes the results what the crate would produce: one lane per vector looks at the sections
// that end within 33 bytes of the vector's end (normally the last one), decodes such a section serially under both
// rules, and where they differ patches the decoded output, or the section's match count and mask bytes.  A dropped
// octet leaves the XOR state of an Xor section as it was (XorSink::process with zeros, src/sink.rs:271-275), so
// every later octet of that section differs as well.
#include "cvec_sections.hpp"

namespace cvec {

template <int ETYPE>
__device__ inline bool tail_match(const PredArgs &pa, typename Elem<(ETYPE == CVEC_U64 ? 8 : 4)>::U x) {
    bool m = false;
    for (uint32_t p = 0; p < pa.count; p++) {
        if constexpr (ETYPE == CVEC_F32) m |= __uint_as_float((uint32_t)x) == __uint_as_float((uint32_t)pa.v[p]);
        else if constexpr (ETYPE == CVEC_U32) m |= (uint32_t)x == (uint32_t)pa.v[p];
        else m |= x == pa.v[p];
    }
    return m;
}

// nbits (<= 64) bits at bit position bitpos of the byte string at p.  `room` = bytes that may be read from p on
// (to the end of the blob): with nine of them behind the field's first byte the field comes from two unaligned
// loads, otherwise byte by byte (no read past the last byte used).
typedef uint64_t __attribute__((aligned(1))) tail_u64u;
__device__ inline uint64_t tail_bits(const uint8_t *p, uint32_t bitpos, uint32_t nbits, uint64_t room) {
    const uint32_t byte = bitpos >> 3, sh = bitpos & 7u, need = (sh + nbits + 7u) >> 3; // <= 9
    unsigned __int128 w = 0;
    if ((uint64_t)byte + 9u <= room) {
        w = (unsigned __int128)*(const tail_u64u *)(p + byte) | ((unsigned __int128)p[byte + 8] << 64);
    } else {
        for (uint32_t i = 0; i < need; i++) w |= (unsigned __int128)p[byte + i] << (8 * i);
    }
    w >>= sh;
    uint64_t v = (uint64_t)w;
    if (nbits < 64) v &= (1ull << nbits) - 1ull;
    return v;
}

template <int ETYPE>
__global__ __launch_bounds__(256) void k_ref_tail_fix(const uint8_t *__restrict__ bytes, const uint64_t *__restrict__ byte_off,
                                                      const uint64_t *__restrict__ sec_base, uint64_t n,
                                                      const uint64_t *__restrict__ desc, uint32_t *__restrict__ sec_res,
                                                      PredArgs pa, SinkArgs sa, int decode, uint8_t *__restrict__ masks,
                                                      void *__restrict__ out_, uint64_t n_bytes,
                                                      const uint64_t *__restrict__ tag, uint64_t cookie) {
    constexpr int W = ETYPE == CVEC_U64 ? 8 : 4;
    using U = typename Elem<W>::U;
    // descriptors reused from an earlier pass are only followed if they are this batch's (like k_sections)
    if (tag != nullptr && *tag != cookie) return;
    const uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n) return;
    const uint64_t vend = byte_off[v + 1];
    const uint64_t a = sec_base[v], b = sec_base[v + 1];
    for (uint64_t s = b; s-- > a;) {
        const uint64_t d = desc[s];
        const uint32_t kind = desc_kind(d);
        if (kind == K_SKIP) continue; // a slot the vector did not fill (its iteration stopped earlier)
        const uint64_t off = desc_off(d);
        const uint32_t len = desc_len(d);
        if (off + len > vend || vend - (off + len) >= 34) break; // earlier sections end even further from the end
        if (kind != K_NP && kind != K_DELTA && kind != K_XOR) continue;
        if ((sec_res[s] >> 16) != 0) continue; // the section failed to decode: nothing of it was delivered
        const uint8_t *sp = bytes + off;
        const uint64_t room = n_bytes - off; // bytes of the blob from the section's start on
        const uint32_t p_first = kind == K_DELTA ? (uint32_t)DELTA_HDR : 3u;
        // Most sections that end near the vector's end hold no octet the fallback would drop: walk the octet
        // headers first (two bytes per octet) and decode the section only if one is found.
        bool any_drop = false;
        {
            uint32_t p = p_first;
            for (uint32_t k = 0; k < (uint32_t)FIXED_LEN / 8 && p < len; k++) {
                const uint32_t m = sp[p];
                if (m == 0) { p += 1; continue; }
                const uint32_t nw = sp[p + 1];
                const uint32_t n4 = 4u * ((nw >> 4) + 1u), tz = 4u * (nw & 15u);
                any_drop |= vend - (off + p) < 34 && n4 >= 20u && n4 <= 32u && (W == 4 || n4 + tz <= 32u);
                p += 2u + ((n4 * (uint32_t)__popc(m) + 7u) >> 3);
            }
        }
        if (!any_drop) continue;
        uint32_t p = p_first;
        U base = 0;
        if (kind == K_DELTA) base = (U)tail_bits(sp + 4, 0, 8 * W, room - 4);
        U stf[8], str[8];
#pragma unroll
        for (int j = 0; j < 8; j++) stf[j] = str[j] = 0;
        int delta = 0;
        for (uint32_t k = 0; k < (uint32_t)FIXED_LEN / 8 && p < len; k++) {
            const uint32_t m = sp[p];
            U xf[8], xr[8];
#pragma unroll
            for (int j = 0; j < 8; j++) xf[j] = xr[j] = 0;
            uint32_t size = 1;
            if (m != 0) {
                const uint32_t nw = sp[p + 1];
                const uint32_t n4 = 4u * ((nw >> 4) + 1u), tz = 4u * (nw & 15u);
                // the reference's fallback: fewer than 34 bytes of the slice left, 5..8 nibbles, the 32-bit unpacker
                const bool drop = vend - (off + p) < 34 && n4 >= 20u && n4 <= 32u && (W == 4 || n4 + tz <= 32u);
                uint32_t idx = 0;
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    if ((m >> j) & 1u) {
                        const uint64_t f = tail_bits(sp + p + 2, idx * n4, n4, room - p - 2);
                        idx++;
                        U x;
                        if constexpr (W == 4) x = tz >= 32u ? 0u : (uint32_t)f << tz;
                        else x = f << tz;
                        xf[j] = x;
                        xr[j] = drop ? (U)0 : x;
                    }
                }
                size = 2u + ((n4 * (uint32_t)__popc(m) + 7u) >> 3);
            }
            bool differ = false;
            U vr[8];
#pragma unroll
            for (int j = 0; j < 8; j++) {
                U vf;
                if (kind == K_XOR) { stf[j] ^= xf[j]; str[j] ^= xr[j]; vf = stf[j]; vr[j] = str[j]; }
                else { vf = xf[j] + base; vr[j] = xr[j] + base; }
                if (vf != vr[j]) {
                    differ = true;
                    if (!decode) delta += (tail_match<ETYPE>(pa, vr[j]) ? 1 : 0) - (tail_match<ETYPE>(pa, vf) ? 1 : 0);
                }
            }
            if (differ) {
                if (decode) {
                    U *o = (U *)out_ + s * FIXED_LEN + 8 * k;
#pragma unroll
                    for (int j = 0; j < 8; j++) {
                        U y = vr[j];
                        if (sa.add_const) {
                            if constexpr (ETYPE == CVEC_F32)
                                y = __float_as_uint(__uint_as_float((uint32_t)y) + __uint_as_float((uint32_t)sa.base));
                            else y += (U)sa.base;
                        }
                        o[j] = y;
                    }
                } else if (masks != nullptr) {
                    uint32_t bits = 0;
#pragma unroll
                    for (int j = 0; j < 8; j++) bits |= (tail_match<ETYPE>(pa, vr[j]) ? 1u : 0u) << j;
                    masks[s * 32 + k] = (uint8_t)bits;
                }
            }
            p += size;
        }
        if (!decode && delta != 0) sec_res[s] = res_pack((uint32_t)((int)(sec_res[s] & 0xffffu) + delta), 0);
    }
}

hipError_t launch_ref_tail_fix(int etype, int decode, const cvec_dbatch_view &b, const uint64_t *desc, uint32_t *sec_res,
                               const PredArgs &pa, const SinkArgs &sa, uint8_t *d_masks, void *d_out, const uint64_t *tag,
                               uint64_t cookie, hipStream_t stream) {
    const uint64_t n = b.n_vectors;
    if (n == 0) return hipSuccess;
    const uint32_t blocks = (uint32_t)((n + 255) / 256);
    switch (etype) {
    case CVEC_U32: k_ref_tail_fix<CVEC_U32><<<blocks, 256, 0, stream>>>(b.d_bytes, b.d_byte_off, b.d_sec_base, n, desc, sec_res, pa, sa, decode, d_masks, d_out, b.n_bytes, tag, cookie); break;
    case CVEC_F32: k_ref_tail_fix<CVEC_F32><<<blocks, 256, 0, stream>>>(b.d_bytes, b.d_byte_off, b.d_sec_base, n, desc, sec_res, pa, sa, decode, d_masks, d_out, b.n_bytes, tag, cookie); break;
    case CVEC_U64: k_ref_tail_fix<CVEC_U64><<<blocks, 256, 0, stream>>>(b.d_bytes, b.d_byte_off, b.d_sec_base, n, desc, sec_res, pa, sa, decode, d_masks, d_out, b.n_bytes, tag, cookie); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

} // namespace cvec
