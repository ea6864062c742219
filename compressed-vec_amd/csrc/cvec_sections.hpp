// cvec_sections.hpp - device code shared by the two read kernels (the section-slot kernel k_sections in
// cvec_decode.hip and the vector-streaming kernel k_vstream in cvec_vstream.hip): the octet loop that
// walks and decodes the 32 octets of up to 16 sections staged in a wave's LDS region.
#pragma once

#include <type_traits>

#include "cvec_device.hpp"

namespace cvec {

// ------------------------------------------------------------------------------------------------
// k_sections
// ------------------------------------------------------------------------------------------------
constexpr int TILE = 16; // sections per chunk: one per quad of lanes
// Staging region of one wave: 18 KiB = 18 wave-wide 16-byte loads / LDS writes of 1 KiB; holds 16 worst-case
// u32/f32 sections (1100 B) plus the vector headers between them.  Four waves per workgroup, two
// workgroups per CU: 144 of the CU's 160 KiB.
// A second shape, 12 KiB regions at three workgroups per CU, serves count-only passes over batches whose
// sections are small enough that 16 of them still fit (dense integers, deltas): the octet loop is bound
// by VALU issue, and a third wave per SIMD fills the slots that two leave empty (u32 dense 0.180 ->
// 0.159 ms per GiB); 16 incompressible f32 sections do not fit 12 KiB and lose more lanes than that gains.
constexpr int REGION_PIECES = 18;
constexpr int REGION_PIECES_SMALL = 12;
constexpr int REGION_PIECES_TINY = 6; // 6 KiB regions at four workgroups per CU: sections of a few hundred bytes
// Workgroups of the section kernel that share a CU (= its waves per SIMD: a workgroup is one wave per SIMD), per
// region shape.  The kernel carries it as an explicit occupancy bound, which makes the compiler size the register
// allocation so that the hardware cannot place more; the build checks the kernel descriptors against it (Makefile).
constexpr int sec_wg_per_cu(int pieces) { return pieces <= REGION_PIECES_TINY ? 4 : (pieces <= REGION_PIECES_SMALL ? 3 : 2); }

// The last 64 bytes of a region stay zero: sections without octets (Null, Const, skipped slots) run
// the octet loop over them (32 empty octets), which keeps the loop free of per-kind masking.
constexpr uint32_t region_span(int pieces) { return (uint32_t)pieces * 1024u - 64u; }
constexpr int SEC_WAVES = 4;
constexpr int SEC_THREADS = SEC_WAVES * 64;

// MODE_FILTER_BITS: count only, one predicate value whose IEEE == is the same as equality of the bit
// patterns (any integer; a float that is neither zero nor NaN): matching needs no compare instruction.
// MODE_FILTER_MASKS_BITS: the same test, with the per-section masks.
enum { MODE_FILTER = 0, MODE_DECODE = 1, MODE_FILTER_MASKS = 2, MODE_FILTER_BITS = 3, MODE_FILTER_MASKS_BITS = 4 };
constexpr bool mode_has_masks(int mode) { return mode == MODE_FILTER_MASKS || mode == MODE_FILTER_MASKS_BITS; }
constexpr bool mode_bit_compare(int mode) { return mode == MODE_FILTER_BITS || mode == MODE_FILTER_MASKS_BITS; }

template <int PIECES>
struct SecShared {
    __attribute__((aligned(16))) uint8_t data[SEC_WAVES][PIECES * 1024];
};

typedef const __attribute__((address_space(3))) uint32_t *lds_u32p;

// Misaligned LDS accesses take a slow path on gfx950 (~16x the cycles of an aligned one), so every
// LDS read here is a single byte or naturally aligned dwords that are then funnel-shifted.
template <int W>
__device__ inline typename Elem<W>::U lds_load_unaligned(uint32_t addr) {
    lds_u32p q = (lds_u32p)(uintptr_t)(addr & ~3u);
    const uint32_t s = (addr & 3u) << 3;
    if constexpr (W == 4) return __builtin_amdgcn_alignbit(q[1], q[0], s);
    else return (uint64_t)__builtin_amdgcn_alignbit(q[1], q[0], s) |
                ((uint64_t)__builtin_amdgcn_alignbit(q[2], q[1], s) << 32);
}

// NPRED = number of predicate values compared, a compile-time bucket (1, 2, 4, 8, 16): the values sit
// in scalar registers and the compares are straight-line code.  (A run-time loop over pa.count reads
// the values from the kernel arguments inside the octet loop, and the wait for those scalar loads is
// the same counter the loop's LDS pipeline runs on: 3x slower for four values.)  The launcher pads
// the list to the bucket with copies of its first value.
template <int ETYPE, int NPRED>
__device__ inline bool pred_match(const PredArgs &pa, typename Elem<(ETYPE == CVEC_U64 ? 8 : 4)>::U x) {
    bool m = false;
#pragma unroll
    for (int p = 0; p < NPRED; p++) {
        if constexpr (ETYPE == CVEC_F32) m |= __uint_as_float((uint32_t)x) == __uint_as_float((uint32_t)pa.v[p]); // IEEE ==
        else if constexpr (ETYPE == CVEC_U32) m |= (uint32_t)x == (uint32_t)pa.v[p];
        else m |= x == pa.v[p];
    }
    return m;
}

// Exact status of a section whose octet walk ran past its end or met an over-wide octet: the first
// failing octet in the order of the reference's checks (src/nibblepack_simd.rs:515-557 for 32-bit
// elements, src/nibblepacking.rs:474-500 for 64-bit): empty input, length < 3, n > 8, payload too
// long.  Rare path (malformed input only); plain byte reads.
template <int W>
__device__ __noinline__ uint32_t walk_status(const uint8_t *data, uint32_t p, uint32_t end, bool raw) {
    for (int k = 0; k < FIXED_LEN / 8; k++) {
        const int rem = (int)end - (int)p;
        if (rem == 0 && raw) return 0; // a stream's last block may hold fewer than 32 octets
        if (rem <= 0) return CVEC_NOT_ENOUGH_SPACE;
        const uint32_t mask = data[p];
        if (mask == 0) { p += 1; continue; }
        if (W == 4 && rem < 3) return CVEC_NOT_ENOUGH_SPACE;
        if (rem < 2) return CVEC_NOT_ENOUGH_SPACE;
        const uint32_t hn = (uint32_t)data[p + 1] >> 4;
        if (W == 4 && hn > 7) return CVEC_INVALID_FORMAT;
        const int size = (int)(2 + (((hn + 1) * __popc(mask) + 1) >> 1));
        if (size > rem) return CVEC_NOT_ENOUGH_SPACE;
        p += size;
    }
    return 0;
}

// The 32-byte mask of a section (byte = octet, LSB = first element, src/filter.rs:46, :95-101) from what the four
// lanes of its quad collected.  Lane s of the quad holds its two match bits of octet k at bits 2k, 2k + 1 of the
// 64-bit word (a_hi : a_lo), one shift-or per element and octet inside the octet loop; here, once per section, the
// quad transposes: lane r ends up with mask bytes 8r .. 8r + 7 (out_lo = bytes 8r .. 8r + 3), so that the quad stores
// 32 contiguous bytes and a chunk of 16 sections 512.  Lane r takes from every lane s the sixteen bits of octets
// 8r .. 8r + 7 (quad broadcasts), packs the four lanes' bytes side by side (byte s = the pairs of four octets from
// lane s) and transposes that 4 x 4 matrix of bit pairs with two masked swaps.
__device__ __forceinline__ void quad_mask_bytes(uint32_t a_lo, uint32_t a_hi, uint32_t q, uint32_t &out_lo, uint32_t &out_hi) {
    uint32_t h[4];
    const bool upper = (q & 2u) != 0;
    const uint32_t sh = (q & 1u) << 4;
#define CVEC_QUAD_BCAST(S, CTRL)                                                                                       \
    {                                                                                                                  \
        const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a_lo, CTRL, 0xf, 0xf, false);                \
        const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a_hi, CTRL, 0xf, 0xf, false);                \
        h[S] = __builtin_amdgcn_ubfe(upper ? hi : lo, sh, 16u);                                                        \
    }
    CVEC_QUAD_BCAST(0, 0x00) CVEC_QUAD_BCAST(1, 0x55) CVEC_QUAD_BCAST(2, 0xaa) CVEC_QUAD_BCAST(3, 0xff)
#undef CVEC_QUAD_BCAST
    const uint32_t x = h[0] | (h[1] << 16), y = h[2] | (h[3] << 16);
    uint32_t v[2] = {__builtin_amdgcn_perm(y, x, 0x06040200u), __builtin_amdgcn_perm(y, x, 0x07050301u)};
#pragma unroll
    for (int i = 0; i < 2; i++) {
        // element (lane s, octet j) sits at bit 8s + 2j and goes to bit 8j + 2s: swap bit 0 of s and j, then bit 1
        uint32_t w = v[i];
        w = (w & ~(0x00cc00ccu | (0x00cc00ccu << 6))) | ((w & 0x00cc00ccu) << 6) | ((w >> 6) & 0x00cc00ccu);
        w = (w & ~(0x0000f0f0u | (0x0000f0f0u << 12))) | ((w & 0x0000f0f0u) << 12) | ((w >> 12) & 0x0000f0f0u);
        v[i] = w;
    }
    out_lo = v[0];
    out_hi = v[1];
}

typedef __amdgpu_buffer_rsrc_t out_rsrc_t;
// Cache policy of the decoded output's stores.  In MODE_DECODE the NPRED template parameter (no predicate there)
// carries the choice: 1 = default policy, 2 = non-temporal.  The launcher only asks for 1: the non-temporal hint was
// faster for dense u32 in a loop that staged 16 sections and stored 8 bytes per lane (2.25 -> 2.62 TB/s), slower
// where reads and writes balance (f32 XOR: 2.14 -> 2.05 TB/s) or the output is long runs of one value (sparse
// vectors: 4.4 -> 2.7 TB/s), and it is slower with every decode loop that exists now.
template <int MODE, int NPRED> struct OutPolicy { static constexpr int AUX = (MODE == 1 /* MODE_DECODE */ && NPRED == 2) ? 2 : 0; };
constexpr uint32_t OOFF_NONE = 0xffff0000u; // beyond any output descriptor (their ranges end below it): the store is dropped

// MODE_DECODE addressing of k_sections: a buffer descriptor over the chunk's 16 consecutive section
// slots of `out` (`slot` = the lane's own slot) and the lane's offset in it.
template <int W, int MODE>
__device__ __forceinline__ out_rsrc_t chunk_out_rsrc(void *out_, uint64_t slot, int lane, uint32_t &ooff) {
    using U = typename Elem<W>::U;
    const uint64_t dslot = slot - ((uint32_t)lane >> 2); // the chunk's first slot, in every lane
    const uint64_t slot0 = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(dslot >> 32)) << 32) |
                           (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)dslot);
    ooff = (((uint32_t)lane >> 2) * FIXED_LEN + 2 * ((uint32_t)lane & 3u)) * (uint32_t)W;
    return __builtin_amdgcn_make_buffer_rsrc((void *)((U *)out_ + slot0 * FIXED_LEN), 0,
                                             MODE == MODE_DECODE ? TILE * FIXED_LEN * W : 0, 0x00020000);
}

// One chunk (up to 16 sections staged in the wave's LDS region) by one wave.  Lane = 4*sec + q: the
// four lanes of a quad walk the octets of section `sec` together (same addresses: LDS broadcasts)
// and lane q extracts elements 2q and 2q+1 of every octet, so nothing has to cross lanes and no
// octet table is needed.  A step of the loop handles octet k of all 16 sections:
//   - the walk chain: two byte reads (always aligned), octet size, next cursor; the reads of the
//     next octet are issued as soon as the cursor is known,
//   - in the shadow of that latency the two fields of the current octet are pulled out of the
//     bitstream with aligned dword reads and funnel shifts (plain bitstream rule of the format,
//     reference src/nibblepacking.rs:436-470, src/nibblepack_simd.rs:571-585), run through the
//     XorSink state (kept in registers: octet k of a lane XORs onto octet k-1 of the same lane,
//     src/sink.rs:271-275, src/section.rs:707) or the Delta base add (src/section.rs:636-641) and
//     the fused epilogue.
// Section kinds are folded into the same loop: a Null or Const section runs it with every mask
// forced to 0 and its value preloaded into the XOR state / the base.
//
// This is the common path: it assumes that no stored octet of a 32-bit section claims more than 8
// nibbles or 8 or more trailing zero nibbles (never produced by an encoder: the value would not fit
// 32 bits) and only records whether one did (the return value, per section); the caller then decodes
// such a section again with section_exact below, which handles both exactly.
//
// ZPAD = offset in the region of 64 zero bytes: sections without octets run the loop over them.
// MODE_DECODE output goes through the buffer descriptor `orsrc`; `ooff` is the lane's byte offset in it
// for elements 2q and 2q+1 of octet 0 of its section (out of range for lanes that must store nothing).
//
// TAILREF: the reference crate's tail decoder (cvec_set_tail_rule).  unpack8_u32_simd leaves its SIMD preload when
// fewer than 34 bytes of its input slice are left (src/nibblepack_simd.rs:546-554), and the fallback for octets of
// 5 to 8 nibbles, preload_u32x8_nibbles (:433-450), returns zeros.  The slice a section's decoder is handed runs to
// the end of the VECTOR (src/section.rs:706, src/vector.rs:443), and 64-bit octets whose bits fit 32 take the same
// unpacker (src/nibblepacking.rs:431-434).  So the crate decodes such an octet to "no lane stored": here its mask
// is cleared for the extraction (the walk still uses the real one), which also leaves an Xor section's state as
// the crate's XorSink leaves it (src/sink.rs:271-275).  `tail_from` = the first LDS address from which an octet
// start is within 33 bytes of the vector's end (beyond every address for sections that end further away).
template <int ETYPE, int MODE, int NPRED, uint32_t ZPAD, int TAILREF = 0, bool PAIRS = false>
__device__ __forceinline__ bool process_chunk(const uint8_t *data, uint32_t lds_base, uint32_t kind, uint32_t rel,
                                              uint32_t len, bool live, int lane, const PredArgs &pa,
                                              const SinkArgs &sa, uint32_t &sec_result, uint32_t &mask_lo,
                                              uint32_t &mask_hi, out_rsrc_t orsrc, uint32_t ooff_in,
                                              int32_t tail_from = 0x7fffffff) {
    constexpr int W = ETYPE == CVEC_U64 ? 8 : 4;
    using U = typename Elem<W>::U;
    const uint32_t q = lane & 3;
    const bool isoct = live && (kind == K_NP || kind == K_DELTA || kind == K_XOR || kind == K_RAW);
    const bool isconst = live && kind == K_CONST, isdelta = live && kind == K_DELTA;
    const bool zeroes = kind == K_NULL; // the section reaches the sink through process_zeroes()
    const uint32_t end = rel + len;
    const uint32_t p0 = isoct ? rel + (kind == K_DELTA ? DELTA_HDR : (kind == K_RAW ? 0u : 3u)) : ZPAD;
    // Const value (section byte 1) or Delta base (section byte 4)
    U v = 0;
    if (isconst || isdelta) v = lds_load_unaligned<W>(lds_base + rel + (isconst ? 1u : 4u));
    U st0 = 0, st1 = 0; // XOR state of the lane's two element columns / the additive base
    if constexpr (ETYPE == CVEC_F32) { st0 = st1 = isconst ? v : (U)0; }
    const U base = (isconst || isdelta) ? v : (U)0;
    // MODE_FILTER_BITS: the predicate's bits are XORed into the state (f32) or moved to the other side of
    // the Delta add (integers), so that an element matches iff its word is zero
    U tgt = 0;
    if constexpr (mode_bit_compare(MODE)) {
        if constexpr (ETYPE == CVEC_F32) { st0 ^= (U)pa.v[0]; st1 = st0; }
        else tgt = (U)pa.v[0] - base;
    }
    uint32_t mism = 0;
    // mask modes: the lane's two match bits (MODE_FILTER_MASKS) or mismatch bits (MODE_FILTER_MASKS_BITS) of octet k at
    // bits 2k, 2k + 1 of (acc_hi : acc_lo); the quad turns them into mask bytes once, behind the loop
    uint32_t acc_lo = 0, acc_hi = 0;
    mask_lo = 0; mask_hi = 0;
    const uint32_t lowmask = (1u << (2 * q)) - 1u;
    uint32_t cnt = 0, racc = 0;
    const uint32_t q2 = 2 * q;
    const uint32_t ooff = (live && kind != K_SKIP && kind != K_ILLEGAL) ? ooff_in : OOFF_NONE;
    const bool addc = sa.add_const != 0;
    // 32-bit decode with 16-byte stores: byte offset of the lane's piece of an octet pair (see the store burst below):
    // half q >> 1 of the pair's first (even q) or second (odd q) octet
    const bool oddq = (q & 1u) != 0;
    const uint32_t ooff16 = ooff - 8u * q + 16u * (q >> 1) + 32u * (q & 1u);

    // The loop is software-pipelined by one octet: iteration k runs the front half of octet k (walk
    // chain, header reads of octet k+1, address + reads of the lane's fields, per-octet constants) and
    // the back half of octet k-1 (extraction, XOR / Delta, epilogue), so every LDS read has a whole
    // iteration to land and the serial chain header -> size -> next header is as short as it can be.
    // All LDS reads are inline asm with counted waits (LDS returns in order): per iteration the wave
    // issues 2 header reads, then NDATA field reads.
    constexpr int NDATA = W == 4 ? 2 : 3;
    struct Oct { // what the back half needs of an octet
        uint64_t d01, d23;
        uint32_t d2, d4, s, sh1, sm0, sm1, t4;
        U fm;
    };
    uint32_t pa_ = lds_base + p0;
    uint32_t mask, nw;
    asm volatile("ds_read_u8 %0, %2\n\tds_read_u8 %1, %2 offset:1\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(mask), "=&v"(nw)
                 : "v"(pa_)
                 : "memory");

    // front half of the octet whose header is in (mask, nw); leaves the next header in flight
    auto front = [&](Oct &o, uint32_t &mask_n, uint32_t &nw_n) {
        // ---- walk chain: octet bytes = 1 if mask == 0, else 2 + (n*cnt + 1)/2; then the reads of the
        //      next octet's header (after the last octet: a harmless read of the bytes that follow) ------
        //      in eighths of a byte with the field width n4 = 4n that the extraction needs anyway:
        //      (n4 * cnt + 20) / 8 = 2 + (n * cnt + 1) / 2, and 8 / 8 for the empty octet
        uint32_t n4;                        // field width in bits, 4 * ((nw >> 4) + 1), as ONE instruction
        asm("v_lshl_add_u32 %0, %1, 2, 4" : "=v"(n4) : "v"(nw >> 4));
        const uint32_t pc = __popc(mask);
        const uint32_t pa_cur = pa_;
        pa_ += ((uint32_t)__umul24(n4, pc) + (mask ? 20u : 8u)) >> 3;
        // mask and nw pass through the statement so that everything below is scheduled behind it
        asm volatile("ds_read_u8 %0, %4\n\tds_read_u8 %1, %4 offset:1"
                     : "=&v"(mask_n), "=&v"(nw_n), "+v"(mask), "+v"(nw)
                     : "v"(pa_)
                     : "memory");
        // ---- address of the lane's two fields, their reads ---------------------------------------------
        // bit address of the first field: behind the two header bytes, idx fields of n4 bits further on
        const uint32_t B = (uint32_t)__umul24(__popc(mask & lowmask), n4) + ((pa_cur << 3) + 16u);
        o.s = B; // funnel shift; only bits 4:0 are used
        const uint32_t qa = (B >> 3) & ~3u;
        if constexpr (W == 4) {
            asm volatile("ds_read2_b32 %0, %2 offset1:1\n\tds_read_b32 %1, %2 offset:8"
                         : "=&v"(o.d01), "=&v"(o.d2)
                         : "v"(qa)
                         : "memory");
        } else {
            asm volatile("ds_read2_b32 %0, %3 offset1:1\n\tds_read2_b32 %1, %3 offset0:2 offset1:3\n\tds_read_b32 %2, %3 offset:16"
                         : "=&v"(o.d01), "=&v"(o.d23), "=&v"(o.d4)
                         : "v"(qa)
                         : "memory");
        }
        // ---- per-octet constants ---------------------------------------------------------------------------
        o.t4 = nw << 2; // trailing zero bits (a 32-bit shift uses bits 4:0 of its count; bits 5:4 only matter to section_exact and 64-bit)
        o.sm0 = (uint32_t)__builtin_amdgcn_sbfe((int)mask, q2, 1u);     // all ones iff element 2q is stored
        o.sm1 = (uint32_t)__builtin_amdgcn_sbfe((int)mask, q2 + 1u, 1u);
        if constexpr (W == 4) racc |= nw & (o.sm0 | o.sm1); // the quad's lanes cover every stored element
        o.sh1 = n4 & o.sm0; // the second field lies n4 bits further on if the first one is stored
        if constexpr (W == 4) {
            // a left shift of 32 or more clears the lane, as the AVX2 variable shift of the reference does
            o.fm = 0xffffffffu >> ((32u - n4) & 31u);
        } else {
            o.fm = ~0ull >> ((64u - n4) & 63u);
        }
        if constexpr (TAILREF) {
            // An octet of 5 .. 8 nibbles within 33 bytes of the vector's end yields zeros: its fields are masked out
            // (the walk and the stored-element bits above keep the real mask).  5 nibbles or more is n4 >= 20; an
            // octet of more than 8 never gets here with a result that counts (32-bit sections: it is flagged for
            // section_exact above and fails there; 64-bit ones take the 64-bit unpacker, tested by 4n + tz <= 32).
            bool drop = (int32_t)pa_cur >= tail_from && n4 >= 20u;
            if constexpr (W == 8) drop = drop && n4 + ((nw & 15u) << 2) <= 32u;
            o.fm = drop ? (U)0 : o.fm;
        }
    };

#ifndef CVEC_HOLD
#define CVEC_HOLD 4
#endif
#ifndef CVEC_HOLD_PAIRS
#define CVEC_HOLD_PAIRS 4
#endif
    constexpr int HOLD = PAIRS ? CVEC_HOLD_PAIRS : CVEC_HOLD; // octets of decoded values held back per store burst (MODE_DECODE)
    U hold[2 * HOLD];
    // back half of octet k: `newer` LDS reads were issued after its field reads
    auto back = [&](Oct &o, int k, auto newer) {
        U x0, x1;
        if constexpr (W == 4) {
            // the octet's values pass through the wait: the compiler may neither use the loaded registers
            // ahead of it nor move the extraction somewhere else
            asm volatile("s_waitcnt lgkmcnt(%c9)"
                         : "+v"(o.d01), "+v"(o.d2), "+v"(o.fm), "+v"(o.sh1), "+v"(o.sm0), "+v"(o.sm1), "+v"(o.s), "+v"(o.t4),
                           "+v"(racc)
                         : "i"(decltype(newer)::value)
                         : "memory");
            const uint32_t d0 = (uint32_t)o.d01, d1 = (uint32_t)(o.d01 >> 32);
            const uint32_t lo = __builtin_amdgcn_alignbit(d1, d0, o.s), hi = __builtin_amdgcn_alignbit(o.d2, d1, o.s);
            const uint32_t w1 = (uint32_t)((((uint64_t)hi << 32) | lo) >> o.sh1);
            x0 = ((lo & o.fm) & o.sm0) << (o.t4 & 31u);
            x1 = ((w1 & o.fm) & o.sm1) << (o.t4 & 31u);
        } else {
            asm volatile("s_waitcnt lgkmcnt(%c7)"
                         : "+v"(o.d01), "+v"(o.d23), "+v"(o.d4), "+v"(o.fm), "+v"(o.sh1), "+v"(o.sm0), "+v"(o.sm1)
                         : "i"(decltype(newer)::value)
                         : "memory");
            const uint32_t d0 = (uint32_t)o.d01, d1 = (uint32_t)(o.d01 >> 32), d2 = (uint32_t)o.d23,
                           d3 = (uint32_t)(o.d23 >> 32);
            const uint32_t w0 = __builtin_amdgcn_alignbit(d1, d0, o.s), w1 = __builtin_amdgcn_alignbit(d2, d1, o.s),
                           w2 = __builtin_amdgcn_alignbit(d3, d2, o.s), w3 = __builtin_amdgcn_alignbit(o.d4, d3, o.s);
            const uint64_t f0 = ((uint64_t)w1 << 32) | w0;
            const unsigned __int128 win = ((unsigned __int128)(((uint64_t)w3 << 32) | w2) << 64) | f0;
            const uint64_t f1 = (uint64_t)(win >> o.sh1);
            x0 = ((f0 & o.fm) & (uint64_t)(int64_t)(int32_t)o.sm0) << (o.t4 & 60u);
            x1 = ((f1 & o.fm) & (uint64_t)(int64_t)(int32_t)o.sm1) << (o.t4 & 60u);
        }
        if constexpr (mode_bit_compare(MODE)) {
            if constexpr (ETYPE == CVEC_F32) { st0 ^= x0; st1 ^= x1; x0 = st0; x1 = st1; }
            else { x0 ^= tgt; x1 ^= tgt; }
            // zero word = match: count the mismatches (min(x, 1)), no compare / carry instructions
            // (as asm: the compiler rewrites min(x, 1) into a compare + select + add-with-carry chain)
            uint32_t w0, w1, m0, m1;
            if constexpr (W == 4) { w0 = (uint32_t)x0; w1 = (uint32_t)x1; }
            else { w0 = (uint32_t)x0 | (uint32_t)(x0 >> 32); w1 = (uint32_t)x1 | (uint32_t)(x1 >> 32); }
            asm("v_min_u32 %0, 1, %1" : "=v"(m0) : "v"(w0));
            asm("v_min_u32 %0, 1, %1" : "=v"(m1) : "v"(w1));
            if constexpr (MODE == MODE_FILTER_BITS) {
                mism += m0 + m1;
            } else {
                if (k < 16) { acc_lo |= m0 << (2 * k); acc_lo |= m1 << (2 * k + 1); }
                else { acc_hi |= m0 << (2 * k - 32); acc_hi |= m1 << (2 * k - 31); }
            }
            return;
        }
        if constexpr (ETYPE == CVEC_F32) {
            st0 ^= x0; st1 ^= x1;
            x0 = st0; x1 = st1;
        } else {
            x0 += base; x1 += base;
        }
        // ---- epilogue ----------------------------------------------------------------------------------------
        if constexpr (MODE == MODE_FILTER || MODE == MODE_FILTER_MASKS) {
            const bool m0 = pred_match<ETYPE, NPRED>(pa, x0), m1 = pred_match<ETYPE, NPRED>(pa, x1);
            if constexpr (MODE == MODE_FILTER_MASKS) {
                const uint32_t bits = (m0 ? 1u : 0u) | (m1 ? 2u : 0u);
                if (k < 16) acc_lo |= bits << (2 * k); else acc_hi |= bits << (2 * k - 32);
            } else {
                cnt += (m0 ? 1u : 0u) + (m1 ? 1u : 0u);
            }
        } else {
            if (addc) {
                // AddConstSink in front of VecSink (src/sink.rs:304-317): process_zeroes forwards the
                // base itself; everything else is x + base (f32: IEEE add, ints: wrapping).
                if constexpr (ETYPE == CVEC_F32) {
                    const float bf = __uint_as_float((uint32_t)sa.base);
                    x0 = zeroes ? (uint32_t)sa.base : __float_as_uint(__uint_as_float(x0) + bf);
                    x1 = zeroes ? (uint32_t)sa.base : __float_as_uint(__uint_as_float(x1) + bf);
                } else {
                    x0 += (U)sa.base; x1 += (U)sa.base;
                }
            }
            // The lane's 8 or 16 bytes are a quarter of a 32- or 64-byte piece of the output; four octets
            // are held back and stored together, so that the pieces of one 128-byte line leave the wave
            // within a few cycles of each other (spread over the octet loop they reach memory as partial
            // line writes: 3.4 TB/s instead of 4.9 for the same pattern issued back to back).
            // The stores are unconditional buffer stores (lanes without output aim past the end of the
            // descriptor and are dropped): straight-line code lets the compiler count them, so the wait
            // for the next chunk's prefetched bytes does not also wait for this chunk's stores.
            hold[2 * (k % HOLD)] = x0;
            hold[2 * (k % HOLD) + 1] = x1;
            if (k % HOLD == HOLD - 1) {
                typedef uint32_t __attribute__((__vector_size__(16))) u32x4;
                if constexpr (W == 4 && PAIRS) {
                    // 32-bit elements, 16-byte stores: the lanes of a pair (q, q ^ 1) trade halves of two consecutive
                    // octets: the even lane stores 16 bytes of the first octet (its own 8 and the odd lane's), the odd
                    // lane 16 bytes of the second (the even lane's 8 and its own), so a quad's store covers 64
                    // contiguous bytes, as for 64-bit elements, with half the store instructions.
                    static_assert(HOLD % 2 == 0, "octets are stored in pairs");
#pragma unroll
                    for (int j = 0; j < HOLD; j += 2) {
                        const uint32_t a0 = hold[2 * j], a1 = hold[2 * j + 1], b0 = hold[2 * j + 2], b1 = hold[2 * j + 3];
                        const uint32_t pa0 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a0, 0xb1 /* quad_perm:[1,0,3,2] */, 0xf, 0xf, true);
                        const uint32_t pa1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a1, 0xb1, 0xf, 0xf, true);
                        const uint32_t pb0 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)b0, 0xb1, 0xf, 0xf, true);
                        const uint32_t pb1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)b1, 0xb1, 0xf, 0xf, true);
                        const u32x4 y = oddq ? u32x4{pb0, pb1, b0, b1} : u32x4{a0, a1, pa0, pa1};
                        const uint32_t at = ooff16 + (uint32_t)(32 * (k - (HOLD - 1) + j));
                        __builtin_amdgcn_raw_buffer_store_b128(y, orsrc, at, 0, OutPolicy<MODE, NPRED>::AUX);
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < HOLD; j++) {
                        const U y0 = hold[2 * j], y1 = hold[2 * j + 1];
                        const uint32_t at = ooff + (uint32_t)(8 * (k - (HOLD - 1) + j) * W);
                        if constexpr (W == 4) {
                            typedef uint32_t __attribute__((__vector_size__(8))) u32x2;
                            __builtin_amdgcn_raw_buffer_store_b64(u32x2{y0, y1}, orsrc, at, 0, OutPolicy<MODE, NPRED>::AUX);
                        } else {
                            __builtin_amdgcn_raw_buffer_store_b128(
                                u32x4{(uint32_t)y0, (uint32_t)(y0 >> 32), (uint32_t)y1, (uint32_t)(y1 >> 32)}, orsrc, at, 0, OutPolicy<MODE, NPRED>::AUX);
                        }
                    }
                }
            }
        }
    };

    Oct prev, cur;
    uint32_t mask_n, nw_n;
    front(prev, mask_n, nw_n);
    asm volatile("s_waitcnt lgkmcnt(%c2)" : "+v"(mask_n), "+v"(nw_n) : "i"(NDATA) : "memory"); // header of octet 1
    mask = mask_n; nw = nw_n;
    // Fully unrolled on purpose: the field reads of an octet are in flight from one iteration to the
    // next, and only straight-line code guarantees that the compiler never copies a register (a loop
    // back-edge copy) whose asm load has not landed yet.
#pragma unroll
    for (int k = 1; k < FIXED_LEN / 8; k++) {
        front(cur, mask_n, nw_n);
        // octet k-1: behind its field reads came 2 header reads and NDATA field reads
        back(prev, k - 1, std::integral_constant<int, 2 + NDATA>());
        prev = cur;
        // the next header has landed once only this iteration's field reads are outstanding; the
        // step's results pass through the wait so that the work above is issued ahead of it
        if constexpr (mode_has_masks(MODE))
            asm volatile("s_waitcnt lgkmcnt(%c6)" : "+v"(mask_n), "+v"(nw_n), "+v"(acc_lo), "+v"(acc_hi), "+v"(st0), "+v"(st1) : "i"(NDATA) : "memory");
        else
            asm volatile("s_waitcnt lgkmcnt(%c6)" : "+v"(mask_n), "+v"(nw_n), "+v"(cnt), "+v"(mism), "+v"(st0), "+v"(st1) : "i"(NDATA) : "memory");
        mask = mask_n; nw = nw_n;
    }
    back(prev, FIXED_LEN / 8 - 1, std::integral_constant<int, 0>());
    // ---- per-section result ------------------------------------------------------------------------------------
    uint32_t status = 0;
    if (kind == K_ILLEGAL) status = CVEC_INVALID_FORMAT;
    const uint32_t p_end = pa_ - lds_base;
    if (isoct && p_end > end) status = walk_status<W>(data, p0, end, kind == K_RAW);
    uint32_t count;
    if constexpr (MODE == MODE_DECODE) {
        count = FIXED_LEN;
    } else {
        if constexpr (MODE == MODE_FILTER_BITS) cnt = 2 * (FIXED_LEN / 8) - mism; // the lane saw 64 elements
        if constexpr (mode_has_masks(MODE)) {
            if constexpr (MODE == MODE_FILTER_MASKS_BITS) { acc_lo = ~acc_lo; acc_hi = ~acc_hi; }
            // null_mask() / a section behind a closed gate
            if (zeroes) acc_lo = acc_hi = pa.match_zero ? ~0u : 0u;
            if (kind == K_GATED) acc_lo = acc_hi = 0u;
            quad_mask_bytes(acc_lo, acc_hi, q, mask_lo, mask_hi);
            cnt = (uint32_t)__popc(mask_lo) + (uint32_t)__popc(mask_hi);
        }
        cnt += __builtin_amdgcn_update_dpp(0u, cnt, 0xb1 /* quad_perm:[1,0,3,2] */, 0xf, 0xf, true);
        cnt += __builtin_amdgcn_update_dpp(0u, cnt, 0x4e /* quad_perm:[2,3,0,1] */, 0xf, 0xf, true);
        // null_mask(): all ones iff the predicate matches zero (src/filter.rs:104-106)
        count = zeroes ? (pa.match_zero ? FIXED_LEN : 0) : cnt;
        if (kind == K_GATED) count = 0;
    }
    sec_result = kind == K_SKIP ? res_pack(0, 0) : (status ? res_pack(0, status) : res_pack(count, 0));
    return isoct && (racc & 0x88u) != 0;
}


// The exact decode of one octet-bearing 32-bit section by its quad of lanes, for the sections that
// process_chunk flagged: an octet that claims more than 8 nibbles makes the section InvalidFormat
// (src/nibblepack_simd.rs:556-557, found by walk_status in the order of the reference's checks), and a
// shift by 8 or more trailing zero nibbles clears the lanes, as the reference's variable shift does
// (src/nibblepack_simd.rs:581).  A plain rolled loop over the 32 octets: such sections only come from damaged or
// hand-made input, so its speed does not matter and its small register footprint does.  Same results
// layout as process_chunk.  Call with all four lanes of the quad, for kinds NP / DELTA / XOR / RAW only.
template <int ETYPE, int MODE, int NPRED, int TAILREF = 0>
__device__ inline void section_exact(const uint8_t *data, uint32_t lds_base, uint32_t kind, uint32_t rel, uint32_t len,
                                     int lane, const PredArgs &pa, const SinkArgs &sa, uint32_t &sec_result,
                                     uint32_t &mask_lo, uint32_t &mask_hi, out_rsrc_t orsrc, uint32_t ooff,
                                     int32_t tail_from = 0x7fffffff) {
    static_assert(ETYPE != CVEC_U64, "64-bit sections have no inexact case");
    const uint32_t q = (uint32_t)lane & 3u;
    const uint32_t end = rel + len;
    const uint32_t p0 = rel + (kind == K_DELTA ? (uint32_t)DELTA_HDR : (kind == K_RAW ? 0u : 3u));
    const uint32_t base = kind == K_DELTA ? lds_load_unaligned<4>(lds_base + rel + 4u) : 0u;
    const bool addc = sa.add_const != 0;
    uint32_t st0 = 0, st1 = 0, cnt = 0, hmax = 0, p = p0, acc_lo = 0, acc_hi = 0;
    mask_lo = 0; mask_hi = 0;
    for (uint32_t k = 0; k < (uint32_t)FIXED_LEN / 8; k++) {
        const uint32_t m = data[p], nw = data[p + 1];
        const uint32_t n4 = 4u * ((nw >> 4) + 1u), pc = (uint32_t)__popc(m);
        if (m) hmax = max(hmax, nw >> 4);
        uint32_t fm = (nw & 8u) ? 0u : 0xffffffffu >> ((32u - n4) & 31u);
        if constexpr (TAILREF) {
            if ((int32_t)(lds_base + p) >= tail_from && (nw >> 6) == 1u) fm = 0u; // the crate's tail decoder: see process_chunk
        }
        uint32_t x[2];
#pragma unroll
        for (uint32_t j = 0; j < 2; j++) {
            const uint32_t e = 2u * q + j;
            const uint32_t bit = 8u * (p + 2u) + (uint32_t)__popc(m & ((1u << e) - 1u)) * n4;
            const uint64_t win = lds_load_unaligned<8>(lds_base + (bit >> 3));
            const uint32_t f = (uint32_t)(win >> (bit & 7u));
            x[j] = ((m >> e) & 1u) ? (f & fm) << ((nw << 2) & 31u) : 0u;
        }
        if constexpr (ETYPE == CVEC_F32) { st0 ^= x[0]; st1 ^= x[1]; x[0] = st0; x[1] = st1; }
        else { x[0] += base; x[1] += base; }
        if constexpr (MODE == MODE_DECODE) {
            if (addc) {
                if constexpr (ETYPE == CVEC_F32) {
                    const float bf = __uint_as_float((uint32_t)sa.base);
                    x[0] = __float_as_uint(__uint_as_float(x[0]) + bf);
                    x[1] = __float_as_uint(__uint_as_float(x[1]) + bf);
                } else { x[0] += (uint32_t)sa.base; x[1] += (uint32_t)sa.base; }
            }
            typedef uint32_t __attribute__((__vector_size__(8))) u32x2;
            __builtin_amdgcn_raw_buffer_store_b64(u32x2{x[0], x[1]}, orsrc, ooff + 32u * k, 0, OutPolicy<MODE, NPRED>::AUX);
        } else {
            const bool m0 = pred_match<ETYPE, NPRED>(pa, x[0]), m1 = pred_match<ETYPE, NPRED>(pa, x[1]);
            cnt += (m0 ? 1u : 0u) + (m1 ? 1u : 0u);
            if constexpr (mode_has_masks(MODE)) {
                const uint32_t bits = (m0 ? 1u : 0u) | (m1 ? 2u : 0u);
                if (k < 16) acc_lo |= bits << (2u * k); else acc_hi |= bits << (2u * k - 32u);
            }
        }
        p += m ? (n4 * pc + 20u) >> 3 : 1u;
    }
    uint32_t status = 0;
    if (p > end || hmax > 7) status = walk_status<4>(data, p0, end, kind == K_RAW);
    uint32_t count = FIXED_LEN;
    if constexpr (MODE != MODE_DECODE) {
        if constexpr (mode_has_masks(MODE)) quad_mask_bytes(acc_lo, acc_hi, q, mask_lo, mask_hi);
        cnt += __builtin_amdgcn_update_dpp(0u, cnt, 0xb1 /* quad_perm:[1,0,3,2] */, 0xf, 0xf, true);
        cnt += __builtin_amdgcn_update_dpp(0u, cnt, 0x4e /* quad_perm:[2,3,0,1] */, 0xf, 0xf, true);
        count = cnt;
    }
    sec_result = status ? res_pack(0, status) : res_pack(count, 0);
}

} // namespace cvec
