// cvec_decode.hip — gfx950 kernels for the read side: section index, fused decode + sink /
// decode + filter, and the per-vector finalize.
//
// Data flow of one operation over a batch (all on one stream):
//   k_index     one lane per vector walks its section headers (the chain of
//               FixedSectIterator::next, reference src/section.rs:864-876) and writes one
//               descriptor per 256-element section slot.
//   k_sections  one workgroup per tile of TILE consecutive section slots: the tile's bytes are one
//               contiguous span of the blob, staged once into LDS with 16-byte coalesced loads;
//               one lane per section walks the 32 octet headers (the chain inside
//               decode_to_sink, src/section.rs:548-553 / :638-641 / :708-711) and leaves a
//               32-entry octet table in LDS; then every wave decodes whole sections, 4 elements
//               per lane, undoes XOR with a wavefront scan and either compares (filter) or stores.
//   k_finalize  one lane per vector folds its sections' results in order, so a failing section
//               ends that vector exactly where the reference's iteration would stop.
#include <stdlib.h>

#include <vector>

#include "cvec_device.hpp"

namespace cvec {

// ------------------------------------------------------------------------------------------------
// sec_base: number of 256-element sections each vector's header announces.
// ------------------------------------------------------------------------------------------------
__global__ void k_header_sections(const uint8_t *__restrict__ bytes, const uint64_t *__restrict__ byte_off,
                                  uint64_t n, uint64_t *__restrict__ nsec) {
    uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n) return;
    uint64_t start = byte_off[v], len = byte_off[v + 1] - start;
    uint64_t c = 0;
    if (len >= HEADER_BYTES) {
        uint32_t ne = *(const u32u *)(bytes + start + 8); // FixedSectStats.num_elements, src/vector.rs:165
        c = ((uint64_t)ne + FIXED_LEN - 1) / FIXED_LEN;
    }
    nsec[v] = c;
}

// Exclusive scan of u64 in three steps (block sums, scan of block sums, add).  SCAN_BLOCK elements
// per workgroup; the middle step is a single workgroup looping over the block sums.
constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_BLOCK = SCAN_THREADS * SCAN_ITEMS;

__device__ inline uint64_t wave_incl_scan_u64(uint64_t x) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint64_t y = __shfl_up(x, d, 64);
        if (lane >= d) x += y;
    }
    return x;
}

// Inclusive scan across a 256-thread workgroup; returns this thread's inclusive value and the
// workgroup total.
__device__ inline uint64_t block_incl_scan_u64(uint64_t x, uint64_t *total, uint64_t *s_wave /*[4]*/) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint64_t incl = wave_incl_scan_u64(x);
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    uint64_t add = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < SCAN_THREADS / 64; w++) {
        uint64_t s = s_wave[w];
        if (w < wave) add += s;
        tot += s;
    }
    __syncthreads();
    *total = tot;
    return incl + add;
}

__global__ __launch_bounds__(SCAN_THREADS) void k_scan_block_sums(const uint64_t *__restrict__ in, uint64_t n,
                                                                  uint64_t *__restrict__ block_sums) {
    __shared__ uint64_t s_wave[4];
    uint64_t base = (uint64_t)blockIdx.x * SCAN_BLOCK + (uint64_t)threadIdx.x * SCAN_ITEMS;
    uint64_t sum = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++)
        if (base + i < n) sum += in[base + i];
    uint64_t total;
    block_incl_scan_u64(sum, &total, s_wave);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}

__global__ __launch_bounds__(SCAN_THREADS) void k_scan_of_sums(uint64_t *__restrict__ block_sums, uint64_t nblocks,
                                                               uint64_t *__restrict__ grand_total) {
    __shared__ uint64_t s_wave[4];
    uint64_t carry = 0;
    for (uint64_t b0 = 0; b0 < nblocks; b0 += SCAN_THREADS) {
        uint64_t i = b0 + threadIdx.x;
        uint64_t x = i < nblocks ? block_sums[i] : 0;
        uint64_t total;
        uint64_t incl = block_incl_scan_u64(x, &total, s_wave);
        if (i < nblocks) block_sums[i] = carry + incl - x; // exclusive
        carry += total;
    }
    if (threadIdx.x == 0) *grand_total = carry;
}

// out[i] = exclusive prefix of in[0..i); out[n] = grand total.  `in` and `out` may alias.
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_apply(const uint64_t *in, uint64_t n,
                                                             const uint64_t *__restrict__ block_excl,
                                                             const uint64_t *__restrict__ grand_total, uint64_t *out) {
    __shared__ uint64_t s_wave[4];
    uint64_t base = (uint64_t)blockIdx.x * SCAN_BLOCK + (uint64_t)threadIdx.x * SCAN_ITEMS;
    uint64_t x[SCAN_ITEMS];
    uint64_t sum = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) {
        x[i] = base + i < n ? in[base + i] : 0;
        sum += x[i];
    }
    uint64_t total;
    uint64_t incl = block_incl_scan_u64(sum, &total, s_wave);
    uint64_t run = block_excl[blockIdx.x] + incl - sum;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) {
        if (base + i < n) out[base + i] = run;
        run += x[i];
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == SCAN_THREADS - 1) out[n] = *grand_total;
}

// ------------------------------------------------------------------------------------------------
// k_index: VectorReader::try_new (src/vector.rs:404-414) + the FixedSectIterator chain.
// ------------------------------------------------------------------------------------------------
template <int ETYPE>
__global__ __launch_bounds__(256) void k_index(const uint8_t *__restrict__ bytes,
                                               const uint64_t *__restrict__ byte_off,
                                               const uint64_t *__restrict__ sec_base, uint64_t n,
                                               uint64_t *__restrict__ desc, int32_t *__restrict__ vstatus,
                                               uint64_t *__restrict__ result) {
    constexpr int W = ETYPE == CVEC_U64 ? 8 : 4;
    uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v < CVEC_RESULT_WORDS) result[v] = 0; // k_finalize accumulates into it two launches later
    if (v >= n) return;
    const uint64_t start = byte_off[v], L = byte_off[v + 1] - start;
    const uint64_t slot0 = sec_base[v], nslots = sec_base[v + 1] - slot0;
    const uint8_t *p = bytes + start;
    int st = CVEC_OK;
    // scroll pread error mapping of try_new, then the reader's own checks
    if (L == 0) st = CVEC_BAD_OFFSET;
    else if (L < 4) st = CVEC_NOT_ENOUGH_SPACE;
    else if (L <= 5) st = CVEC_BAD_OFFSET;
    else {
        uint64_t from_header = *(const u32u *)p;
        uint32_t subtype = p[5];
        if (L < from_header + 4) st = CVEC_INPUT_TOO_SHORT;
        else if (subtype != (uint32_t)ETYPE) st = CVEC_WRONG_VECTOR_TYPE;
        else if (L < HEADER_BYTES) st = CVEC_INPUT_TOO_SHORT;
    }
    uint64_t k = 0, pos = HEADER_BYTES;
    if (st == CVEC_OK) {
        while (pos < L) { // sections are walked to the end of the slice (src/vector.rs:443)
            const uint64_t rem = L - pos;
            const uint32_t b0 = p[pos];
            uint64_t nb = 0;
            int e = CVEC_OK;
            if (b0 == K_NULL) {
                nb = 1;
            } else if (b0 == K_CONST) {
                if (rem < 1 + (uint64_t)W) e = CVEC_INPUT_TOO_SHORT;
                nb = 1 + W;
            } else if (b0 == K_NP || b0 == K_DELTA || b0 == K_XOR) {
                if (rem <= 1) e = CVEC_BAD_OFFSET;
                else if (rem < 3) e = CVEC_NOT_ENOUGH_SPACE;
                else {
                    uint32_t n16 = *(const u16u *)(p + pos + 1);
                    nb = b0 == K_NP ? (uint64_t)n16 + 3 : (b0 == K_DELTA ? (uint64_t)n16 + DELTA_HDR : n16);
                    if (nb > rem) e = CVEC_SCROLL_ERR;
                    else if (nb < 3) e = CVEC_INVALID_FORMAT;
                }
            } else {
                e = CVEC_INVALID_SECTION_TYPE;
            }
            if (e) { st = e; break; }
            if (k >= nslots) { st = CVEC_ERR_SECTION_COUNT; break; }
            const bool legal = ETYPE == CVEC_F32 ? (b0 == K_NULL || b0 == K_CONST || b0 == K_XOR) : (b0 != K_XOR);
            desc[slot0 + k] = desc_pack(start + pos, (uint32_t)nb, legal ? b0 : K_ILLEGAL);
            pos += nb;
            k++;
        }
        if (st == CVEC_OK && k != nslots) st = CVEC_ERR_SECTION_COUNT;
    }
    const uint64_t fill_off = start + (pos < L ? pos : L);
    for (; k < nslots; k++) desc[slot0 + k] = desc_pack(fill_off, 0, K_SKIP);
    vstatus[v] = st;
}

// ------------------------------------------------------------------------------------------------
// k_sections
// ------------------------------------------------------------------------------------------------
constexpr int TILE = 16;            // section slots per chunk (= walker lanes in use)
// Octet table: one 36-dword row per section (rows are 16-byte aligned and 36 is bank-conflict free
// for the walker's per-lane rows).  Entries 0..31: byte address of the octet inside the staging slot
// (16 bits) | nonzero mask (8) | nibble word (8).  Entries 32..35: per-section err, kind, start, length.
constexpr int OCT_STRIDE = 36;
constexpr int ROW_ERR = 32, ROW_KIND = 33, ROW_REL = 34, ROW_LEN = 35;
// Staging slot: 18 KiB = 18 LDS-DMA wave instructions of 1 KiB; holds 16 worst-case u32/f32
// sections (1100 B) plus the vector headers between them.  Four slots + two tables stay under
// 80 KiB, so two workgroups share a CU's 160 KiB.
constexpr int SLOT_PIECES = 18;
constexpr uint32_t SLOT_BYTES = SLOT_PIECES * 1024;
constexpr int N_SLOTS = 4;
static_assert(SLOT_PIECES == 18, "the loader's s_waitcnt vmcnt(18) counts SLOT_PIECES instructions");

enum { MODE_FILTER = 0, MODE_DECODE = 1, MODE_FILTER_MASKS = 2 };

constexpr int SEC_THREADS = 512; // wave 0 walks, wave 1 plans, wave 7 loads, every wave but 7 decodes
constexpr int LOADER_WAVE = SEC_THREADS / 64 - 1;

// A chunk = up to TILE consecutive section slots whose bytes fit one staging slot as one span.
struct ChunkParams {
    uint64_t s0;      // first section slot
    uint64_t gbase;   // 16-byte aligned blob offset of the span
    uint32_t n;       // sections in the chunk (0 = no such chunk)
    uint32_t nchunks; // 16-byte pieces of the span
};

struct SecShared {
    __attribute__((aligned(16))) uint8_t data[N_SLOTS][SLOT_BYTES];
    __attribute__((aligned(16))) uint32_t octab[2][TILE * OCT_STRIDE];
    __attribute__((aligned(16))) uint4 meta[8][TILE]; // {0, kind, start inside the slot, length} per section
    ChunkParams P[8];
    uint32_t ticket[2];
};

// One LDS-DMA wave instruction: lane l copies 16 bytes from its own global address to LDS byte
// address lds_dst + 16*l (lds_dst wave-uniform).  Nothing lands in registers and the compiler does
// not count it: completion is waited for with an explicit s_waitcnt vmcnt.  M0 (the LDS base of the
// instruction) is saved and restored inside the statement.
__device__ inline void lds_dma16(const void *gsrc, uint32_t lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
}

// Misaligned LDS accesses take a slow path on gfx950 (~16x the cycles of an aligned one), so every
// LDS read here is a naturally aligned dword pair / triple that is then funnel-shifted.
__device__ inline uint32_t lds_u16_at(const uint8_t *data, uint32_t p) {
    const uint32_t *q = (const uint32_t *)(data + (p & ~3u));
    return __builtin_amdgcn_alignbit(q[1], q[0], (p & 3u) << 3) & 0xffffu;
}

// Inclusive XOR scan over the 32 lanes of each half wave (lane = half*32 + octet), DPP only:
// shifts by 1, 2, 4, 8 inside a 16-lane row, then lane 15 of rows 0 and 2 into rows 1 and 3.
__device__ inline uint32_t xor_scan_half(uint32_t x) {
    x ^= __builtin_amdgcn_update_dpp(0u, x, 0x111 /* row_shr:1 */, 0xf, 0xf, true);
    x ^= __builtin_amdgcn_update_dpp(0u, x, 0x112 /* row_shr:2 */, 0xf, 0xf, true);
    x ^= __builtin_amdgcn_update_dpp(0u, x, 0x114 /* row_shr:4 */, 0xf, 0xf, true);
    x ^= __builtin_amdgcn_update_dpp(0u, x, 0x118 /* row_shr:8 */, 0xf, 0xf, true);
    x ^= __builtin_amdgcn_update_dpp(0u, x, 0x142 /* row_bcast:15 */, 0xa, 0xf, false);
    return x;
}

template <int ETYPE, bool SINGLE>
__device__ inline bool pred_match(const PredArgs &pa, typename Elem<(ETYPE == CVEC_U64 ? 8 : 4)>::U x) {
    if constexpr (SINGLE) {
        if constexpr (ETYPE == CVEC_F32) return __uint_as_float((uint32_t)x) == __uint_as_float((uint32_t)pa.v[0]);
        else if constexpr (ETYPE == CVEC_U32) return (uint32_t)x == (uint32_t)pa.v[0];
        else return x == pa.v[0];
    } else {
        bool m = false;
        if constexpr (ETYPE == CVEC_F32) {
            const float f = __uint_as_float((uint32_t)x);
            for (uint32_t p = 0; p < pa.count; p++) m |= (f == __uint_as_float((uint32_t)pa.v[p])); // IEEE ==
        } else if constexpr (ETYPE == CVEC_U32) {
            for (uint32_t p = 0; p < pa.count; p++) m |= ((uint32_t)x == (uint32_t)pa.v[p]);
        } else {
            for (uint32_t p = 0; p < pa.count; p++) m |= (x == pa.v[p]);
        }
        return m;
    }
}

// Decodes two sections of a chunk with one wave: lane = half*32 + o owns octet o (8 consecutive
// elements) of section sec0 + half.  The octet's header fields are unpacked once per lane; the 8
// fields of its bitstream are then pulled with one aligned dword-pair read and one funnel shift
// each (plain bitstream rule of the format, reference src/nibblepacking.rs:436-470,
// src/nibblepack_simd.rs:571-585), run through the implied XorSink / Delta add and the fused
// epilogue.  Section kinds are handled by selects, so the two halves may hold different kinds.
template <int ETYPE, int MODE, bool SINGLE>
__device__ inline void decode_pair(const uint8_t *data, const uint32_t *octab, uint32_t sec0, uint32_t nsec,
                                   uint64_t slot0, int lane, const PredArgs &pa, const SinkArgs &sa,
                                   uint32_t *__restrict__ sec_res, uint8_t *__restrict__ masks,
                                   void *__restrict__ out_) {
    constexpr int W = ETYPE == CVEC_U64 ? 8 : 4;
    using U = typename Elem<W>::U;
    const uint32_t half = (uint32_t)lane >> 5, o = lane & 31;
    const bool live = sec0 + half < nsec;
    const uint32_t sec = live ? sec0 + half : sec0;
    const uint32_t *row = octab + sec * OCT_STRIDE;
    const uint4 meta = *(const uint4 *)(row + ROW_ERR);
    uint32_t d = row[o];
    const uint32_t err = meta.x, kind = meta.y, rel = meta.z;
    const bool isoct = (kind == K_NP || kind == K_DELTA || kind == K_XOR) && err == 0;
    const bool isconst = kind == K_CONST, isdelta = kind == K_DELTA;
    d = isoct ? d : 0u; // mask 0: every element of the lane is zero, reads stay at the buffer start

    // Const value (section byte 1) or Delta base (section byte 4): aligned dwords + funnel shift
    U v;
    {
        const uint32_t a = (isconst || isdelta) ? rel + (isconst ? 1u : 4u) : 0u;
        const uint32_t *q = (const uint32_t *)(data + (a & ~3u));
        const uint32_t s = (a & 3u) << 3;
        if constexpr (W == 4) v = __builtin_amdgcn_alignbit(q[1], q[0], s);
        else v = (uint64_t)__builtin_amdgcn_alignbit(q[1], q[0], s) |
                 ((uint64_t)__builtin_amdgcn_alignbit(q[2], q[1], s) << 32);
    }

    // ---- the lane's octet ---------------------------------------------------------------------------
    const uint32_t mask = (d >> 16) & 0xffu, hn = d >> 28;
    const uint32_t n4 = (hn << 2) + 4u;          // field width in bits
    const uint32_t t4 = (d >> 22) & 0x3cu;       // trailing zero bits
    uint32_t pb = ((d & 0xffffu) << 3) + 16u;    // bit position of the next stored field
    U x[8];
    if constexpr (W == 4) {
        // a left shift of 32 or more clears the lane, as the AVX2 variable shift of the reference does
        const uint32_t fm = t4 >= 32u ? 0u : 0xffffffffu >> (32u - n4);
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const uint32_t sm = (uint32_t)((int32_t)(mask << (31 - k)) >> 31); // all ones iff element k is stored
            const uint32_t *q = (const uint32_t *)(data + ((pb >> 3) & ~3u));
            const uint32_t f = __builtin_amdgcn_alignbit(q[1], q[0], pb);       // uses pb & 31
            x[k] = ((f & fm) << (t4 & 31u)) & sm;
            pb += n4 & sm;
        }
    } else {
        const uint64_t fm = ~0ull >> (64u - n4);
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const uint32_t sm = (uint32_t)((int32_t)(mask << (31 - k)) >> 31);
            const uint32_t *q = (const uint32_t *)(data + ((pb >> 3) & ~3u));
            const uint32_t d0 = q[0], d1 = q[1], d2 = q[2];
            const uint64_t f = (uint64_t)__builtin_amdgcn_alignbit(d1, d0, pb) |
                               ((uint64_t)__builtin_amdgcn_alignbit(d2, d1, pb) << 32);
            x[k] = ((f & fm) << t4) & (uint64_t)(int64_t)(int32_t)sm;
            pb += n4 & sm;
        }
    }
    if constexpr (ETYPE == CVEC_F32) {
        // XorSink: out[i] = out[i-8] ^ x[i], state zero at the section start (src/sink.rs:271-275,
        // src/section.rs:707): per element index k an inclusive scan over the section's 32 octets.
#pragma unroll
        for (int k = 0; k < 8; k++) x[k] = xor_scan_half((uint32_t)x[k]);
    } else {
        // AddConstSink(base) around the octet decode of a Delta section (src/section.rs:636-641), wrapping
        const U base = isdelta ? v : (U)0;
#pragma unroll
        for (int k = 0; k < 8; k++) x[k] += base;
    }
#pragma unroll
    for (int k = 0; k < 8; k++) x[k] = isconst ? v : x[k];

    // ---- per-section status ----------------------------------------------------------------------------
    uint32_t status = err;
    if (kind == K_ILLEGAL) status = CVEC_INVALID_FORMAT;
    if constexpr (W == 4) {
        // a stored octet of a 32-bit section claiming more than 8 nibbles (src/nibblepack_simd.rs:541):
        // every octet before it fitted, so it is the section's first error
        const unsigned long long wide = __ballot(mask != 0 && hn > 7);
        if (((uint32_t)(half ? wide >> 32 : wide)) != 0 && status == 0) status = CVEC_INVALID_FORMAT;
    }
    const bool skip = kind == K_SKIP;
    const bool zeroes = kind == K_NULL; // the section reaches the sink through process_zeroes()
    const bool ok = live && !skip && status == 0;
    uint32_t count;
    if constexpr (MODE == MODE_FILTER || MODE == MODE_FILTER_MASKS) {
        uint32_t c_lo = 0, c_hi = 0, mbyte = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const bool m = pred_match<ETYPE, SINGLE>(pa, x[k]);
            const unsigned long long b = __ballot(m);
            c_lo += __popc((uint32_t)b);
            c_hi += __popc((uint32_t)(b >> 32));
            if constexpr (MODE == MODE_FILTER_MASKS) mbyte |= m ? (1u << k) : 0u;
        }
        count = half ? c_hi : c_lo;
        if (zeroes) { // null_mask(): all ones iff the predicate matches zero (src/filter.rs:104-106)
            count = pa.match_zero ? FIXED_LEN : 0;
            mbyte = pa.match_zero ? 0xffu : 0u;
        }
        if constexpr (MODE == MODE_FILTER_MASKS) {
            if (ok) masks[(slot0 + half) * 32 + o] = (uint8_t)mbyte;
        }
    } else {
        if (sa.add_const) {
            // AddConstSink in front of VecSink (src/sink.rs:304-317): process_zeroes forwards the
            // base itself; everything else is x + base (f32: IEEE add, ints: wrapping).
            if constexpr (ETYPE == CVEC_F32) {
                const float bf = __uint_as_float((uint32_t)sa.base);
#pragma unroll
                for (int k = 0; k < 8; k++)
                    x[k] = zeroes ? (uint32_t)sa.base : __float_as_uint(__uint_as_float(x[k]) + bf);
            } else {
#pragma unroll
                for (int k = 0; k < 8; k++) x[k] += (U)sa.base;
            }
        }
        if (ok) {
            // 8 consecutive elements per lane, lanes consecutive: full-line stores
            U *op = (U *)out_ + (slot0 + half) * FIXED_LEN + 8 * o;
            if constexpr (W == 4) {
                ((uint4 *)op)[0] = make_uint4(x[0], x[1], x[2], x[3]);
                ((uint4 *)op)[1] = make_uint4(x[4], x[5], x[6], x[7]);
            } else {
#pragma unroll
                for (int k = 0; k < 4; k++)
                    ((uint4 *)op)[k] = make_uint4((uint32_t)x[2 * k], (uint32_t)(x[2 * k] >> 32),
                                                  (uint32_t)x[2 * k + 1], (uint32_t)(x[2 * k + 1] >> 32));
            }
        }
        count = FIXED_LEN;
    }
    if (live && o == 0) sec_res[slot0 + half] = skip ? res_pack(0, 0) : (status ? res_pack(0, status) : res_pack(count, 0));
}

// The octet walk of one section by one lane: 32 dependent steps.  A step's critical chain is one
// LDS round trip (two byte reads, always aligned) and eight VALU operations ending in the next
// cursor; the reads of the next octet are issued as soon as that cursor is known, and the table
// entry of the current octet is packed and stored in the shadow of their latency.  Branch-free; a
// lone wave issues one VALU operation per ~4 cycles, so every instruction in the chain is paid 32
// times.  The loop only notices a cursor that ends past the section; the exact status of the first
// failing octet is worked out afterwards.  After an error the cursor keeps moving over garbage
// (reads stay inside the workgroup's LDS allocation or return 0); the error decides the section's
// result.  An octet of a 32-bit section claiming more than 8 nibbles is left to the decoder, which
// sees every octet header anyway (the walk itself does not depend on it).
template <int W>
__device__ __forceinline__ void walk_section(const uint8_t *data, uint32_t *row) {
    const uint32_t kind = row[ROW_KIND];
    if (!(kind == K_NP || kind == K_DELTA || kind == K_XOR)) return;
    const uint32_t rel = row[ROW_REL];
    const uint32_t end = rel + row[ROW_LEN];
    uint32_t p = rel + (kind == K_DELTA ? DELTA_HDR : 3);
    // same cursor as a raw LDS byte address, advanced beside p (not behind it)
    uint32_t pa = (uint32_t)(uintptr_t)data + p;
    // Two separate byte reads (hipcc would merge plain C byte loads into one 16-bit read, which at
    // an odd address takes the slow misaligned-LDS path).
    uint32_t mask, nw;
    asm volatile("ds_read_u8 %0, %2\n\tds_read_u8 %1, %2 offset:1\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(mask), "=&v"(nw)
                 : "v"(pa)
                 : "memory");
#pragma unroll 8
    for (int k = 0; k < FIXED_LEN / 8; k++) {
        // octet bytes: 1 if mask == 0, else 2 + (n*cnt + 1)/2  ==  (hn*cnt + cnt + 2 + 3*[mask != 0]) >> 1
        const uint32_t hn = nw >> 4; // n - 1
        const uint32_t cnt = __popc(mask);
        const uint32_t a = (uint32_t)__umul24(min(mask, 1u), 3u) + (cnt + 2u);
        const uint32_t size = ((uint32_t)__umul24(hn, cnt) + a) >> 1;
        pa += size;
        // Reads of the next octet (after the last one: a harmless read of the byte that follows).
        // mask and nw pass through the statement so that the packing below is scheduled behind it.
        uint32_t mask_n, nw_n;
        asm volatile("ds_read_u8 %0, %4\n\tds_read_u8 %1, %4 offset:1"
                     : "=&v"(mask_n), "=&v"(nw_n), "+v"(mask), "+v"(nw)
                     : "v"(pa)
                     : "memory");
        row[k] = p | (mask << 16) | (nw << 24); // nw is don't-care when mask == 0
        p += size;
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(mask_n), "+v"(nw_n) : : "memory");
        mask = mask_n;
        nw = nw_n;
    }
    // Common case: the cursor ended inside the section.
    uint32_t err = 0;
    if (p > end) {
        // exact status of the first failing octet, in the order of the reference's checks
        // (src/nibblepack_simd.rs:515-557): empty, length < 3, n > 8, payload too long
        for (int k = 0; k < FIXED_LEN / 8 && !err; k++) {
            const uint32_t e = row[k];
            const uint32_t pk = e & 0xffffu, mk = (e >> 16) & 0xffu, hn = e >> 28;
            const int rem = (int)end - (int)pk;
            const int size = mk ? (int)(2 + (((hn + 1) * __popc(mk) + 1) >> 1)) : 1;
            if (rem <= 0) err = CVEC_NOT_ENOUGH_SPACE;
            else if (mk) {
                if (W == 4 && rem < 3) err = CVEC_NOT_ENOUGH_SPACE;
                else if (W == 4 && hn > 7) err = CVEC_INVALID_FORMAT;
                else if (size > rem || rem < 2) err = CVEC_NOT_ENOUGH_SPACE;
            }
        }
    }
    row[ROW_ERR] = err;
}

// k_sections: persistent workgroups, each owning a contiguous range of section slots, run a
// five-stage software pipeline over chunks of that range with ONE barrier per step.  In step t:
//     wave 1     plans chunk t (extent from the descriptors)
//     wave 0     walks the octets of chunk t-3
//     wave 7     issues the LDS-DMA loads of chunk t-1 into the slot chunk t-5 vacated and waits until
//                chunk t-2 has landed; it does nothing else, so a full memory queue stalls only it
//     all but 7  decode chunk t-4, drawing section pairs from a ticket counter (waves 0 and 1
//                join when their own job is done)
// A chunk's bytes therefore have two whole steps to arrive, the blob streams at a steady rate
// whatever the compute waves do, and the serial octet walk of one chunk runs beside the decode of
// another.
template <int ETYPE, int MODE, bool SINGLE>
__global__ __launch_bounds__(SEC_THREADS, 4) void k_sections(const uint8_t *__restrict__ bytes,
                                                          const uint64_t *__restrict__ desc, uint64_t n_sections,
                                                          uint64_t sections_per_wg, PredArgs pa, SinkArgs sa,
                                                          uint32_t *__restrict__ sec_res,
                                                          uint8_t *__restrict__ masks, void *__restrict__ out_) {
    constexpr int W = ETYPE == CVEC_U64 ? 8 : 4;
    __shared__ SecShared sm;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint64_t sA = (uint64_t)blockIdx.x * sections_per_wg;
    const uint64_t sB = sA + sections_per_wg < n_sections ? sA + sections_per_wg : n_sections;
    if (sA >= sB) return;

    // planning state, meaningful in wave 1 only (lanes 0..TILE-1 hold one descriptor each)
    auto load_desc = [&](uint64_t s) -> uint64_t {
        return (lane < TILE && s + (uint64_t)lane < sB) ? desc[s + lane] : desc_pack(0, 0, K_SKIP);
    };
    auto clip_len = [](uint64_t d) -> uint32_t {
        const uint32_t len = desc_len(d);
        return len < SectLimits<W>::MAX_BYTES ? len : SectLimits<W>::MAX_BYTES;
    };
    uint64_t d_cur = desc_pack(0, 0, K_SKIP), s_cur = sA;
    // Two workgroups share a CU and wave w of a workgroup runs on SIMD w % 4: the latency-critical
    // walkers of co-resident workgroups (block ids 256 apart in dispatch order) go to different SIMDs.
    const int walker_wave = ((blockIdx.x >> 8) & 1) * 2, planner_wave = walker_wave + 1;
    if (wave == planner_wave) d_cur = load_desc(sA);

    for (uint32_t t = 0;; t++) {
        if (wave == planner_wave) {
            // ---- plan chunk t: longest run of sections from s_cur whose span fits a slot ---------------
            const uint64_t left = s_cur < sB ? sB - s_cur : 0;
            const int avail = (int)(left < (uint64_t)TILE ? left : (uint64_t)TILE);
            const uint64_t off = desc_off(d_cur);
            const uint64_t gb = __shfl(off, 0, 64) & ~15ull;
            const uint32_t len = clip_len(d_cur);
            const uint64_t endl = off + len - gb; // wraps to a huge value if off < gb
            const bool fits = lane >= avail || endl <= (uint64_t)SLOT_BYTES;
            const unsigned long long m = __ballot(fits);
            int n = m == ~0ull ? avail : __builtin_ctzll(~m);
            n = n < avail ? n : avail;
            uint32_t e = lane < n ? (uint32_t)endl : 0u;
#pragma unroll
            for (int d = 8; d >= 1; d >>= 1) { uint32_t o = __shfl_xor(e, d, 64); e = o > e ? o : e; }
            if (lane == 0) {
                ChunkParams cp;
                cp.s0 = s_cur; cp.gbase = gb; cp.n = (uint32_t)n; cp.nchunks = (e + 15) >> 4;
                sm.P[t & 7] = cp;
                sm.ticket[(t + 1) & 1] = 0; // counter of chunk t-3, decoded in the next step
            }
            if (lane < n) sm.meta[t & 7][lane] = make_uint4(0u, desc_kind(d_cur), (uint32_t)(off - gb), len);
            // descriptors of chunk t+1 start where chunk t ends
            s_cur += (uint64_t)n;
            d_cur = load_desc(s_cur);
        } else if (wave == walker_wave) {
            // ---- walk chunk t-3 -----------------------------------------------------------------------
            if (t >= 3 && sa._pad != 1) {
                const uint32_t nw = sm.P[(t - 3) & 7].n;
                if ((uint32_t)lane < nw) {
                    uint32_t *row = sm.octab[(t - 3) & 1] + lane * OCT_STRIDE;
                    *(uint4 *)(row + ROW_ERR) = sm.meta[(t - 3) & 7][lane];
                    __builtin_amdgcn_s_setprio(3); // the other waves will wait on this chain at the barrier
                    walk_section<W>(sm.data[(t - 3) & 3], row);
                    __builtin_amdgcn_s_setprio(0);
                }
            }
        } else if (wave == LOADER_WAVE) {
            // ---- LDS-DMA loads of chunk t-1; afterwards everything but those has landed ------------------
            uint32_t nch = 0;
            ChunkParams cp;
            if (t >= 1) {
                cp = sm.P[(t - 1) & 7];
                nch = __builtin_amdgcn_readfirstlane(cp.n) ? __builtin_amdgcn_readfirstlane(cp.nchunks) : 0u;
            }
            if (nch && sa._pad != 3) {
                const uint8_t *src = bytes + cp.gbase;
                const uint32_t dst = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)&sm.data[(t - 1) & 3][0]);
#pragma unroll
                for (int i = 0; i < SLOT_PIECES; i++) {
                    const uint32_t c = (uint32_t)lane + 64u * i;
                    // always SLOT_PIECES instructions (the wait below counts them); lanes past the
                    // span re-read piece 0
                    lds_dma16(src + 16ull * (c < nch ? c : 0u), dst + 1024u * i);
                }
                asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        }
        // ---- decode chunk t-4: section pairs by ticket, the next ticket fetched while a pair decodes ------
        if (wave != LOADER_WAVE && t >= 4 && sa._pad == 0) {
            const ChunkParams cp = sm.P[(t - 4) & 7];
            const uint32_t nd = __builtin_amdgcn_readfirstlane(cp.n);
            const uint32_t npairs = (nd + 1) >> 1;
            uint32_t *tk = &sm.ticket[(t - 4) & 1];
            uint32_t nxt = 0;
            if (lane == 0) nxt = atomicAdd(tk, 1u);
            nxt = __builtin_amdgcn_readfirstlane(nxt);
            while (nxt < npairs) {
                const uint32_t cur = nxt;
                if (lane == 0) nxt = atomicAdd(tk, 1u);
                decode_pair<ETYPE, MODE, SINGLE>(sm.data[(t - 4) & 3], sm.octab[(t - 4) & 1], 2 * cur, nd,
                                                 cp.s0 + 2 * cur, lane, pa, sa, sec_res, masks, out_);
                nxt = __builtin_amdgcn_readfirstlane(nxt);
            }
        }
        __syncthreads();
        if (t >= 4 && sm.P[(t - 4) & 7].n == 0) break; // chunk t-4 does not exist: nothing is in flight any more
    }
}

// ------------------------------------------------------------------------------------------------
// k_finalize: fold the per-section results of each vector in order.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_finalize(const uint64_t *__restrict__ sec_base, uint64_t n,
                                                  const uint32_t *__restrict__ sec_res,
                                                  const int32_t *__restrict__ vstatus,
                                                  uint64_t *__restrict__ per_vec, int32_t *__restrict__ status_out,
                                                  uint64_t *__restrict__ result) {
    __shared__ uint64_t s_cnt[4];
    __shared__ uint32_t s_nerr[4], s_max[4];
    uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t cnt = 0;
    uint32_t st = 0;
    if (v < n) {
        const uint64_t a = sec_base[v], b = sec_base[v + 1];
        for (uint64_t s = a; s < b; s++) {
            const uint32_t r = sec_res[s];
            if (r >> 16) { st = r >> 16; break; } // the reference's iteration stops at the first failing section
            cnt += r & 0xffffu;
        }
        if (!st) st = (uint32_t)vstatus[v];
        if (per_vec) per_vec[v] = cnt;
        if (status_out) status_out[v] = (int32_t)st;
    }
    uint32_t nerr = st ? 1 : 0, mx = st;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        cnt += __shfl_down(cnt, d, 64);
        nerr += __shfl_down(nerr, d, 64);
        uint32_t o = __shfl_down(mx, d, 64);
        mx = o > mx ? o : mx;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { s_cnt[wave] = cnt; s_nerr[wave] = nerr; s_max[wave] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint64_t c = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
        uint32_t e = s_nerr[0] + s_nerr[1] + s_nerr[2] + s_nerr[3];
        uint32_t m = max(max(s_max[0], s_max[1]), max(s_max[2], s_max[3]));
        if (c) atomicAdd((unsigned long long *)&result[0], (unsigned long long)c);
        if (e) {
            atomicAdd((unsigned long long *)&result[1], (unsigned long long)e);
            atomicMax((unsigned long long *)&result[2], (unsigned long long)m);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------

// Optional per-kernel timing of the read path: when armed, every launch_read records a pair of HIP
// events around each of its three kernels on the launch stream.
struct KernelTimer {
    std::vector<hipEvent_t> ev; // 6 per record: index start/stop, sections start/stop, finalize start/stop
    int capacity = 0, used = 0;
    bool armed = false;
};
static KernelTimer g_timer;

hipError_t timing_begin(int max_records) {
    for (hipEvent_t e : g_timer.ev) (void)hipEventDestroy(e);
    g_timer.ev.clear();
    g_timer.capacity = max_records;
    g_timer.used = 0;
    g_timer.ev.resize((size_t)max_records * 6);
    for (auto &e : g_timer.ev) {
        hipError_t r = hipEventCreate(&e);
        if (r != hipSuccess) return r;
    }
    g_timer.armed = true;
    return hipSuccess;
}

// Sums the elapsed milliseconds of each kernel over the recorded launches (synchronises the events).
hipError_t timing_end(float ms[3], int *n_records) {
    g_timer.armed = false;
    ms[0] = ms[1] = ms[2] = 0.f;
    for (int r = 0; r < g_timer.used; r++) {
        for (int k = 0; k < 3; k++) {
            hipEvent_t a = g_timer.ev[(size_t)r * 6 + 2 * k], b = g_timer.ev[(size_t)r * 6 + 2 * k + 1];
            hipError_t e = hipEventSynchronize(b);
            if (e != hipSuccess) return e;
            float t = 0.f;
            e = hipEventElapsedTime(&t, a, b);
            if (e != hipSuccess) return e;
            ms[k] += t;
        }
    }
    *n_records = g_timer.used;
    for (hipEvent_t e : g_timer.ev) (void)hipEventDestroy(e);
    g_timer.ev.clear();
    g_timer.capacity = g_timer.used = 0;
    return hipSuccess;
}

static inline void timer_mark(int rec, int which, hipStream_t stream) {
    if (rec >= 0) (void)hipEventRecord(g_timer.ev[(size_t)rec * 6 + which], stream);
}
hipError_t launch_sec_base(const uint8_t *d_bytes, const uint64_t *d_byte_off, uint64_t n, uint64_t *d_sec_base,
                           uint64_t *d_scratch /* ceil(n/SCAN_BLOCK)+1 u64 */, hipStream_t stream) {
    if (n == 0) return hipMemsetAsync(d_sec_base, 0, sizeof(uint64_t), stream);
    const uint32_t nblk = (uint32_t)((n + SCAN_BLOCK - 1) / SCAN_BLOCK);
    k_header_sections<<<(uint32_t)((n + 255) / 256), 256, 0, stream>>>(d_bytes, d_byte_off, n, d_sec_base);
    k_scan_block_sums<<<nblk, SCAN_THREADS, 0, stream>>>(d_sec_base, n, d_scratch);
    k_scan_of_sums<<<1, SCAN_THREADS, 0, stream>>>(d_scratch, nblk, d_scratch + nblk);
    k_scan_apply<<<nblk, SCAN_THREADS, 0, stream>>>(d_sec_base, n, d_scratch, d_scratch + nblk, d_sec_base);
    return hipGetLastError();
}

uint64_t scan_scratch_words(uint64_t n) { return (n + SCAN_BLOCK - 1) / SCAN_BLOCK + 2; }

hipError_t launch_exclusive_scan(const uint64_t *d_in, uint64_t n, uint64_t *d_out /* n+1 */, uint64_t *d_scratch,
                                 hipStream_t stream) {
    if (n == 0) return hipMemsetAsync(d_out, 0, sizeof(uint64_t), stream);
    const uint32_t nblk = (uint32_t)((n + SCAN_BLOCK - 1) / SCAN_BLOCK);
    k_scan_block_sums<<<nblk, SCAN_THREADS, 0, stream>>>(d_in, n, d_scratch);
    k_scan_of_sums<<<1, SCAN_THREADS, 0, stream>>>(d_scratch, nblk, d_scratch + nblk);
    k_scan_apply<<<nblk, SCAN_THREADS, 0, stream>>>(d_in, n, d_scratch, d_scratch + nblk, d_out);
    return hipGetLastError();
}

template <int ETYPE>
static void launch_read_t(int mode, const cvec_dbatch_view &b, const PredArgs &pa, const SinkArgs &sa,
                          uint64_t *desc, uint32_t *sec_res, int32_t *vstatus, uint64_t *d_result,
                          uint64_t *d_per_vector, int32_t *d_status, uint8_t *d_masks, void *d_out,
                          hipStream_t stream) {
    const uint64_t n = b.n_vectors, ns = b.n_sections;
    const int rec = (g_timer.armed && g_timer.used < g_timer.capacity) ? g_timer.used++ : -1;
    timer_mark(rec, 0, stream);
    k_index<ETYPE><<<(uint32_t)((n > 0 ? n + 255 : 256) / 256), 256, 0, stream>>>(b.d_bytes, b.d_byte_off, b.d_sec_base,
                                                                                 n, desc, vstatus, d_result);
    timer_mark(rec, 1, stream);
    timer_mark(rec, 2, stream);
    if (ns > 0) {
        // Persistent workgroups: two fit a CU (LDS), so 2 x 256 x 2 rounds; each owns a contiguous
        // range of section slots (a multiple of TILE) and pipelines over it.
        static const int rounds = [] { const char *e = getenv("CVEC_WG_ROUNDS"); return e ? atoi(e) : 2; }();
        uint64_t per = (ns + 512 * rounds - 1) / (512 * rounds);
        per = (per + TILE - 1) / TILE * TILE;
        if (per < 8 * TILE) per = 8 * TILE;
        const uint32_t nwg = (uint32_t)((ns + per - 1) / per);
        const bool single = pa.count == 1;
        static const int ablate = [] { const char *e = getenv("CVEC_ABLATE"); return e ? atoi(e) : 0; }();
        SinkArgs sa2 = sa;
        sa2._pad = (uint32_t)ablate; // timing experiments only: 1 = stage only, 2 = stage + walk
#define CVEC_LAUNCH_SEC(M, S)                                                                                       \
    k_sections<ETYPE, M, S><<<nwg, SEC_THREADS, 0, stream>>>(b.d_bytes, desc, ns, per, pa, sa2, sec_res, d_masks, d_out)
        if (mode == MODE_DECODE) CVEC_LAUNCH_SEC(MODE_DECODE, true);
        else if (d_masks) { if (single) CVEC_LAUNCH_SEC(MODE_FILTER_MASKS, true); else CVEC_LAUNCH_SEC(MODE_FILTER_MASKS, false); }
        else { if (single) CVEC_LAUNCH_SEC(MODE_FILTER, true); else CVEC_LAUNCH_SEC(MODE_FILTER, false); }
#undef CVEC_LAUNCH_SEC
    }
    timer_mark(rec, 3, stream);
    timer_mark(rec, 4, stream);
    if (n > 0)
        k_finalize<<<(uint32_t)((n + 255) / 256), 256, 0, stream>>>(b.d_sec_base, n, sec_res, vstatus, d_per_vector,
                                                                    d_status, d_result);
    timer_mark(rec, 5, stream);
}

hipError_t launch_read(int mode, const cvec_dbatch_view &b, const PredArgs &pa, const SinkArgs &sa, uint64_t *desc,
                       uint32_t *sec_res, int32_t *vstatus, uint64_t *d_result, uint64_t *d_per_vector,
                       int32_t *d_status, uint8_t *d_masks, void *d_out, hipStream_t stream) {
    switch (b.etype) {
    case CVEC_U32:
        launch_read_t<CVEC_U32>(mode, b, pa, sa, desc, sec_res, vstatus, d_result, d_per_vector, d_status, d_masks,
                                d_out, stream);
        break;
    case CVEC_U64:
        launch_read_t<CVEC_U64>(mode, b, pa, sa, desc, sec_res, vstatus, d_result, d_per_vector, d_status, d_masks,
                                d_out, stream);
        break;
    case CVEC_F32:
        launch_read_t<CVEC_F32>(mode, b, pa, sa, desc, sec_res, vstatus, d_result, d_per_vector, d_status, d_masks,
                                d_out, stream);
        break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

} // namespace cvec
