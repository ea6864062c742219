// cvec_decode.hip — gfx950 kernels for the read side: section index, fused decode + sink /
// decode + filter, and the per-vector finalize.
//
// Data flow of one operation over a batch (all on one stream):
//   k_index     one lane per vector walks its section headers (the chain of
//               FixedSectIterator::next, reference src/section.rs:864-876) and writes one
//               descriptor per 256-element section slot.
//   k_sections  persistent, wave-autonomous: every wave draws ranges of section slots from counters
//               in the workspace, owns an 18 KiB LDS region, streams 16 sections at a time into it (bounds-checked
//               16-byte buffer loads -> registers -> LDS, the next chunk prefetched into registers
//               while the current one is processed) and runs the 32 octet steps of all 16 sections
//               at once: a quad of lanes per section walks the octet headers (the chain inside
//               decode_to_sink, src/section.rs:548-553 / :638-641 / :708-711) and each lane pulls
//               its two elements of every octet straight out of the bitstream, keeps the XOR state
//               in registers and either compares (filter) or stores.  No barriers, no octet table.
//   k_finalize  one lane per vector folds its sections' results in order, so a failing section
//               ends that vector exactly where the reference's iteration would stop; a pass that
//               wants the total alone and met no failure takes the section kernel's sums instead.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <type_traits>
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
        c = header_slots(ne, len);
    }
    nsec[v] = c;
}

// Exclusive scan of u64 in two steps (block sums; add): SCAN_BLOCK elements per workgroup, and every workgroup of
// the second step adds up the sums of the blocks in front of it itself (a few hundred values at most: a batch of a
// million sections has 512 blocks), which saves the launch of a kernel that scans the block sums.  Arrays of one
// block take the second step alone.
constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_BLOCK = SCAN_THREADS * SCAN_ITEMS;
constexpr uint32_t SCAN_SELF_BLOCKS = 2048; // up to here a block of the second step adds up its predecessors' sums itself

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

// (arrays of more than SCAN_SELF_BLOCKS blocks: the block sums are scanned by one workgroup in between, as every
// block adding up its predecessors would be quadratic)
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
// `block_sums` (the first step's totals, null for a single block): the block's offset is the sum of those in
// front of it.
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_apply(const uint64_t *in, uint64_t n,
                                                             const uint64_t *__restrict__ block_sums, int scanned,
                                                             uint64_t *out) {
    __shared__ uint64_t s_wave[4];
    uint64_t before = 0;
    if (block_sums != nullptr && scanned) {
        before = block_sums[blockIdx.x];
    } else if (block_sums != nullptr) {
        uint64_t part = 0;
        for (uint32_t b = threadIdx.x; b < blockIdx.x; b += SCAN_THREADS) part += block_sums[b];
        block_incl_scan_u64(part, &before, s_wave);
    }
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
    uint64_t run = before + incl - sum;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) {
        if (base + i < n) out[base + i] = run;
        run += x[i];
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == SCAN_THREADS - 1) out[n] = before + total;
}

// ------------------------------------------------------------------------------------------------
// k_index: VectorReader::try_new (src/vector.rs:404-414) + the FixedSectIterator chain.
// ------------------------------------------------------------------------------------------------
template <int ETYPE>
__global__ __launch_bounds__(256) void k_index(const uint8_t *__restrict__ bytes,
                                               const uint64_t *__restrict__ byte_off,
                                               const uint64_t *__restrict__ sec_base, uint64_t n,
                                               uint64_t padded_bytes, const uint8_t *__restrict__ gate,
                                               uint64_t *__restrict__ desc, int32_t *__restrict__ vstatus,
                                               uint64_t *__restrict__ result, uint64_t *__restrict__ tag,
                                               uint64_t cookie, uint32_t *__restrict__ work, uint32_t epoch,
                                               uint8_t *__restrict__ tail) {
    constexpr int W = ETYPE == CVEC_U64 ? 8 : 4;
    // One 4-byte load per section header (type byte + 16-bit length), pulled back inside the padded
    // allocation at the very end of the blob: the chain of dependent loads is what this kernel costs.
    // A plain load, not a non-temporal one: the header lines (one 128-byte line per section, an eighth of
    // a blob of 1 KiB sections) then stay in the 256 MiB last-level cache, where the section kernel finds
    // them again (section kernel 0.195 -> 0.191 ms per GiB, this kernel 0.030 -> 0.032 ms).
    auto load4 = [&](uint64_t a) -> uint32_t {
        const uint64_t a4 = a + 4 <= padded_bytes ? a : padded_bytes - 4;
        return *(const u32u *)(bytes + a4) >> (8 * (uint32_t)(a - a4));
    };
    uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v < CVEC_RESULT_WORDS) result[v] = 0; // k_finalize accumulates into it two launches later
    // Whose index this workspace holds from now on (a gated index is nobody's): a later pass that reuses
    // the descriptors instead of walking the headers again checks it.
    if (v == 0) { tag[0] = gate ? 0ull : cookie; tag[1] = 0; }
    // the section kernel's range counters and, on the same lines, its partial totals
    if (v < WORK_COUNTERS) { work[v * WORK_STRIDE] = 0; work[v * WORK_STRIDE + 2] = 0; work[v * WORK_STRIDE + 3] = 0; }
    if (v >= n) return;
    const uint64_t start = byte_off[v], L = byte_off[v + 1] - start;
    const uint64_t slot0 = sec_base[v], nslots = sec_base[v + 1] - slot0;
    int st = CVEC_OK;
    // scroll pread error mapping of try_new, then the reader's own checks
    if (L == 0) st = CVEC_BAD_OFFSET;
    else if (L < 4) st = CVEC_NOT_ENOUGH_SPACE;
    else if (L <= 5) st = CVEC_BAD_OFFSET;
    else {
        const uint64_t a8 = start + 8 <= padded_bytes ? start : padded_bytes - 8; // L >= 6: padded_bytes >= 16
        const uint64_t h8 = *(const u64u *)(bytes + a8) >> (8 * (uint32_t)(start - a8));
        uint64_t from_header = (uint32_t)h8;
        uint32_t subtype = (uint32_t)(h8 >> 40) & 0xffu;
        if (L < from_header + 4) st = CVEC_INPUT_TOO_SHORT;
        else if (subtype != (uint32_t)ETYPE) st = CVEC_WRONG_VECTOR_TYPE;
        else if (L < HEADER_BYTES) st = CVEC_INPUT_TOO_SHORT;
    }
    uint64_t k = 0, pos = HEADER_BYTES;
    if (st == CVEC_OK) {
        // Sections are walked to the end of the slice (src/vector.rs:443), eight at a time: their
        // descriptors collect in registers and are stored after the eighth.  A store between two header
        // loads would make every load wait for that store's round trip as well (loads and stores in
        // flight cannot be told apart by the wait counter).
        bool walking = true;
        while (walking && pos < L) {
            uint64_t found[8];
            uint8_t ftail[8]; // bytes between the section's end and the vector's end (255: that many or more)
            int nfound = 0;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                found[j] = 0;
                ftail[j] = 255;
                if (walking && pos < L) {
                    const uint64_t rem = L - pos;
                    const uint32_t hw = load4(start + pos);
                    const uint32_t b0 = hw & 0xffu;
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
                            uint32_t n16 = (hw >> 8) & 0xffffu;
                            nb = b0 == K_NP ? (uint64_t)n16 + 3 : (b0 == K_DELTA ? (uint64_t)n16 + DELTA_HDR : n16);
                            if (nb > rem) e = CVEC_SCROLL_ERR;
                            else if (nb < 3) e = CVEC_INVALID_FORMAT;
                        }
                    } else {
                        e = CVEC_INVALID_SECTION_TYPE;
                    }
                    if (e == CVEC_OK && k >= nslots) e = CVEC_ERR_SECTION_COUNT;
                    if (e) {
                        st = e;
                        walking = false;
                    } else {
                        const bool legal = ETYPE == CVEC_F32 ? (b0 == K_NULL || b0 == K_CONST || b0 == K_XOR) : (b0 != K_XOR);
                        // A section whose gate mask (the 256-bit mask of the same slot of an earlier column)
                        // is empty is only parsed, like MultiVectorFilter's advance() (src/filter.rs:204-209,
                        // :279-282).
                        bool off = false;
                        if (gate) {
                            const uint4 *g = (const uint4 *)(gate + 32 * (slot0 + k));
                            const uint4 a = g[0], b = g[1];
                            off = ((a.x | a.y | a.z | a.w) | (b.x | b.y | b.z | b.w)) == 0;
                        }
                        found[j] = off ? desc_pack(start + pos, 0, K_GATED)
                                       : desc_pack(start + pos, (uint32_t)nb, legal ? b0 : K_ILLEGAL);
                        ftail[j] = (uint8_t)(rem - nb < 255 ? rem - nb : 255);
                        nfound = j + 1;
                        pos += nb;
                        k++;
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < 8; j++)
                if (j < nfound) { desc[slot0 + k - nfound + j] = found[j]; tail[slot0 + k - nfound + j] = ftail[j]; }
        }
        if (st == CVEC_OK && k != nslots) st = CVEC_ERR_SECTION_COUNT;
    }
    const uint64_t fill_off = start + (pos < L ? pos : L);
    for (; k < nslots; k++) { desc[slot0 + k] = desc_pack(fill_off, 0, K_SKIP); tail[slot0 + k] = 255; }
    vstatus[v] = st;
    if (st != CVEC_OK) work[FAIL_FLAG_WORD] = epoch; // a vector of this pass failed: the finalize pass has to fold per vector
}

} // namespace cvec

#include "cvec_sections.hpp"

namespace cvec {

// largest value over the wave
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t x) {
    x = max(x, (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0xb1 /* quad_perm:[1,0,3,2] */, 0xf, 0xf, false));
    x = max(x, (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x4e /* quad_perm:[2,3,0,1] */, 0xf, 0xf, false));
    x = max(x, (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x141 /* row_half_mirror */, 0xf, 0xf, false));
    x = max(x, (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x140 /* row_mirror */, 0xf, 0xf, false));
    const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)x, 0), b = (uint32_t)__builtin_amdgcn_readlane((int)x, 16),
                   c = (uint32_t)__builtin_amdgcn_readlane((int)x, 32), d = (uint32_t)__builtin_amdgcn_readlane((int)x, 48);
    return max(max(a, b), max(c, d));
}

// Plan of one chunk: the longest run of sections from s whose span fits a staging region.
struct ChunkPlan {
    uint64_t gb;   // 128-byte aligned blob offset of the span
    uint32_t n;    // sections in the chunk
    uint32_t nch;  // 16-byte pieces of the span
    uint32_t rel;  // the lane's section: start inside the region,
    uint32_t len;  //   clipped length,
    uint32_t kind; //   kind (K_SKIP beyond the chunk)
    int32_t tail_from; // reference tail rule only: region offset from which an octet starts within 33 bytes of the vector's end
};

#ifdef CVEC_PROBES
// probe build: per-wave wall-clock stamps (100 MHz) of k_sections (start, end), read back by cvec_debug_sections_times
__device__ uint64_t g_sec_stamps[8192 * 2];
hipError_t debug_sections_times(uint64_t *h_out) { return hipMemcpyFromSymbol(h_out, HIP_SYMBOL(g_sec_stamps), sizeof(g_sec_stamps)); }
#define SEC_STAMP(i) do { const uint32_t gw_ = blockIdx.x * SEC_WAVES + wave; if (lane == 0 && gw_ < 8192) g_sec_stamps[gw_ * 2 + (i)] = wall_clock64(); } while (0)
#else
#define SEC_STAMP(i) do { } while (0)
#endif

// k_sections: every wave is on its own.  A wave works through ranges of section slots (its first one fixed,
// the others drawn from the workspace's counters) with an 18, 12 or 6 KiB staging region in LDS, and loops
// over chunks of up to 16 sections.  Per chunk: the bytes
// prefetched into registers during the previous chunk's compute are written to the region; the next
// chunk is planned from its descriptors and its global loads are issued into those registers (plus
// the descriptor loads of the chunk after it); then the 16 sections are walked and decoded at once
// (process_chunk) while the loads fly.  Registers, not a second LDS region, double-buffer the
// stream: LDS capacity bounds the resident waves (8 per CU), the register file has room to spare.
// No barriers; the only traffic between waves are the range counters.
// TAILREF: the reference crate's tail decoder instead of the format's rule for the last octets of a vector (see
// process_chunk); `tail` = the index pass's per-section distances to the vector's end.
// DSHAPE (decode passes): 0 = stagings of up to 16 sections, 8-byte stores for 32-bit elements; 1 = wide stagings:
// up to 64 sections, worked off in up to four rounds of 16, so the bytes of the next staging have up to four octet
// loops' time to arrive behind the output stores that fill the CU's memory pipeline (one descriptor per lane instead
// of one per quad); 2 = stagings of 16 with the 16-byte stores of the wide shape (process_chunk's PAIRS).
template <int ETYPE, int MODE, int NPRED, int PIECES, int TAILREF = 0, int DSHAPE = 0>
__global__ __launch_bounds__(SEC_THREADS) __attribute__((amdgpu_waves_per_eu(sec_wg_per_cu(PIECES), sec_wg_per_cu(PIECES)))) void k_sections(const uint8_t *__restrict__ bytes,
                                                          const uint64_t *__restrict__ desc, uint64_t n_sections,
                                                          uint64_t sections_per_wave, PredArgs pa, SinkArgs sa,
                                                          uint32_t *__restrict__ sec_res,
                                                          uint8_t *__restrict__ masks, void *__restrict__ out_,
                                                          const uint64_t *__restrict__ tag, uint64_t cookie,
                                                          const uint8_t *__restrict__ gate, uint32_t *__restrict__ work,
                                                          uint32_t n_groups, uint32_t epoch, uint32_t keep_cached,
                                                          const uint8_t *__restrict__ tail) {
    constexpr int W = ETYPE == CVEC_U64 ? 8 : 4;
    __shared__ SecShared<PIECES> sm;
    // Descriptors reused from an earlier pass: they are only followed if the workspace says that they are
    // this batch's (k_finalize reports the mismatch); a stale table could point anywhere.
    if (tag != nullptr && *tag != cookie) return;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // The slots are cut into ranges of sections_per_wave.  Wave g starts with range g; with range counters
    // (`work`, zero at the start of the kernel) every further range goes to whichever wave asks first.  The
    // issue arbiter of a SIMD prefers its older wave, so with equal shares the waves of the workgroup that
    // reached a CU first finish a third earlier than those of the second one, which then run on alone at
    // less than half the throughput of a pair (measured: 127 against 177 us); handed out on demand, the
    // ranges keep every SIMD paired up until the batch is done.  Atomics on one address complete at about
    // 7 ns apiece, so workgroup b shares counter b % n_groups (its own 128-byte line) with a few dozen waves
    // only: the k-th range that counter c hands out is n_static + k * n_groups + c.
    const uint64_t sA = ((uint64_t)blockIdx.x * SEC_WAVES + wave) * sections_per_wave;
    if (sA >= n_sections) return;
    uint64_t sB = sA + sections_per_wave < n_sections ? sA + sections_per_wave : n_sections; // end of the range being planned
    const uint32_t n_static = gridDim.x * SEC_WAVES; // ranges that have an owner from the start
    SEC_STAMP(0);
    uint8_t *data = sm.data[wave];
    const uint32_t lds_base = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)data);
    const uint32_t sec = (uint32_t)lane >> 2;
    static_assert(DSHAPE == 0 || MODE == MODE_DECODE, "wide stagings and paired stores are decode shapes");
    constexpr bool WIDE = DSHAPE == 1;
    // the section of a staging whose descriptor and plan the lane holds: its quad's, or (WIDE) its own
    const uint32_t pl = WIDE ? (uint32_t)lane : sec;
    constexpr uint32_t STAGE_SECS = WIDE ? 64u : (uint32_t)TILE;

    // `gate` (a pass over a resident index only; otherwise the index pass has applied it already): the
    // 256-bit mask of the same slot of an earlier column.  Lane q of the quad looks at its quarter; a
    // section whose gate mask is empty becomes a K_GATED descriptor, as k_index would have made it.
    // Only the general filter modes take a gate (the launcher sends gated passes there): the count-only
    // bit-pattern mode and decode keep the pointer out of their loops.
    const uint8_t *const gate_ = (MODE == MODE_FILTER || mode_has_masks(MODE)) ? gate : nullptr;
    auto load_desc = [&](uint64_t s) -> uint64_t {
        uint64_t d = s + pl < sB ? desc[s + pl] : desc_pack(0, 0, K_SKIP);
        if (gate_ != nullptr) {
            uint32_t any = 0;
            if (s + sec < sB) {
                const uint2 g = *(const uint2 *)(gate_ + 32 * (s + sec) + 8 * ((uint32_t)lane & 3u));
                any = g.x | g.y;
            }
            any |= __builtin_amdgcn_update_dpp(0u, any, 0xb1 /* quad_perm:[1,0,3,2] */, 0xf, 0xf, true);
            any |= __builtin_amdgcn_update_dpp(0u, any, 0x4e /* quad_perm:[2,3,0,1] */, 0xf, 0xf, true);
            if (any == 0 && desc_kind(d) != K_SKIP) d = desc_pack(desc_off(d), 0, K_GATED);
        }
        return d;
    };
    auto plan = [&](uint64_t s, uint64_t d) -> ChunkPlan {
        ChunkPlan c;
        const uint64_t left = s < sB ? sB - s : 0;
        const uint32_t avail = (uint32_t)(left < (uint64_t)STAGE_SECS ? left : (uint64_t)STAGE_SECS);
        const uint64_t off = desc_off(d);
        // gated sections need no bytes: the span runs from the first to the last section that does
        const bool wants = pl < avail && desc_kind(d) != K_GATED;
        const unsigned long long mw = __ballot(wants);
        const int f = mw ? (int)__builtin_ctzll(mw) : 0;
        const uint64_t off0 = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(off >> 32), f) << 32) |
                              (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)off, f);
        // the span starts on a 128-byte line: every 1 KiB load instruction then covers eight whole lines (a
        // span that starts inside a line makes each instruction touch nine, and the memory system delivers
        // 6.2 instead of 7 TB/s)
        c.gb = off0 & ~127ull;
        uint32_t len = desc_len(d);
        c.len = len < SectLimits<W>::MAX_BYTES ? len : SectLimits<W>::MAX_BYTES;
        const uint64_t endl = off + c.len - c.gb; // wraps to a huge value if off < gb
        const bool fits = !wants || endl <= (uint64_t)region_span(PIECES);
        const unsigned long long m = __ballot(fits);
        uint32_t n = m == ~0ull ? avail : (uint32_t)__builtin_ctzll(~m) >> (WIDE ? 0 : 2);
        c.n = n < avail ? n : avail;
        // whole rounds of 16, the rest starts the next staging; a last round that is at least half full is kept
        if (WIDE && c.n > (uint32_t)TILE && (c.n & ((uint32_t)TILE - 1u)) < (uint32_t)TILE / 2) c.n &= ~((uint32_t)TILE - 1u);
        // section offsets do not decrease with the slot number, so the span ends with the last of the
        // chunk's sections that wants bytes
        constexpr uint32_t LANES_PER_SEC = WIDE ? 1u : 4u;
        const unsigned long long mn = c.n * LANES_PER_SEC >= 64u ? mw : mw & ((1ull << (LANES_PER_SEC * c.n)) - 1ull);
        const uint32_t e = mn ? __builtin_amdgcn_readlane((uint32_t)endl, 63 - (int)__builtin_clzll(mn)) : 0u;
        c.nch = (e + 15) >> 4;
        c.rel = (uint32_t)(off - c.gb);
        c.kind = pl < c.n ? desc_kind(d) : K_SKIP;
        c.tail_from = 0x7fffffff;
        if constexpr (TAILREF != 0) {
            // (a load per chunk, issued with the chunk's staging loads and used one loop pass later)
            const uint32_t td = pl < c.n ? tail[s + pl] : 255u;
            if (td < 34u) c.tail_from = (int32_t)(c.rel + c.len + td) - 33;
        }
        return c;
    };

    // Staging registers: piece i of a span = 64 lanes x 16 bytes.  The loads go through a buffer
    // descriptor over exactly the span, so lanes past its end cost no traffic and return zeros;
    // the whole region (zero pad included) is rewritten for every chunk without any per-lane test.
    typedef uint32_t __attribute__((__vector_size__(16))) u32x4;
    u32x4 R[PIECES];
    auto issue_loads = [&](const ChunkPlan &c) {
#ifdef CVEC_PROBES
        const uint32_t span_bytes = (sa._pad == 5 || sa._pad == 7) ? 0u : c.nch * 16u; // probe build: 5 = no staging loads (7: and no output)
#else
        const uint32_t span_bytes = c.nch * 16u;
#endif
        const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(bytes + c.gb), 0, (int)span_bytes, 0x00020000);
        // Cache policy of the span loads (a uniform branch: the policy is an immediate of the instruction).
        // Non-temporal for blobs far beyond the 256 MiB last-level cache: every byte is read once per pass, and
        // without the hint the same loads take 13 % longer (0.217 against 0.191 ms per GiB of incompressible
        // f32).  Default policy for count and mask passes over smaller blobs: a batch that is queried again and
        // again then stays in that cache from one pass to the next, index pass included (dense u32, 0.2 GB: 0.203
        // -> 0.186 ms per pass; 0.6 GB of f32: 0.189 -> 0.187).  Decode passes keep the hint above 256 MiB (see the
        // launcher).
        if (keep_cached) {
#pragma unroll
            for (int i = 0; i < PIECES; i++) R[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane * 16, i * 1024, 0);
        } else {
#pragma unroll
            for (int i = 0; i < PIECES; i++) R[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane * 16, i * 1024, 2);
        }
    };

    uint64_t s_cur = sA, s_next = sA;
    // the wave's next range is asked for one range ahead: the counter's answer has a whole range to arrive
    uint32_t ticket = 0;
    // the counter this wave draws from: one of its XCD's (n_groups / 8 per XCD, so that a counter's line stays in one
    // L2), later one that has ranges left
    const uint32_t xcc = (uint32_t)__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 7u; // HW_REG_XCC_ID[3:0]
    uint32_t group = n_groups >= 8 ? xcc * (n_groups >> 3) + (blockIdx.x >> 3) % (n_groups >> 3) : blockIdx.x % n_groups;
    const uint64_t n_ranges = (n_sections + sections_per_wave - 1) / sections_per_wave;
    const bool dyn = work != nullptr && n_ranges > n_static; // otherwise no range is left to hand out
    const uint32_t n_dyn = dyn ? (uint32_t)(n_ranges - n_static) : 0u;
    // ranges counter g has to hand out: the indices j < n_dyn with j % n_groups == g
    auto quota = [&](uint32_t g) -> uint32_t { return g < n_dyn ? (n_dyn - g + n_groups - 1) / n_groups : 0u; };
    if (dyn && lane == 0) ticket = atomicAdd(work + (size_t)group * WORK_STRIDE, 1u);
    uint64_t d_next = load_desc(sA);
    // A chunk's per-section results are stored one loop pass later, behind the wait for the prefetched
    // bytes: loads and stores share one in-order counter only among their own kind, so a wait for loads
    // while a store is in flight has to drain the store too (a write round trip per chunk otherwise).
    uint32_t res_word = 0, mk_lo = 0, mk_hi = 0; // mask bytes 8q .. 8q + 7 of the lane's section (q = lane & 3)
    uint64_t res_slot = 0;
    // What the wave's sections counted, and whether one of them failed: while no section and no vector header of
    // a pass fails, the pass's total is the sum of these, and the finalize pass need not fold per vector.
    uint64_t acc_cnt = 0;
    uint32_t acc_bad = 0;
    bool res_pending = false, mk_pending = false;
    // A use of the first descriptor ahead of the loop: entering the loop with that load in flight would
    // put a full wait in front of the plan in every pass, right behind the deferred store.
    asm volatile("" ::"v"(d_next));
    // ---- plan the next chunk and start its loads ----------------------------------------------------------------
    // (the loads are unconditional, so that the staging registers never become conditionally assigned: past the
    // wave's range the plan is empty and the loads fetch nothing)
    ChunkPlan nxt;
    uint64_t s_after;
    auto advance = [&]() {
        nxt = plan(s_next, d_next);
        if constexpr (MODE == MODE_FILTER || mode_has_masks(MODE)) {
            // Behind a selective gate most chunks are closed as a whole: their results (count 0, empty
            // masks) are written here and the wave moves on, asking for the descriptors and gate masks
            // of the next four chunks at once instead of paying one round trip per chunk of 16.
            if (gate_ != nullptr) {
                auto closed = [&](uint64_t s, uint64_t d) -> bool {
                    return s + TILE <= sB && __ballot(desc_kind(d) == K_GATED) == ~0ull;
                };
                auto store_closed = [&](uint64_t s) {
                    if ((lane & 3) == 0) sec_res[s + sec] = res_pack(0, 0);
                    if constexpr (mode_has_masks(MODE)) *(uint2 *)(masks + (s + sec) * 32 + 8 * (lane & 3)) = uint2{0u, 0u};
                };
                while (closed(s_next, d_next)) {
                    store_closed(s_next);
                    s_next += TILE;
                    const uint64_t q0 = load_desc(s_next), q1 = load_desc(s_next + TILE), q2 = load_desc(s_next + 2 * TILE),
                                   q3 = load_desc(s_next + 3 * TILE);
                    d_next = q0;
                    if (closed(s_next, q0)) {
                        store_closed(s_next); s_next += TILE; d_next = q1;
                        if (closed(s_next, q1)) {
                            store_closed(s_next); s_next += TILE; d_next = q2;
                            if (closed(s_next, q2)) { store_closed(s_next); s_next += TILE; d_next = q3; }
                        }
                    }
                }
                nxt = plan(s_next, d_next);
            }
        }
        // where the chunk after the next one starts: behind it, or in the wave's next range once this one is used up
        // (ahead of the staging loads: the counter's answer is then the only thing the wait for it can wait for)
        s_after = s_next + nxt.n;
        if (dyn && s_after >= sB && nxt.n != 0) {
            uint32_t t = (uint32_t)__builtin_amdgcn_readfirstlane((int)ticket);
            bool ok = t < quota(group);
            // The wave's counter has run dry: it moves to the counter with most ranges left (one load fetches all
            // of them) and asks there.  Only the last passes of a wave get here, so this wait is not hidden.
            for (int tries = 0; tries < 4 && !ok; tries++) {
                uint32_t key = 0; // ranges left << 8 | counter, the lane's best
                for (uint32_t g = (uint32_t)lane; g < n_groups; g += 64) {
                    const uint32_t v = __hip_atomic_load(work + (size_t)g * WORK_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const uint32_t q = quota(g);
                    const uint32_t left = v < q ? q - v : 0u;
                    const uint32_t k = ((left < 0xffffffu ? left : 0xffffffu) << 8) | g;
                    key = k > key ? k : key;
                }
                const uint32_t best = wave_max_u32(key);
                if ((best >> 8) == 0) break;
                group = best & 255u;
                uint32_t tt = 0;
                if (lane == 0) tt = atomicAdd(work + (size_t)group * WORK_STRIDE, 1u);
                t = (uint32_t)__builtin_amdgcn_readfirstlane((int)tt);
                ok = t < quota(group);
            }
            if (ok) {
                s_after = ((uint64_t)n_static + (uint64_t)t * n_groups + group) * sections_per_wave;
                sB = s_after + sections_per_wave < n_sections ? s_after + sections_per_wave : n_sections;
                if (lane == 0) ticket = atomicAdd(work + (size_t)group * WORK_STRIDE, 1u);
            } else {
                s_after = n_sections;
                sB = n_sections;
            }
        }
        issue_loads(nxt);
        d_next = load_desc(s_after);
    };
    // Everything the wave has in flight (staging loads, next descriptor, counter's answer) passes through here:
    // the compiler puts its waits for them at this point.
    auto landed = [&]() {
#pragma unroll
        for (int i = 0; i < PIECES; i++) asm volatile("" ::"v"(R[i]));
        asm volatile("" ::"v"(d_next), "v"(ticket));
    };
    // Chunk 0 is planned and loaded ahead of the loop, and waited for: every pass of the loop then has a chunk to
    // work on, and the loop is entered with nothing in flight.  (With a first pass that only loads, the loop body
    // has a way around the octet loop on which loads are in flight at its end; merged with the way through the
    // octet loop, whose output stores are in flight there, the compiler has to drain everything at the end of
    // every pass, and the decode pass waits for its own stores once per chunk.)
    advance();
    landed();
    ChunkPlan cur = nxt;
    s_cur = s_next;
    s_next = s_after;
    while (WIDE && cur.n != 0) {
        {
            u32x4 *dst = (u32x4 *)data;
#pragma unroll
            for (int i = 0; i < PIECES; i++) dst[64 * i + lane] = R[i];
        }
        __builtin_amdgcn_sched_barrier(0);
        advance();
        // rounds of 16 sections: the quad takes the plan of its section from the lane that holds it
#pragma nounroll
        for (uint32_t j0 = 0; j0 < cur.n; j0 += TILE) {
            const int src = (int)(j0 + sec);
            const bool live = (uint32_t)src < cur.n;
            // (all lanes take part in the exchange, whatever they do with the result: a lane that is switched off
            // hands out zeros, and in a round that is not full the lanes that hold the plans are not live themselves;
            // the results pass through an empty statement so that the reads stay in front of anything conditional)
            uint32_t kind_x = (uint32_t)__builtin_amdgcn_ds_bpermute(src << 2, (int)cur.kind);
            uint32_t rel_q = (uint32_t)__builtin_amdgcn_ds_bpermute(src << 2, (int)cur.rel);
            uint32_t len_q = (uint32_t)__builtin_amdgcn_ds_bpermute(src << 2, (int)cur.len);
            int32_t tf = 0x7fffffff;
            if constexpr (TAILREF != 0) tf = __builtin_amdgcn_ds_bpermute(src << 2, cur.tail_from);
            asm volatile("" : "+v"(kind_x), "+v"(rel_q), "+v"(len_q), "+v"(tf));
            const uint32_t kind_q = live ? kind_x : (uint32_t)K_SKIP;
            int32_t tail_from = 0x7fffffff;
            if constexpr (TAILREF != 0) {
                if (live && tf != 0x7fffffff) tail_from = (int32_t)lds_base + tf;
            }
            uint32_t ooff;
#ifdef CVEC_PROBES
            const out_rsrc_t orsrc = (sa._pad == 2 || sa._pad == 7) ? chunk_out_rsrc<W, MODE_FILTER>(out_, s_cur + j0 + sec, lane, ooff)
                                                                    : chunk_out_rsrc<W, MODE>(out_, s_cur + j0 + sec, lane, ooff);
#else
            const out_rsrc_t orsrc = chunk_out_rsrc<W, MODE>(out_, s_cur + j0 + sec, lane, ooff);
#endif
            const bool odd = process_chunk<ETYPE, MODE, NPRED, region_span(PIECES), TAILREF, true>(
                data, lds_base, kind_q, rel_q, len_q, live, lane, pa, sa, res_word, mk_lo, mk_hi, orsrc, ooff, tail_from);
            if constexpr (W == 4) {
                const unsigned long long odd_m = __ballot(odd);
                if (odd_m != 0) {
                    if ((odd_m >> ((uint32_t)lane & ~3u)) & 0xfull) section_exact<ETYPE, MODE, NPRED, TAILREF>(data, lds_base, kind_q, rel_q, len_q, lane, pa, sa, res_word,
                                                               mk_lo, mk_hi, orsrc, ooff, tail_from);
                }
            }
            // (stored at once: the waits of this loop are for loads that are older than any store of the staging)
            if (live && (lane & 3) == 0) { sec_res[s_cur + j0 + sec] = res_word; acc_cnt += res_word & 0xffffu; acc_bad |= res_word; }
        }
        landed();
        s_cur = s_next;
        cur = nxt;
        s_next = s_after;
    }
    while (!WIDE && cur.n != 0) {
        // ---- registers -> staging region ---------------------------------------------------------------------
        {
            u32x4 *dst = (u32x4 *)data;
#pragma unroll
            for (int i = 0; i < PIECES; i++) dst[64 * i + lane] = R[i];
        }
        __builtin_amdgcn_sched_barrier(0); // keep the store below the staging writes (and their wait)
        if (res_pending) { sec_res[res_slot] = res_word; acc_cnt += res_word & 0xffffu; acc_bad |= res_word; }
        res_pending = false;
        if constexpr (mode_has_masks(MODE)) {
            // one 8-byte store per lane: the quads of a chunk write 16 x 32 contiguous bytes
            if (mk_pending) *(uint2 *)(masks + res_slot * 32 + 8 * (lane & 3)) = uint2{mk_lo, mk_hi};
            mk_pending = false;
        }
        advance();
        // ---- walk + decode the current chunk ----------------------------------------------------------------------
        const bool has_octets = cur.kind == K_NP || cur.kind == K_DELTA || cur.kind == K_XOR || cur.kind == K_RAW;
        if (MODE != MODE_DECODE && __ballot(has_octets) == 0) {
            // A chunk of Null / Const / gated sections only (sparse vectors, or a column behind a
            // selective one): no octet to walk, every section's count and mask follow from its kind
            // (src/filter.rs:222-223, :104-106; a Const section is 256 copies of one value).
            if (sec < cur.n) {
                bool all = false;
                if (cur.kind == K_NULL) all = pa.match_zero != 0;
                else if (cur.kind == K_CONST) all = pred_match<ETYPE, NPRED>(pa, lds_load_unaligned<W>(lds_base + cur.rel + 1u));
                res_word = cur.kind == K_ILLEGAL ? res_pack(0, CVEC_INVALID_FORMAT) : res_pack(all ? FIXED_LEN : 0, 0);
                res_slot = s_cur + sec;
                res_pending = (lane & 3) == 0;
                mk_lo = mk_hi = all ? ~0u : 0u;
                mk_pending = cur.kind != K_SKIP && cur.kind != K_ILLEGAL;
            }
#ifdef CVEC_PROBES
        } else if (sa._pad != 1) { // probe build: 1 = stage the bytes but skip the octet loop
#else
        } else {
#endif
            uint32_t ooff;
#ifdef CVEC_PROBES
            // probe build: 2 = the decode pass with every store aimed past its descriptor (issued, then dropped)
            const out_rsrc_t orsrc = (sa._pad == 2 || sa._pad == 7) ? chunk_out_rsrc<W, MODE_FILTER>(out_, s_cur + sec, lane, ooff)
                                                  : chunk_out_rsrc<W, MODE>(out_, s_cur + sec, lane, ooff);
#else
            const out_rsrc_t orsrc = chunk_out_rsrc<W, MODE>(out_, s_cur + sec, lane, ooff);
#endif
            // (region offset -> LDS address; sections that end far from the vector's end keep "never")
            const int32_t tail_from = TAILREF != 0 && cur.tail_from != 0x7fffffff ? (int32_t)lds_base + cur.tail_from : 0x7fffffff;
            const bool odd = process_chunk<ETYPE, MODE, NPRED, region_span(PIECES), TAILREF, DSHAPE == 2>(
                data, lds_base, cur.kind, cur.rel, cur.len, sec < cur.n, lane, pa, sa, res_word, mk_lo, mk_hi, orsrc, ooff, tail_from);
            // Decode: the next chunk's bytes are waited for here, behind the chunk's output stores in program order.
            // The counter of memory operations retires in order, so this wait does not include those stores, and
            // the end of the pass is reached with only stores in flight, which the next pass never waits for (its
            // own loads are issued behind them, but they have had a whole chunk's time by then).
            if constexpr (MODE == MODE_DECODE) landed();
            if constexpr (W == 4) {
                const unsigned long long odd_m = __ballot(odd); // a lane only sees the octets that store its own elements
                if (odd_m != 0) {
                    if ((odd_m >> ((uint32_t)lane & ~3u)) & 0xfull) section_exact<ETYPE, MODE, NPRED, TAILREF>(data, lds_base, cur.kind, cur.rel, cur.len, lane, pa, sa, res_word,
                                                               mk_lo, mk_hi, orsrc, ooff, tail_from);
                }
            }
            res_slot = s_cur + sec;
            res_pending = sec < cur.n && (lane & 3) == 0;
            mk_pending = sec < cur.n && cur.kind != K_SKIP && cur.kind != K_ILLEGAL;
        }
        s_cur = s_next;
        cur = nxt;
        s_next = s_after;
    }
    if (res_pending) { sec_res[res_slot] = res_word; acc_cnt += res_word & 0xffffu; acc_bad |= res_word; }
    if constexpr (mode_has_masks(MODE)) {
        if (mk_pending) *(uint2 *)(masks + res_slot * 32 + 8 * (lane & 3)) = uint2{mk_lo, mk_hi};
    }
    if (work != nullptr) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) acc_cnt += __shfl_down(acc_cnt, d, 64);
        const bool bad = __ballot((acc_bad >> 16) != 0) != 0;
        if (lane == 0) {
            if (acc_cnt) atomicAdd((unsigned long long *)(work + (size_t)(blockIdx.x % n_groups) * WORK_STRIDE + 2), (unsigned long long)acc_cnt);
            if (bad) work[FAIL_FLAG_WORD] = epoch;
        }
    }
    SEC_STAMP(1);
}

// ------------------------------------------------------------------------------------------------
// k_finalize: fold the per-section results of each vector in order.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_finalize(const uint64_t *__restrict__ sec_base, uint64_t n,
                                                  const uint32_t *__restrict__ sec_res,
                                                  const int32_t *__restrict__ vstatus,
                                                  uint64_t *__restrict__ per_vec, int32_t *__restrict__ status_out,
                                                  uint64_t *__restrict__ result, const uint64_t *__restrict__ tag,
                                                  uint64_t cookie, const uint32_t *__restrict__ work, uint32_t epoch) {
    __shared__ uint64_t s_cnt[4];
    __shared__ uint32_t s_nerr[4], s_max[4];
    uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t cnt = 0;
    uint32_t st = 0;
    const bool stale = tag != nullptr && *tag != cookie; // the section pass did not run: see k_sections
    // Nobody asks for per-vector results and nothing failed in this pass (no vector header, no section: the flag
    // word would carry the pass's number): the total is the sum of what the section kernel's waves counted, and
    // the number of failed vectors and the largest status stay zero.
    if (work != nullptr && per_vec == nullptr && status_out == nullptr && !stale && work[FAIL_FLAG_WORD] != epoch) {
        if (blockIdx.x != 0) return;
        for (uint32_t g = threadIdx.x; g < (uint32_t)WORK_COUNTERS; g += blockDim.x) cnt += *(const uint64_t *)(work + (size_t)g * WORK_STRIDE + 2);
        v = n; // no vector of its own: straight to the block's reduction
    }
    if (v < n) {
        const uint64_t a = sec_base[v], b = stale ? a : sec_base[v + 1];
        const uint32_t hst = (uint32_t)vstatus[v];
        // eight result words per round trip (the loads of a round do not depend on each other)
        for (uint64_t s = a; s < b && !st; s += 8) {
            uint32_t r[8];
#pragma unroll
            for (int j = 0; j < 8; j++) r[j] = s + j < b ? sec_res[s + j] : 0u;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                if (!st) {
                    if (r[j] >> 16) st = r[j] >> 16; // the reference's iteration stops at the first failing section
                    else cnt += r[j] & 0xffffu;
                }
            }
        }
        if (stale) st = CVEC_ERR_WORKSPACE;
        if (!st) st = hst;
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

// Start of a pass over a resident index (no index pass in front of it): clears the result block that k_finalize
// accumulates into and the section kernel's range counter.
__global__ void k_begin_indexed(uint64_t *__restrict__ result, uint32_t *__restrict__ work) {
    if (threadIdx.x < CVEC_RESULT_WORDS) result[threadIdx.x] = 0;
    for (uint32_t i = threadIdx.x; i < (uint32_t)WORK_COUNTERS; i += blockDim.x) work[i * WORK_STRIDE] = 0;
}

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
constexpr int RANGES_PER_WAVE = 16; // section ranges per wave of a full launch when they are handed out on demand
constexpr int WORK_GROUPS = 128; // range counters in use (of the workspace's WORK_COUNTERS): 16 waves per counter in a full launch

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
    if (nblk > 1) k_scan_block_sums<<<nblk, SCAN_THREADS, 0, stream>>>(d_sec_base, n, d_scratch);
    if (nblk > SCAN_SELF_BLOCKS) k_scan_of_sums<<<1, SCAN_THREADS, 0, stream>>>(d_scratch, nblk, d_scratch + nblk);
    k_scan_apply<<<nblk, SCAN_THREADS, 0, stream>>>(d_sec_base, n, nblk > 1 ? d_scratch : nullptr, nblk > SCAN_SELF_BLOCKS, d_sec_base);
    return hipGetLastError();
}

uint64_t scan_scratch_words(uint64_t n) { return (n + SCAN_BLOCK - 1) / SCAN_BLOCK + 2; }

hipError_t launch_exclusive_scan(const uint64_t *d_in, uint64_t n, uint64_t *d_out /* n+1 */, uint64_t *d_scratch,
                                 hipStream_t stream) {
    if (n == 0) return hipMemsetAsync(d_out, 0, sizeof(uint64_t), stream);
    const uint32_t nblk = (uint32_t)((n + SCAN_BLOCK - 1) / SCAN_BLOCK);
    if (nblk > 1) k_scan_block_sums<<<nblk, SCAN_THREADS, 0, stream>>>(d_in, n, d_scratch);
    if (nblk > SCAN_SELF_BLOCKS) k_scan_of_sums<<<1, SCAN_THREADS, 0, stream>>>(d_scratch, nblk, d_scratch + nblk);
    k_scan_apply<<<nblk, SCAN_THREADS, 0, stream>>>(d_in, n, nblk > 1 ? d_scratch : nullptr, nblk > SCAN_SELF_BLOCKS, d_out);
    return hipGetLastError();
}

// Launches k_sections over a descriptor table (one descriptor per 256-element slot).
template <int ETYPE>
static void launch_sections(int mode, const uint8_t *d_bytes, const uint64_t *desc, uint64_t ns, const PredArgs &pa,
                            const SinkArgs &sa, uint32_t *sec_res, uint8_t *d_masks, void *d_out, const uint64_t *tag,
                            uint64_t cookie, const uint8_t *d_gate, uint64_t n_bytes, uint32_t *work, uint32_t epoch,
                            const uint8_t *tail /* the index pass's tail distances: the reference's tail rule; or null */,
                            hipStream_t stream) {
    if (ns > 0) {
        // Persistent waves: 2 workgroups x 4 waves per CU (LDS-bound), each wave owning a contiguous
        // range of section slots (a multiple of TILE).  Count-only passes over a batch whose average
        // section is small enough for 16 of them to fit a 12 KiB region run 3 workgroups per CU on such
        // regions (n_bytes == 0: size unknown, the large shape).
        // The product build takes no run-time switches.  A probe build (make probes: -DCVEC_PROBES, a separate
        // library for the timing tools) reads experiment settings from the environment.
#ifdef CVEC_PROBES
        static const int rounds = [] { const char *e = getenv("CVEC_WG_ROUNDS"); return e ? atoi(e) : 1; }();
        static const int wgcu_env = [] { const char *e = getenv("CVEC_WG_PER_CU"); return e ? atoi(e) : 0; }();
        static const int shape_env = [] { const char *e = getenv("CVEC_REGION_SHAPE"); return e ? atoi(e) : 0; }(); // 1 = always large, 2 = small where possible
        static const int ablate = [] { const char *e = getenv("CVEC_ABLATE"); return e ? atoi(e) : 0; }();
#else
        constexpr int rounds = 1, wgcu_env = 0, shape_env = 0, ablate = 0;
#endif
        // bit-pattern equality is IEEE == unless the predicate is a float zero or NaN
        const uint32_t fb = (uint32_t)pa.v[0];
        const bool bits_ok = pa.count <= 1 && (ETYPE != CVEC_F32 || ((fb & 0x7fffffffu) != 0 && (fb & 0x7fffffffu) <= 0x7f800000u));
        // the smaller region shapes: count-only passes, and mask passes of the bit-pattern compare (the general mask
        // mode needs more registers than three workgroups per CU leave)
        const bool count_only = mode != MODE_DECODE && (d_masks == nullptr || bits_ok);
        bool small = count_only && n_bytes != 0 && (n_bytes / ns) * TILE + 2 * 16 <= region_span(REGION_PIECES_SMALL);
        // (decode passes were measured on the 12 KiB shape as well: 3 to 6 % slower, like the 6 KiB one)
        // A third shape, 6 KiB regions at four workgroups per CU, for count-only passes over batches whose sections
        // average at most 378 bytes (dense integers, deltas): dense u32 0.202 -> 0.190 ms per GiB, u64 delta 0.171 ->
        // 0.169.  (Decode passes were measured on it too and lose 3 to 6 %: more waves interleave more partial-line
        // stores; they keep the 18 KiB shape.)
        bool tiny = count_only && n_bytes != 0 && (n_bytes / ns) * TILE + 2 * 16 <= region_span(REGION_PIECES_TINY);
        if (shape_env == 1) small = tiny = false;
        if (shape_env == 2) { small = count_only; tiny = false; }
        if (shape_env == 3) tiny = count_only;
        if (shape_env == 4) tiny = false;
        if (tiny) small = false;
        // Decode shapes by average section size (tools/decode_shape_probe.py, sections kernel per GiB of output;
        // stagings of 16 with 8-byte stores / stagings of 16 with 16-byte stores / wide stagings):
        //   u32, 189 B: 0.341 / 0.316 / 0.274 ms     313 B: 0.367 / 0.308 / 0.317     567 B: 0.373 / 0.346 / 0.355
        //        692 B: 0.390 / 0.379 / 0.386        817 B: 0.409 / 0.420 / 0.396    1067 B: 0.451 / 0.482 / 0.444
        //   f32, 317 B: 0.368 / 0.301 / 0.316        588 B: 0.374 / 0.341 / 0.367     845 B: 0.410 / 0.401 / 0.399
        //   u64, 189 B: 0.277 / - / 0.246            316 B: 0.285 / - / 0.276         442 B: 0.291 / - / 0.299
        // 32-bit elements: wide stagings up to 260 B and from 760 B per section (there a staging is one round, and
        // what wins is the loop's form), stagings of 16 with 16-byte stores between; 64-bit elements: wide stagings
        // up to 380 B.
        const uint64_t avg_sec = n_bytes / ns;
        int dshape = 0;
        if (mode == MODE_DECODE && n_bytes != 0) {
            if (ETYPE == CVEC_U64) dshape = avg_sec <= 380 ? 1 : 0;
            else dshape = (avg_sec <= 260 || avg_sec > 760) ? 1 : 2;
        }
        bool wide = dshape == 1;
        if (mode == MODE_DECODE && shape_env >= 5 && shape_env <= 7) { dshape = shape_env - 5; wide = dshape == 1; } // probe build
        const int wgcu = wgcu_env > 0 ? wgcu_env : (tiny ? 4 : (small ? 3 : 2));
        uint64_t nwaves_max = 256ull * wgcu * SEC_WAVES * (uint64_t)(rounds > 0 ? rounds : 1);
#ifdef CVEC_PROBES
        // probe build: a cap on the waves of a launch, so that small test batches run with many ranges per wave
        static const int max_waves_env = [] { const char *e = getenv("CVEC_MAX_WAVES"); return e ? atoi(e) : 0; }();
        if (max_waves_env > 0) nwaves_max = (uint64_t)max_waves_env;
#endif
        // `work` (a counter that is zero when the kernel starts; not for gated passes, whose waves skip ahead in
        // their own ranges): ranges an eighth of a wave's even share, handed out on demand (see k_sections).
        // Without it every wave gets one range of that even share.
        if (d_gate != nullptr) work = nullptr;
        uint64_t per = (ns + nwaves_max - 1) / nwaves_max;
#ifdef CVEC_PROBES
        static const uint32_t n_groups_max = [] { const char *e = getenv("CVEC_WORK_GROUPS"); const int x = e ? atoi(e) : WORK_GROUPS; return (uint32_t)(x >= 1 && x <= WORK_COUNTERS ? x : WORK_GROUPS); }();
#else
        constexpr uint32_t n_groups_max = WORK_GROUPS;
#endif
#ifdef CVEC_PROBES
        static const int rpw = [] { const char *e = getenv("CVEC_RANGES_PER_WAVE"); return e ? atoi(e) : RANGES_PER_WAVE; }();
        if (rpw <= 0) work = nullptr;
#else
        constexpr int rpw = RANGES_PER_WAVE;
#endif
        if (work != nullptr) per = (per + rpw - 1) / rpw;
        per = (per + TILE - 1) / TILE * TILE;
        // Wide stagings (decode over sections of which at least 32 fit the region): ranges are whole stagings of 64
        if (wide) per = (per + 63) / 64 * 64;
#ifdef CVEC_PROBES
        static const int min_chunks = [] { const char *e = getenv("CVEC_RANGE_MIN_CHUNKS"); return e ? atoi(e) : 2; }();
#else
        constexpr int min_chunks = 2;
#endif
        if (per < (uint64_t)min_chunks * TILE) per = (uint64_t)min_chunks * TILE;
        uint64_t nwaves = (ns + per - 1) / per; // ranges
        if (nwaves > nwaves_max) nwaves = nwaves_max;
        const uint32_t nwg = (uint32_t)((nwaves + SEC_WAVES - 1) / SEC_WAVES);
        SinkArgs sa2 = sa;
        sa2._pad = (uint32_t)ablate; // always 0 in the product build
        // blobs up to 640 MiB are worth keeping in the last-level cache between passes (see issue_loads)
        // (decode passes: only blobs up to 256 MiB, the index pass's lines of which the section pass then finds in
        // that cache although a GiB of output passes through it: dense u32 0.286 -> 0.280 ms, u64 delta 0.243 ->
        // 0.236; a 0.6 GB blob of f32 0.339 -> 0.341)
        uint32_t keep_cached = n_bytes != 0 && n_bytes <= ((mode != MODE_DECODE ? 640ull : 256ull) << 20) ? 1u : 0u;
        if (ablate == 6) keep_cached = 1;
        if (ablate == 7) keep_cached = 0;
        // the predicate list padded to its bucket with copies of the first value
        PredArgs pb = pa;
        uint32_t bucket = 1;
        while (bucket < pb.count) bucket *= 2;
        for (uint32_t i = pb.count; i < bucket && i < CVEC_MAX_PRED_VALUES; i++) pb.v[i] = pb.v[0];
#define CVEC_LAUNCH_SEC_PW(M, N, P, T, WD)                                                                          \
    k_sections<ETYPE, M, N, P, T, WD><<<nwg, SEC_THREADS, 0, stream>>>(d_bytes, desc, ns, per, pb, sa2, sec_res, d_masks, d_out, \
                                                                   tag, cookie, d_gate, work, nwg < n_groups_max ? nwg : n_groups_max, epoch, keep_cached, tail)
#define CVEC_LAUNCH_SEC_P(M, N, P, T)                                                                               \
    do {                                                                                                            \
        if (M == MODE_DECODE && dshape == 1) CVEC_LAUNCH_SEC_PW(M, N, P, T, (M == MODE_DECODE ? 1 : 0));           \
        else if (M == MODE_DECODE && dshape == 2) CVEC_LAUNCH_SEC_PW(M, N, P, T, (M == MODE_DECODE && ETYPE != CVEC_U64 ? 2 : 0)); \
        else CVEC_LAUNCH_SEC_PW(M, N, P, T, 0);                                                                     \
    } while (0)
#define CVEC_LAUNCH_SEC_T(M, N, T)                                                                                  \
    do {                                                                                                            \
        if ((M == MODE_FILTER || mode_bit_compare(M)) && tiny)                                                      \
            CVEC_LAUNCH_SEC_P(M, N, ((M == MODE_FILTER || mode_bit_compare(M)) ? REGION_PIECES_TINY : REGION_PIECES), T); \
        else if ((M == MODE_FILTER || mode_bit_compare(M)) && small)                                                \
            CVEC_LAUNCH_SEC_P(M, N, ((M == MODE_FILTER || mode_bit_compare(M)) ? REGION_PIECES_SMALL : REGION_PIECES), T); \
        else CVEC_LAUNCH_SEC_P(M, N, REGION_PIECES, T);                                                             \
    } while (0)
#define CVEC_LAUNCH_SEC(M, N)                                                                                       \
    do {                                                                                                            \
        /* (the conditional constants keep the shapes a mode never takes from being instantiated) */                \
        if (tail != nullptr) CVEC_LAUNCH_SEC_T(M, N, 1); /* the reference's tail rule */                            \
        else CVEC_LAUNCH_SEC_T(M, N, 0);                                                                            \
    } while (0)
#define CVEC_LAUNCH_BUCKET(M)                                                                                       \
    switch (bucket) {                                                                                               \
    case 1: CVEC_LAUNCH_SEC(M, 1); break;                                                                           \
    case 2: CVEC_LAUNCH_SEC(M, 2); break;                                                                           \
    case 4: CVEC_LAUNCH_SEC(M, 4); break;                                                                           \
    case 8: CVEC_LAUNCH_SEC(M, 8); break;                                                                           \
    default: CVEC_LAUNCH_SEC(M, 16); break;                                                                         \
    }
        if (mode == MODE_DECODE) {
            // (the second template argument is the store policy in this mode: 1 = default, 2 = non-temporal.  The
            // non-temporal hint paid for dense u32 while every decode ran stagings of 16 with 8-byte stores (2.25 ->
            // 2.62 TB/s); with the shapes above the default policy is faster everywhere, and only it is built.)
            CVEC_LAUNCH_SEC(MODE_DECODE, 1);
        }
        else if (bucket == 1) {
            if (d_masks) { if (bits_ok) CVEC_LAUNCH_SEC(MODE_FILTER_MASKS_BITS, 1); else CVEC_LAUNCH_SEC(MODE_FILTER_MASKS, 1); }
            else if (bits_ok && !d_gate) CVEC_LAUNCH_SEC(MODE_FILTER_BITS, 1);
            else CVEC_LAUNCH_SEC(MODE_FILTER, 1);
        }
        else if (d_masks) { CVEC_LAUNCH_BUCKET(MODE_FILTER_MASKS); }
        else {
            CVEC_LAUNCH_BUCKET(MODE_FILTER);
        }
#undef CVEC_LAUNCH_BUCKET
#undef CVEC_LAUNCH_SEC
#undef CVEC_LAUNCH_SEC_T
#undef CVEC_LAUNCH_SEC_P
#undef CVEC_LAUNCH_SEC_PW
    }
}

// Decodes a table of bare 64-bit octet blocks (K_RAW descriptors built by the legacy stream index)
// into out[256 * slot ...]; sec_res receives one status word per block.
hipError_t launch_raw_blocks_u64(const uint8_t *d_bytes, const uint64_t *desc, uint64_t n_blocks, uint32_t *sec_res,
                                 uint64_t *d_out, hipStream_t stream) {
    PredArgs pa;
    SinkArgs sa;
    memset(&pa, 0, sizeof(pa));
    memset(&sa, 0, sizeof(sa));
    launch_sections<CVEC_U64>(MODE_DECODE, d_bytes, desc, n_blocks, pa, sa, sec_res, nullptr, d_out, nullptr, 0, nullptr, 0, nullptr, 0, nullptr, stream);
    return hipGetLastError();
}

// phase: READ_ALL = index, sections, finalize; READ_INDEX_ONLY = the index pass alone (descriptors and
// per-vector header statuses stay in the caller's workspace); READ_INDEXED = sections + finalize over
// descriptors that an earlier call left there.
// cvec_vstream.hip
bool vstream_eligible(int mode, const cvec_dbatch_view &b, bool forced);
hipError_t launch_vstream(int mode, const cvec_dbatch_view &b, const PredArgs &pa, const SinkArgs &sa, uint32_t *nvalid,
                          uint64_t *d_result, uint64_t *d_per_vector, int32_t *d_status, uint8_t *d_masks, void *d_out,
                          uint64_t *tag, hipStream_t stream);
// Process-wide defaults (cvec_set_read_path, cvec_set_tail_rule) and the options of the entry point that is running
// on this thread (cvec_*_opt_*: set for the duration of the call; -1 = the default applies).  A read operation is
// launched by the thread that entered the library, so two threads can run with different options at once.
static std::atomic<int> g_read_path{PATH_AUTO};
static std::atomic<int> g_tail_rule{0}; // 0: the format's bitstream rule; 1: the reference crate's tail decoder
static thread_local int t_read_path = -1, t_tail_rule = -1;
void set_read_path(int path) { g_read_path.store(path, std::memory_order_relaxed); }
void set_tail_rule(int rule) { g_tail_rule.store(rule, std::memory_order_relaxed); }
void set_call_options(int path, int tail) { t_read_path = path; t_tail_rule = tail; }
void get_call_options(int *path, int *tail) { *path = t_read_path; *tail = t_tail_rule; }
int effective_read_path() { return t_read_path >= 0 ? t_read_path : g_read_path.load(std::memory_order_relaxed); }
int effective_tail_rule() { return t_tail_rule >= 0 ? t_tail_rule : g_tail_rule.load(std::memory_order_relaxed); }

template <int ETYPE>
static void launch_read_t(int mode, int phase, const cvec_dbatch_view &b, const PredArgs &pa, const SinkArgs &sa,
                          uint64_t *desc, uint32_t *sec_res, int32_t *vstatus, uint64_t *d_result,
                          uint64_t *d_per_vector, int32_t *d_status, uint8_t *d_masks, void *d_out,
                          const uint8_t *d_gate, uint64_t *tag, uint32_t *work, hipStream_t stream) {
    const uint64_t n = b.n_vectors, ns = b.n_sections;
    // identifies the batch whose index the workspace holds: pointers and shape, never zero
    uint64_t cookie = 0x9e3779b97f4a7c15ull;
    for (uint64_t x : {(uint64_t)(uintptr_t)b.d_bytes, (uint64_t)(uintptr_t)b.d_byte_off, b.n_bytes, n, ns, (uint64_t)b.etype})
        cookie = (cookie ^ x) * 0xff51afd7ed558ccdull, cookie ^= cookie >> 29;
    cookie |= 1;
    const uint64_t *check = phase == READ_INDEXED ? tag : nullptr;
    // The number of this pass: kernels that meet a failed vector or section write it into the workspace's flag word,
    // and the finalize pass takes the short way (total = the section kernel's sums) only if the word holds another
    // value.  Stale or uninitialised contents can only send it the long way.  Not with a gate (closed chunks are
    // written off outside the accumulating store, and an indexed gated pass has fixed ranges), not without
    // sections (no section kernel runs), and not over a resident index (the vector headers were checked by an
    // earlier pass).
    static std::atomic<uint32_t> g_epoch{1};
    const uint32_t epoch = g_epoch.fetch_add(1, std::memory_order_relaxed);
    const bool fast_total = phase == READ_ALL && d_gate == nullptr && ns > 0 && n > 0;
    // per-section distance to the vector's end, behind the range counters: written by every index pass, read by the
    // section kernel under the reference's tail rule
    uint8_t *const tail = (uint8_t *)(work + (size_t)(WORK_COUNTERS + 1) * WORK_STRIDE);
    const int rec = (g_timer.armed && g_timer.used < g_timer.capacity) ? g_timer.used++ : -1;
    const int read_path = effective_read_path(), tail_rule = effective_tail_rule();
    // (the reference's tail rule needs the index pass's distances: the section-slot path)
    if (phase == READ_ALL && d_gate == nullptr && read_path != PATH_SLOTS && tail_rule == 0) {
        // sparse batches of many vectors: one kernel, no section index, no per-section results (the timing slots of
        // the index and finalize kernels hold the memset and nothing)
        const bool vs = read_path == PATH_VSTREAM ? vstream_eligible(mode, b, true)
                                                  : (read_path == PATH_AUTO && vstream_eligible(mode, b, false));
        if (vs) {
            timer_mark(rec, 0, stream);
            (void)hipMemsetAsync(d_result, 0, CVEC_RESULT_WORDS * sizeof(uint64_t), stream);
            timer_mark(rec, 1, stream);
            timer_mark(rec, 2, stream);
            (void)launch_vstream(mode, b, pa, sa, (uint32_t *)vstatus, d_result, d_per_vector, d_status, d_masks, d_out, tag, stream);
            timer_mark(rec, 3, stream);
            timer_mark(rec, 4, stream);
            timer_mark(rec, 5, stream);
            return;
        }
    }
    timer_mark(rec, 0, stream);
    // One wave per workgroup: a batch of 26214 long vectors is 410 waves, which 256-thread workgroups would put
    // on 103 of the 256 CUs (f32 of 10000: step 0.236 -> 0.229 ms; the headline's 2048 waves: no difference).
    // At least 256 threads in all: the first ones also clear the result block and the range counters.
    constexpr uint32_t INDEX_THREADS = 64;
    if (phase != READ_INDEXED)
        k_index<ETYPE><<<(uint32_t)(((n > 256 ? n : 256) + INDEX_THREADS - 1) / INDEX_THREADS), INDEX_THREADS, 0, stream>>>(
            b.d_bytes, b.d_byte_off, b.d_sec_base, n, (b.n_bytes + 15) & ~15ull, d_gate, desc, vstatus, d_result, tag,
            cookie, work, epoch, tail);
    else
        k_begin_indexed<<<1, 256, 0, stream>>>(d_result, work); // k_index does this otherwise
    timer_mark(rec, 1, stream);
    timer_mark(rec, 2, stream);
    if (phase != READ_INDEX_ONLY)
        launch_sections<ETYPE>(mode, b.d_bytes, desc, ns, pa, sa, sec_res, d_masks, d_out, check, cookie,
                               phase == READ_INDEXED ? d_gate : nullptr, b.n_bytes, work, epoch,
                               tail_rule != 0 ? tail : nullptr, stream);
    timer_mark(rec, 3, stream);
    timer_mark(rec, 4, stream);
    if (n > 0 && phase != READ_INDEX_ONLY)
        k_finalize<<<(uint32_t)((n + 255) / 256), 256, 0, stream>>>(b.d_sec_base, n, sec_res, vstatus, d_per_vector,
                                                                    d_status, d_result, check, cookie, fast_total ? work : nullptr,
                                                                    epoch);
    timer_mark(rec, 5, stream);
}

hipError_t launch_read(int mode, int phase, const cvec_dbatch_view &b, const PredArgs &pa, const SinkArgs &sa,
                       uint64_t *desc, uint32_t *sec_res, int32_t *vstatus, uint64_t *d_result, uint64_t *d_per_vector,
                       int32_t *d_status, uint8_t *d_masks, void *d_out, const uint8_t *d_gate, uint64_t *tag,
                       uint32_t *work, hipStream_t stream) {
    switch (b.etype) {
    case CVEC_U32:
        launch_read_t<CVEC_U32>(mode, phase, b, pa, sa, desc, sec_res, vstatus, d_result, d_per_vector, d_status,
                                d_masks, d_out, d_gate, tag, work, stream);
        break;
    case CVEC_U64:
        launch_read_t<CVEC_U64>(mode, phase, b, pa, sa, desc, sec_res, vstatus, d_result, d_per_vector, d_status,
                                d_masks, d_out, d_gate, tag, work, stream);
        break;
    case CVEC_F32:
        launch_read_t<CVEC_F32>(mode, phase, b, pa, sa, desc, sec_res, vstatus, d_result, d_per_vector, d_status,
                                d_masks, d_out, d_gate, tag, work, stream);
        break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

} // namespace cvec
