// cvec_fused.hip - the whole read operation over batches of short vectors in one kernel.
//
// k_fused is the section-slot kernel (k_sections, cvec_decode.hip) with the two passes around it folded in:
//
//   * A wave owns whole vectors: those whose first section slot lies in its slot range (two 64-ary searches in
//     sec_base at the start).  Its low lanes walk the section headers of these vectors themselves, one vector per
//     lane and one header per pass of the chunk loop - VectorReader::try_new (reference src/vector.rs:404-414) and
//     the FixedSectIterator chain (src/section.rs:864-876), the job of k_index - and write the descriptors to the
//     workspace, a few chunks ahead of the octet loop that consumes them.  The dependent header loads of a vector
//     are in flight while a chunk of 16 sections is decoded, so after the first chunk the walk costs no time of
//     its own, and a header line is still in the cache hierarchy when the staging loads ask for it a few
//     microseconds later.
//   * When its range is done the wave folds the per-section results of its vectors in order (the first failing
//     section ends a vector, src/filter.rs:226, src/vector.rs:460-462 - the job of k_finalize) and adds its
//     totals to the result block.
//
// Descriptors, per-section results and per-vector header statuses land in the workspace exactly as the three-kernel
// path leaves them, so everything that reads a workspace afterwards (valid-section counts, error details, passes
// over the resident index) works unchanged.
//
// The walk of one vector is a serial chain, so a lane yields one section per loop pass and the chain of the first
// vector is exposed at the start of every wave: the kernel is for batches of many short vectors (see
// fused_eligible); long vectors keep the index pass, which walks all vectors of the batch at once.
#include <stdlib.h>

#include "cvec_device.hpp"
#include "cvec_sections.hpp"

namespace cvec {

namespace {

struct FChunk {
    uint32_t gb;   // 128-byte aligned blob offset of the span
    uint32_t n;    // sections in the chunk
    uint32_t nch;  // 16-byte pieces of the span
    uint32_t rel;  // the lane's section: start inside the region,
    uint32_t len;  //   clipped length,
    uint32_t kind; //   kind (K_SKIP beyond the chunk)
};

// smallest value over the wave
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t x) {
    x = min(x, (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0xb1 /* quad_perm:[1,0,3,2] */, 0xf, 0xf, false));
    x = min(x, (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x4e /* quad_perm:[2,3,0,1] */, 0xf, 0xf, false));
    x = min(x, (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x141 /* row_half_mirror */, 0xf, 0xf, false));
    x = min(x, (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x140 /* row_mirror */, 0xf, 0xf, false));
    const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)x, 0), b = (uint32_t)__builtin_amdgcn_readlane((int)x, 16),
                   c = (uint32_t)__builtin_amdgcn_readlane((int)x, 32), d = (uint32_t)__builtin_amdgcn_readlane((int)x, 48);
    return min(min(a, b), min(c, d));
}

// Loads that must see what this wave stored earlier in the kernel (descriptors, per-section results, header
// statuses): device scope, so they are served by the L2 and never by a line a neighbouring wave pulled into
// the CU's vector cache before the store.
__device__ __forceinline__ uint64_t load_dev_u64(const uint64_t *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint32_t load_dev_u32(const uint32_t *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

} // namespace

#ifdef CVEC_PROBES
// probe build: per-wave wall-clock stamps (100 MHz) of the kernel's phases, read back by cvec_debug_fused_times
__device__ uint64_t g_fused_stamps[2048 * 8];
#define FUSED_STAMP(i) do { if (lane == 0 && gw < 2048) g_fused_stamps[gw * 8 + (i)] = wall_clock64(); } while (0)
#else
#define FUSED_STAMP(i) do { } while (0)
#endif

template <int ETYPE, int MODE, int NPRED>
__global__ __launch_bounds__(SEC_THREADS, 2) void k_fused(
    const uint8_t *__restrict__ bytes, uint32_t padded_bytes, const uint64_t *__restrict__ byte_off,
    const uint64_t *__restrict__ sec_base, uint32_t n_vectors, uint32_t n_sections, uint32_t sections_per_wave,
    uint32_t walk_lanes, PredArgs pa, SinkArgs sa, uint64_t *desc, uint32_t *sec_res, int32_t *vstatus,
    uint64_t *__restrict__ per_vec, int32_t *__restrict__ status_out, uint8_t *__restrict__ masks, void *__restrict__ out_,
    uint64_t *__restrict__ result, uint64_t *__restrict__ tag, uint64_t cookie) {
    constexpr int W = ETYPE == CVEC_U64 ? 8 : 4;
    constexpr int PIECES = REGION_PIECES;
    __shared__ SecShared<PIECES> sm;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t gw = blockIdx.x * SEC_WAVES + wave;
    // the descriptors this kernel leaves behind are the batch's section index (an ungated one)
    if (gw == 0 && lane == 0) { tag[0] = cookie; tag[1] = 0; }
    FUSED_STAMP(0);
    const uint64_t tA64 = (uint64_t)gw * sections_per_wave;
    if (tA64 >= n_sections) return;
    const uint32_t tA = (uint32_t)tA64;
    const uint32_t tB = tA + sections_per_wave < n_sections && tA + sections_per_wave > tA ? tA + sections_per_wave : n_sections;
    const uint32_t *const off32 = (const uint32_t *)byte_off; // the blob is below 4 GiB: low words only
    const uint32_t *const sb32 = (const uint32_t *)sec_base;

    // ---- which vectors are this wave's: the first vector whose first slot is >= tA, up to the first one whose
    //      first slot is >= tB (the last wave also takes the empty vectors at the end of the batch) ------------------------
    uint32_t lo0 = 0, hi0 = n_vectors, lo1 = 0, hi1 = n_vectors;
    while (hi0 > lo0 || hi1 > lo1) {
        const uint32_t st0 = (hi0 - lo0 + 63) >> 6, st1 = (hi1 - lo1 + 63) >> 6;
        const uint32_t p0 = lo0 + (uint32_t)lane * st0 + (st0 - 1), p1 = lo1 + (uint32_t)lane * st1 + (st1 - 1);
        bool ge0 = true, ge1 = true;
        if (hi0 > lo0 && p0 < hi0) ge0 = sb32[2 * (size_t)p0] >= tA;
        if (hi1 > lo1 && p1 < hi1) ge1 = sb32[2 * (size_t)p1] >= tB;
        // (no lane answers when all 64 blocks are real and end below the target: the answer is then hi itself)
        const unsigned long long m0 = __ballot(ge0), m1 = __ballot(ge1);
        const uint32_t f0 = m0 ? (uint32_t)__builtin_ctzll(m0) : 64u, f1 = m1 ? (uint32_t)__builtin_ctzll(m1) : 64u;
        if (hi0 > lo0) {
            const uint32_t pf = lo0 + f0 * st0 + (st0 - 1);
            lo0 += f0 * st0;
            hi0 = pf < hi0 ? pf : hi0;
        }
        if (hi1 > lo1) {
            const uint32_t pf = lo1 + f1 * st1 + (st1 - 1);
            lo1 += f1 * st1;
            hi1 = pf < hi1 ? pf : hi1;
        }
    }
    const uint32_t vA = lo0, vB = tB >= n_sections ? n_vectors : lo1;
    if (vA >= vB) return;
    FUSED_STAMP(1);

    uint8_t *data = sm.data[wave];
    const uint32_t lds_base = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)data);
    const uint32_t sec = (uint32_t)lane >> 2;

    // ---- the header walk: lane l < walk_lanes walks vectors vA + l, vA + l + walk_lanes, ... ---------------------------
    auto load4 = [&](uint32_t a) -> uint32_t {
        const uint32_t a4 = a + 4 <= padded_bytes ? a : padded_bytes - 4;
        return *(const u32u *)(bytes + a4) >> (8 * (a - a4));
    };
    uint32_t wv = (uint32_t)lane < walk_lanes ? vA + (uint32_t)lane : vB; // the lane's vector; >= vB: none left
    if (wv > vB) wv = vB;
    uint32_t w_start = 0, w_len = 0, w_slot0 = 0, w_nslots = 0, w_pos = 0, w_k = 0, w_hw = 0;
    int w_st = CVEC_OK;
    bool w_fresh = false;
    uint64_t w_h8 = 0;
    uint32_t n_start = 0, n_end = 0, n_slot0 = 0, n_slot1 = 0; // offsets of the lane's next vector, asked for one vector ahead
    auto prefetch_offsets = [&](uint32_t v) {
        if (v < vB) {
            n_start = off32[2 * (size_t)v]; n_end = off32[2 * (size_t)v + 2];
            n_slot0 = sb32[2 * (size_t)v]; n_slot1 = sb32[2 * (size_t)v + 2];
        }
    };
    // takes up vector wv (its offsets are in n_*): asks for its header, its first section header and the offsets
    // of the lane's next vector
    auto begin_vector = [&]() {
        w_start = n_start; w_len = n_end - n_start; w_slot0 = n_slot0; w_nslots = n_slot1 - n_slot0;
        w_pos = HEADER_BYTES; w_k = 0; w_st = CVEC_OK; w_fresh = true;
        prefetch_offsets(wv + walk_lanes);
        if (w_len > 5) {
            const uint32_t a8 = w_start + 8 <= padded_bytes ? w_start : padded_bytes - 8;
            w_h8 = *(const u64u *)(bytes + a8) >> (8 * (w_start - a8));
        }
        if (w_len > (uint32_t)HEADER_BYTES) w_hw = load4(w_start + HEADER_BYTES);
    };
    // one step of every walking lane: the header that was asked for in the step before is parsed, the section's
    // descriptor stored and the next header asked for; a vector that ends here is closed (unused slots marked,
    // header status stored) and the lane's next vector taken up
    auto walk_step = [&]() {
        if (wv < vB) {
            bool fin = false;
            if (w_fresh) {
                w_fresh = false;
                // scroll pread error mapping of try_new, then the reader's own checks
                if (w_len == 0) w_st = CVEC_BAD_OFFSET;
                else if (w_len < 4) w_st = CVEC_NOT_ENOUGH_SPACE;
                else if (w_len <= 5) w_st = CVEC_BAD_OFFSET;
                else {
                    const uint64_t from_header = (uint32_t)w_h8;
                    const uint32_t subtype = (uint32_t)(w_h8 >> 40) & 0xffu;
                    if ((uint64_t)w_len < from_header + 4) w_st = CVEC_INPUT_TOO_SHORT;
                    else if (subtype != (uint32_t)ETYPE) w_st = CVEC_WRONG_VECTOR_TYPE;
                    else if (w_len < (uint32_t)HEADER_BYTES) w_st = CVEC_INPUT_TOO_SHORT;
                }
                if (w_st != CVEC_OK || w_pos >= w_len) fin = true;
            }
            if (!fin) {
                const uint32_t rem = w_len - w_pos;
                const uint32_t hw = w_hw, b0 = hw & 0xffu;
                uint32_t nb = 0;
                int e = CVEC_OK;
                if (b0 == K_NULL) {
                    nb = 1;
                } else if (b0 == K_CONST) {
                    if (rem < 1u + (uint32_t)W) e = CVEC_INPUT_TOO_SHORT;
                    nb = 1 + W;
                } else if (b0 == K_NP || b0 == K_DELTA || b0 == K_XOR) {
                    if (rem <= 1) e = CVEC_BAD_OFFSET;
                    else if (rem < 3) e = CVEC_NOT_ENOUGH_SPACE;
                    else {
                        const uint32_t n16 = (hw >> 8) & 0xffffu;
                        nb = b0 == K_NP ? n16 + 3 : (b0 == K_DELTA ? n16 + (uint32_t)DELTA_HDR : n16);
                        if (nb > rem) e = CVEC_SCROLL_ERR;
                        else if (nb < 3) e = CVEC_INVALID_FORMAT;
                    }
                } else {
                    e = CVEC_INVALID_SECTION_TYPE;
                }
                if (e == CVEC_OK && w_k >= w_nslots) e = CVEC_ERR_SECTION_COUNT;
                if (e) {
                    w_st = e;
                    fin = true;
                } else {
                    const bool legal = ETYPE == CVEC_F32 ? (b0 == K_NULL || b0 == K_CONST || b0 == K_XOR) : (b0 != K_XOR);
                    desc[w_slot0 + w_k] = desc_pack(w_start + w_pos, nb, legal ? b0 : K_ILLEGAL);
                    w_pos += nb;
                    w_k++;
                    if (w_pos < w_len) w_hw = load4(w_start + w_pos);
                    else fin = true;
                }
            }
            if (fin) {
                if (w_st == CVEC_OK && w_k != w_nslots) w_st = CVEC_ERR_SECTION_COUNT;
                const uint32_t fill_off = w_start + (w_pos < w_len ? w_pos : w_len);
                for (; w_k < w_nslots; w_k++) desc[w_slot0 + w_k] = desc_pack(fill_off, 0, K_SKIP);
                vstatus[wv] = w_st;
                wv += walk_lanes;
                if (wv < vB) begin_vector();
                else wv = vB;
            }
        }
    };

    // the wave's slot range: from the first slot of its first vector to the first slot of the vector behind its last
    prefetch_offsets(wv);
    const uint32_t sB_raw = sb32[2 * (size_t)vB];
    const uint32_t sA = (uint32_t)__builtin_amdgcn_readfirstlane((int)n_slot0); // lane 0 holds vector vA
    const uint32_t sB = (uint32_t)__builtin_amdgcn_readfirstlane((int)sB_raw);
    if (wv < vB) begin_vector();
    // every slot below the frontier has its descriptor stored: a lane's vectors ascend, so that is the smallest
    // "first slot without descriptor" over the lanes
    auto frontier = [&]() -> uint32_t { return wave_min_u32(wv < vB ? w_slot0 + w_k : sB); };

    auto load_desc = [&](uint32_t s) -> uint64_t {
        return s + sec < sB ? load_dev_u64(desc + s + sec) : desc_pack(0, 0, K_SKIP);
    };
    auto plan = [&](uint32_t s, uint64_t d) -> FChunk {
        FChunk c;
        const uint32_t left = s < sB ? sB - s : 0;
        const uint32_t avail = left < (uint32_t)TILE ? left : (uint32_t)TILE;
        const uint32_t off = (uint32_t)desc_off(d);
        const bool wants = sec < avail;
        const unsigned long long mw = __ballot(wants);
        const uint32_t off0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)off);
        // the span starts on a 128-byte line: every 1 KiB load instruction then covers eight whole lines
        c.gb = off0 & ~127u;
        const uint32_t len = desc_len(d);
        c.len = len < SectLimits<W>::MAX_BYTES ? len : SectLimits<W>::MAX_BYTES;
        const uint32_t endl = off + c.len - c.gb;
        const bool fits = !wants || (off >= c.gb && endl <= region_span(PIECES));
        const unsigned long long m = __ballot(fits);
        const uint32_t n = m == ~0ull ? avail : (uint32_t)__builtin_ctzll(~m) >> 2;
        c.n = n < avail ? n : avail;
        // section offsets do not decrease with the slot number: the span ends with the chunk's last section
        const unsigned long long mn = c.n >= 16 ? mw : mw & ((1ull << (4 * c.n)) - 1ull);
        const uint32_t e = mn ? (uint32_t)__builtin_amdgcn_readlane((int)endl, 63 - (int)__builtin_clzll(mn)) : 0u;
        c.nch = (e + 15) >> 4;
        c.rel = off - c.gb;
        c.kind = sec < c.n ? desc_kind(d) : K_SKIP;
        return c;
    };

    typedef uint32_t __attribute__((__vector_size__(16))) u32x4;
    u32x4 R[PIECES];
    auto issue_loads = [&](const FChunk &c) {
        const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(bytes + c.gb), 0, (int)(c.nch * 16u), 0x00020000);
#pragma unroll
        for (int i = 0; i < PIECES; i++)
            R[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane * 16, i * 1024, 2); // nt: every byte is read once
    };

#pragma unroll
    for (int i = 0; i < PIECES; i++) R[i] = u32x4{0, 0, 0, 0};
    uint32_t s_cur = sA;
    FChunk cur;
    cur.gb = 0; cur.n = 0; cur.nch = 0; cur.rel = 0; cur.len = 0; cur.kind = K_SKIP;
    // the descriptors of the first chunk: the only part of the walk that nothing hides
    {
        const uint32_t need = sA + TILE < sB ? sA + TILE : sB;
        while (frontier() < need) walk_step();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    FUSED_STAMP(2);
    uint64_t d_next = load_desc(sA);
    uint32_t res_word = 0, mk_lo = 0, mk_hi = 0;
    uint32_t res_slot = 0;
    bool res_pending = false, mk_pending = false;
    asm volatile("" ::"v"(d_next));
    for (;;) {
        // ---- registers -> staging region (everything this wave has in flight lands here: the staged bytes, the
        //      next descriptors, the walkers' headers, and the stores of the pass before) ------------------------------
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        {
            u32x4 *dst = (u32x4 *)data;
#pragma unroll
            for (int i = 0; i < PIECES; i++) dst[64 * i + lane] = R[i];
        }
        __builtin_amdgcn_sched_barrier(0);
        if (res_pending) sec_res[res_slot] = res_word;
        res_pending = false;
        if constexpr (MODE == MODE_FILTER_MASKS) {
            if (mk_pending) {
                uint32_t *m = (uint32_t *)(masks + (size_t)res_slot * 32) + (lane & 3);
                m[0] = mk_lo; m[4] = mk_hi;
            }
            mk_pending = false;
        }
        // ---- plan the next chunk; make sure the descriptors of the one after it are stored -----------------------------
        const uint32_t s_next = s_cur + cur.n;
        const FChunk nxt = plan(s_next, d_next);
        {
            const uint32_t s_after = s_next + nxt.n;
            const uint32_t need = s_after + TILE < sB ? s_after + TILE : sB;
            // Only what was stored before the wait above counts: the descriptor loads below then cannot overtake
            // the stores they depend on.
            if (frontier() < need) {
                do walk_step(); while (frontier() < need);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else if (__ballot(wv < vB) != 0) {
                walk_step(); // the free step: its loads and stores fly while the chunk below is decoded
            }
        }
        issue_loads(nxt);
        d_next = load_desc(s_next + nxt.n);
        // ---- walk + decode the current chunk ----------------------------------------------------------------------------
        const bool has_octets = cur.kind == K_NP || cur.kind == K_DELTA || cur.kind == K_XOR;
        if (MODE != MODE_DECODE && cur.n != 0 && __ballot(has_octets) == 0) {
            // a chunk of Null / Const sections only: every section's count and mask follow from its kind
            // (src/filter.rs:222-223, :104-106; a Const section is 256 copies of one value)
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
        } else if (cur.n != 0) {
            uint32_t ooff;
            const out_rsrc_t orsrc = chunk_out_rsrc<W, MODE>(out_, (uint64_t)s_cur + sec, lane, ooff);
            const bool odd = process_chunk<ETYPE, MODE, NPRED, region_span(PIECES)>(
                data, lds_base, cur.kind, cur.rel, cur.len, sec < cur.n, lane, pa, sa, res_word, mk_lo, mk_hi, orsrc, ooff);
            if constexpr (W == 4) {
                const unsigned long long odd_m = __ballot(odd);
                if (odd_m != 0) {
                    if ((odd_m >> ((uint32_t)lane & ~3u)) & 0xfull)
                        section_exact<ETYPE, MODE, NPRED>(data, lds_base, cur.kind, cur.rel, cur.len, lane, pa, sa, res_word, mk_lo,
                                                          mk_hi, orsrc, ooff);
                }
            }
            res_slot = s_cur + sec;
            res_pending = sec < cur.n && (lane & 3) == 0;
            mk_pending = sec < cur.n && cur.kind != K_SKIP && cur.kind != K_ILLEGAL;
        }
        if (nxt.n == 0) break;
        s_cur = s_next;
        cur = nxt;
    }
    if (res_pending) sec_res[res_slot] = res_word;
    if constexpr (MODE == MODE_FILTER_MASKS) {
        if (mk_pending) {
            uint32_t *m = (uint32_t *)(masks + (size_t)res_slot * 32) + (lane & 3);
            m[0] = mk_lo; m[4] = mk_hi;
        }
    }
    FUSED_STAMP(3);
    // vectors without any section slot at the end of the range (the chunk loop ends with the last slot)
    while (__ballot(wv < vB) != 0) walk_step();

    // ---- per-vector results: the sections of a vector in order up to the first one that failed -----------------------
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    uint64_t acc_cnt = 0;
    uint32_t acc_nerr = 0, acc_max = 0;
    for (uint32_t v0 = vA; v0 < vB; v0 += 64) {
        const uint32_t v = v0 + (uint32_t)lane;
        if (v < vB) {
            const uint32_t a = sb32[2 * (size_t)v], b = sb32[2 * (size_t)v + 2];
            uint32_t cnt = 0, st = 0;
            for (uint32_t s = a; s < b; s++) {
                const uint32_t r = load_dev_u32(sec_res + s);
                if (r >> 16) { st = r >> 16; break; }
                cnt += r & 0xffffu;
            }
            if (!st) st = load_dev_u32((const uint32_t *)vstatus + v);
            if (per_vec) per_vec[v] = cnt;
            if (status_out) status_out[v] = (int32_t)st;
            acc_cnt += cnt;
            acc_nerr += st ? 1u : 0u;
            acc_max = st > acc_max ? st : acc_max;
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        acc_cnt += __shfl_down(acc_cnt, d, 64);
        acc_nerr += __shfl_down(acc_nerr, d, 64);
        const uint32_t o = __shfl_down(acc_max, d, 64);
        acc_max = o > acc_max ? o : acc_max;
    }
    FUSED_STAMP(4);
    if (lane == 0) {
        if (acc_cnt) atomicAdd((unsigned long long *)&result[0], (unsigned long long)acc_cnt);
        if (acc_nerr) {
            atomicAdd((unsigned long long *)&result[1], (unsigned long long)acc_nerr);
            atomicMax((unsigned long long *)&result[2], (unsigned long long)acc_max);
        }
    }
}

#ifdef CVEC_PROBES
hipError_t debug_fused_times(uint64_t *h_out) { return hipMemcpyFromSymbol(h_out, HIP_SYMBOL(g_fused_stamps), sizeof(uint64_t) * 2048 * 8); }
#endif

// The result block is accumulated into by every wave: it has to be zero when the kernel starts.
__global__ void k_result_zero(uint64_t *__restrict__ result) {
    if (threadIdx.x < CVEC_RESULT_WORDS) result[threadIdx.x] = 0;
}

// Which batches take the one-kernel path when nothing is forced: many short vectors (the header chain of a
// wave's first vector is exposed, and a lane yields one section per loop pass, so 16 walking lanes keep up with
// the octet loop only while vectors are short), offsets that fit 32 bits, and sections large enough that the
// section-slot path would run its 18 KiB regions anyway (the count-only shape with 12 KiB regions and three
// workgroups per CU has no room for the walk's registers).
bool fused_possible(const cvec_dbatch_view &b) {
    if (b.n_vectors == 0 || b.n_sections == 0) return false;
    if (b.n_vectors >= (1ull << 31) || b.n_sections >= (1ull << 31)) return false;
    if (b.n_bytes >= (1ull << 32) - (1ull << 20)) return false;
    return true;
}
bool fused_eligible(int mode, const cvec_dbatch_view &b, bool has_masks) {
    if (!fused_possible(b)) return false;
    if (b.n_vectors < 8192 || b.n_sections > 16 * b.n_vectors) return false;
    const bool count_only = mode != MODE_DECODE && !has_masks;
    const bool small = count_only && (b.n_bytes / b.n_sections) * TILE + 2 * 16 <= region_span(REGION_PIECES_SMALL);
    return !small;
}

template <int ETYPE>
static void launch_fused_t(int mode, const cvec_dbatch_view &b, const PredArgs &pa, const SinkArgs &sa, uint64_t *desc,
                           uint32_t *sec_res, int32_t *vstatus, uint64_t *d_result, uint64_t *d_per_vector, int32_t *d_status,
                           uint8_t *d_masks, void *d_out, uint64_t *tag, uint64_t cookie, hipStream_t stream) {
    const uint32_t n = (uint32_t)b.n_vectors, ns = (uint32_t)b.n_sections;
    const uint32_t nwaves_max = 256u * 2u * SEC_WAVES;
    uint32_t per = (ns + nwaves_max - 1) / nwaves_max;
    per = (per + TILE - 1) / TILE * TILE;
    if (per < 2 * TILE) per = 2 * TILE;
    const uint32_t nwaves = (ns + per - 1) / per;
    const uint32_t nwg = (nwaves + SEC_WAVES - 1) / SEC_WAVES;
    const uint32_t padded = (uint32_t)((b.n_bytes + 15) & ~15ull);
    static const uint32_t walk_lanes = [] {
        const char *e = getenv("CVEC_FUSED_WALK_LANES");
        const int x = e ? atoi(e) : 0;
        return (uint32_t)(x >= 1 && x <= 64 ? x : 16);
    }();
    PredArgs pb = pa;
    uint32_t bucket = 1;
    while (bucket < pb.count) bucket *= 2;
    for (uint32_t i = pb.count; i < bucket && i < CVEC_MAX_PRED_VALUES; i++) pb.v[i] = pb.v[0];
    SinkArgs sb = sa;
    sb._pad = 0;
#define CVEC_FU(M, N)                                                                                                        \
    k_fused<ETYPE, M, N><<<nwg, SEC_THREADS, 0, stream>>>(b.d_bytes, padded, b.d_byte_off, b.d_sec_base, n, ns, per, walk_lanes, \
                                                         pb, sb, desc, sec_res, vstatus, d_per_vector, d_status, d_masks,     \
                                                         d_out, d_result, tag, cookie)
#define CVEC_FU_BUCKET(M)                                                                                                    \
    switch (bucket) {                                                                                                        \
    case 1: CVEC_FU(M, 1); break;                                                                                            \
    case 2: CVEC_FU(M, 2); break;                                                                                            \
    case 4: CVEC_FU(M, 4); break;                                                                                            \
    case 8: CVEC_FU(M, 8); break;                                                                                            \
    default: CVEC_FU(M, 16); break;                                                                                          \
    }
    if (mode == MODE_DECODE) {
        // the store policy of the decoded output: see launch_sections
        const uint64_t raw = (uint64_t)ns * (uint64_t)(FIXED_LEN * 4);
        const bool nt_out = ETYPE != CVEC_U64 && b.n_bytes * 20 >= raw && b.n_bytes * 20 <= raw * 7;
        if (nt_out) CVEC_FU(MODE_DECODE, 2); else CVEC_FU(MODE_DECODE, 1);
    } else if (d_masks) { CVEC_FU_BUCKET(MODE_FILTER_MASKS); }
    else if (bucket == 1) {
        const uint32_t fb = (uint32_t)pa.v[0];
        const bool bits_ok = ETYPE != CVEC_F32 || ((fb & 0x7fffffffu) != 0 && (fb & 0x7fffffffu) <= 0x7f800000u);
        if (bits_ok) CVEC_FU(MODE_FILTER_BITS, 1); else CVEC_FU(MODE_FILTER, 1);
    } else {
        CVEC_FU_BUCKET(MODE_FILTER);
    }
#undef CVEC_FU_BUCKET
#undef CVEC_FU
}

// One operation = a one-wave kernel that clears the result block + k_fused.
hipError_t launch_fused(int mode, const cvec_dbatch_view &b, const PredArgs &pa, const SinkArgs &sa, uint64_t *desc,
                        uint32_t *sec_res, int32_t *vstatus, uint64_t *d_result, uint64_t *d_per_vector, int32_t *d_status,
                        uint8_t *d_masks, void *d_out, uint64_t *tag, uint64_t cookie, hipStream_t stream) {
    k_result_zero<<<1, 64, 0, stream>>>(d_result);
    switch (b.etype) {
    case CVEC_U32:
        launch_fused_t<CVEC_U32>(mode, b, pa, sa, desc, sec_res, vstatus, d_result, d_per_vector, d_status, d_masks, d_out, tag, cookie, stream);
        break;
    case CVEC_U64:
        launch_fused_t<CVEC_U64>(mode, b, pa, sa, desc, sec_res, vstatus, d_result, d_per_vector, d_status, d_masks, d_out, tag, cookie, stream);
        break;
    case CVEC_F32:
        launch_fused_t<CVEC_F32>(mode, b, pa, sa, desc, sec_res, vstatus, d_result, d_per_vector, d_status, d_masks, d_out, tag, cookie, stream);
        break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

} // namespace cvec
