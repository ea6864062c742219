// cvec_vstream.hip — the vector-streaming read kernel: decode + sink / decode + filter of a batch of
// many short vectors in ONE launch, without a section index.
//
// The section-slot kernel (cvec_decode.hip) gives every lane quad one of 16 CONSECUTIVE sections, so the
// start of every section has to be known before the octet loop runs: a separate pass chases the section
// headers of every vector through HBM first (FixedSectIterator::next, reference src/section.rs:864-876)
// and a third one folds the per-section results per vector.  Here the 16 quads of a wave work on 16
// DIFFERENT vectors, one section each per pass of the octet loop: the header of a quad's next section
// sits right behind the section it has just staged, exactly as the reference's iterator finds it, so
// the chain is followed in LDS as a by-product, every byte of the blob is read once, and a vector's
// count and status stay in the quad's registers until the vector ends (filter::count_hits,
// src/filter.rs:296-299; the iteration stops at the first failing section, src/filter.rs:226,
// src/vector.rs:460-462).
//
//   wave     owns a contiguous range of vectors; its quads take the next vector of the range as they
//            finish one.
//   quad     keeps a window of its vector's bytes in its own LDS slot (16-byte pieces, loaded by the
//            quad's four lanes: 64 contiguous bytes per load instruction and quad).  While the octet loop
//            runs over the current windows, the next windows (each starting at the quad's next section
//            header, whose position follows from the header of the current section) are prefetched into
//            registers, sized by the length of the section before.
//   per pass (1) registers -> LDS slots, (2) every quad parses at its cursor: vector header checks
//            (VectorReader::try_new, src/vector.rs:404-414), then section headers; Null / Const
//            sections are settled on the spot (src/filter.rs:222-223, :104-106; src/section.rs:454-460,
//            :784-791), an octet-bearing section becomes the quad's work of this pass, (3) windows that
//            turned out too short are fetched again at full length, (4) the next windows are requested,
//            (5) the shared octet loop (process_chunk) decodes the 16 sections, (6) results are folded.
//
// Used for batches of many vectors with few sections each (see vstream_eligible); everything else, gated
// passes and passes over a resident index take the section-slot path.
#include <stdlib.h>
#include <string.h>

#include "cvec_sections.hpp"

namespace cvec {

template <int SP> struct VsShape {
    static constexpr uint32_t SLOT_PIECES = SP;
    static constexpr uint32_t SLOT_BYTES = SP * 16u;
    // Slots are 16 bytes further apart than they are long: the stride is an odd multiple of 4 dwords, so the
    // 16 quads, which mostly read at similar offsets of their slots, spread over all 64 LDS banks.
    static constexpr uint32_t STRIDE = SLOT_BYTES + 16u;
    static constexpr uint32_t ZPAD = TILE * STRIDE; // 64 zero bytes: quads without a section run the loop over them
    static constexpr uint32_t REGION = ZPAD + 64u;
    static constexpr int LOADS = SP / 4;
    static_assert(SP % 4 == 0 && ((STRIDE / 16u) & 1u) == 1u, "slot stride must be an odd number of 16-byte pieces");
};
constexpr int VS_PIECES_LARGE = 72; // 1152-byte slots: any 32-bit section (at most 1100 bytes) at any 16-byte phase
constexpr uint32_t VS_NONE = 0xffffffffu;
constexpr int VS_MAX_RUN = 96; // Null / Const sections a quad settles per pass before it asks for a new window

// Probe build with -DCVEC_PROBE_CYCLES only: cycles per phase of the main loop, summed over all waves into the
// words 16.. of the workspace's scratch area (tag + 8 ...), read by tools/vstream_probe.py.  The counter reads
// slow the kernel down by a third: the shares are meaningful, the absolute time is not.
#ifdef CVEC_PROBE_CYCLES
#define VS_PROBE_DECL uint64_t pr_t[8] = {0, 0, 0, 0, 0, 0, 0, 0}; uint64_t pr_last = __builtin_readcyclecounter();
#define VS_PROBE(i) do { const uint64_t now_ = __builtin_readcyclecounter(); pr_t[i] += now_ - pr_last; pr_last = now_; } while (0)
#define VS_PROBE_FLUSH do { if (lane == 0) for (int i_ = 0; i_ < 8; i_++) atomicAdd((unsigned long long *)&tag[8 + i_], (unsigned long long)pr_t[i_]); } while (0)
#else
#define VS_PROBE_DECL
#define VS_PROBE(i)
#define VS_PROBE_FLUSH
#endif

// Per wave: 16 slots + zero pad, the table of the wave's upcoming vectors (64 entries of start, end, first
// section slot, section count) and the wave's totals.
template <int SP> struct VsWave {
    __attribute__((aligned(16))) uint8_t region[VsShape<SP>::REGION];
    __attribute__((aligned(16))) uint32_t table[64][4];
    __attribute__((aligned(16))) unsigned long long acc_cnt;
    uint32_t acc_nerr, acc_max;
};
template <int SP> struct VsShared {
    VsWave<SP> w[SEC_WAVES];
};

constexpr uint32_t VS_K_NEW = 0xffffffffu; // value of the section counter while the vector's header is unchecked

// workgroups of the kernel per CU, by window size (stated on the kernel; the build checks the kernel descriptors
// against it, see the Makefile)
#ifndef VS_MINBLOCKS
#define VS_MINBLOCKS(sp) ((sp) <= 48 ? 3 : 2)
#endif
constexpr int vs_wg_per_cu(int sp) { return VS_MINBLOCKS(sp); }
template <int ETYPE, int MODE, int NPRED, int SP>
__global__ __launch_bounds__(SEC_THREADS) __attribute__((amdgpu_waves_per_eu(vs_wg_per_cu(SP), vs_wg_per_cu(SP)))) void k_vstream(
    const uint8_t *__restrict__ bytes, uint32_t padded_bytes, const uint64_t *__restrict__ byte_off,
    const uint64_t *__restrict__ sec_base, uint32_t n, uint32_t per, PredArgs pa, SinkArgs sa,
    uint64_t *__restrict__ per_vec, int32_t *__restrict__ status_out, uint8_t *__restrict__ masks, void *__restrict__ out_,
    uint32_t out_bytes, uint32_t *__restrict__ nvalid, uint64_t *__restrict__ result, uint64_t *__restrict__ tag,
    uint32_t keep_cached) {
    constexpr int W = ETYPE == CVEC_U64 ? 8 : 4;
    using U = typename Elem<W>::U;
    using S = VsShape<SP>;
    __shared__ VsShared<SP> sm;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t gw = blockIdx.x * SEC_WAVES + wave;
    if (gw == 0 && lane == 0) { tag[0] = 0; tag[1] = VS_MAGIC; } // no section index here; per-vector valid-section counts
    const uint64_t vA64 = (uint64_t)gw * per;
    if (vA64 >= n) return;
    const uint32_t vA = (uint32_t)vA64;
    const uint32_t vB = n - vA < per ? n : vA + per;
    VsWave<SP> &me = sm.w[wave];
    uint8_t *data = me.region;
    const uint32_t lds_base = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)data);
    const uint32_t q = (uint32_t)lane & 3u, w = (uint32_t)lane >> 2;
    const uint32_t slot_lds = w * S::STRIDE; // the quad's slot, relative to the region
    if (lane < 16) ((uint32_t *)(data + S::ZPAD))[lane] = 0;
    if (lane == 0) { me.acc_cnt = 0; me.acc_nerr = 0; me.acc_max = 0; }

    const auto brsrc = __builtin_amdgcn_make_buffer_rsrc((void *)bytes, 0, (int)padded_bytes, 0x00020000);
    const out_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(out_, 0, MODE == MODE_DECODE ? (int)out_bytes : 0, 0x00020000);
    const bool addc = sa.add_const != 0;
    typedef uint32_t __attribute__((__vector_size__(16))) u32x4;

    // ---- the wave's table of upcoming vectors: entry l describes vector tb + l -----------------------------------
    uint32_t tb = vA;
    auto load_table = [&]() {
        const uint32_t v = tb + (uint32_t)lane;
        const bool ok = v < vB && v >= tb;
        const uint64_t a = ok ? byte_off[v] : 0, b = ok ? byte_off[v + 1] : 0;
        const uint64_t s0 = ok ? sec_base[v] : 0, s1 = ok ? sec_base[v + 1] : 0;
        *(u32x4 *)me.table[lane] = u32x4{(uint32_t)a, (uint32_t)b, (uint32_t)s0, (uint32_t)(s1 - s0)};
    };
    load_table();
    // every pass of the main loop moves each busy quad on by a section, a vector boundary or a new window
    // (at most one per header); the cap only guards against a wave that would otherwise never drain
    const uint32_t slots_first = (uint32_t)sec_base[vA], slots_last = (uint32_t)sec_base[vB];
    uint32_t passes_left = 3u * (slots_last - slots_first) + 6u * (vB - vA) + 64u;

    // ---- quad state (the same in the four lanes of a quad) ------------------------------------------------------
    uint32_t vec = VS_NONE; // the quad's vector; vend = its end in the blob, pos = the cursor, k = sections done
    uint32_t vend = 0, pos = 0, k = 0, nslots = 0, slot0 = 0;
    uint32_t st = 0;        // non-zero: the vector has failed and ends at the next look
    uint32_t kvalid = 0;    // sections that produced a result; bit 31: it ended at a header that does not parse
    uint64_t cnt = 0;
    uint32_t rvec = VS_NONE; // the vector reserved for when this one ends (its first window is on its way)
    // The quad's slot is a sliding window over its vector's bytes: blob bytes [wbase, wend) sit at slot offset 0 ...
    // (wbase 16-byte aligned).  Planned for the next pass: the new base nwbase and, packed into `plan`, pieces
    // [f0, f1) of the new window (on their way from HBM into the staging registers) and c_n pieces carried over
    // inside the slot from offset nwbase - wbase (the bytes behind the section just decoded: the head of the next
    // one).  Fetches end on a 128-byte line of the blob wherever the slot has room, so that a line is requested once.
    uint32_t wbase = 0, wend = 0, nwbase = 0, plan = 0;
    auto mkplan = [](uint32_t f0, uint32_t f1, uint32_t cn) -> uint32_t { return f0 | (f1 << 8) | (cn << 16); };
    uint32_t vnext = vA; // wave-uniform: the next vector of the range that no quad has taken
    uint32_t mk_lo = 0, mk_hi = 0, mk_slot = 0, mks_lo = 0, mks_hi = 0;
    bool mk_pending = false;

    u32x4 R[S::LOADS];
#pragma unroll
    for (int i = 0; i < S::LOADS; i++) R[i] = u32x4{0, 0, 0, 0};
    // Fetch mapping: main load w (w = 0..15) takes pieces 0..63 of window w, one per lane: 1 KiB of contiguous blob per
    // instruction, its base and piece range scalars (readlane from the owning quad); the pieces from 64 on go into
    // TAILS tail loads, eight lanes per window (128 contiguous bytes).  Lanes outside a window's range aim past the
    // descriptor: no traffic, zeros.
    constexpr int TAILS = (S::SLOT_PIECES - 64) / 8 * 2;      // 8 pieces x 8 windows per tail load
    static_assert(S::SLOT_PIECES >= 64 && (S::SLOT_PIECES - 64) % 8 == 0 && 16 + TAILS == S::LOADS, "piece mapping");
    const uint32_t lane16 = 16u * (uint32_t)lane;
    const uint32_t tail_src = ((uint32_t)lane >> 3) << 4;      // bpermute address of the quad that owns this lane's tail window (of 8)
    // (the cache policy is an immediate of the load instruction: non-temporal for blobs far beyond the last-level
    // cache, where every byte is read once per pass; the default policy for the count and mask passes over smaller
    // blobs, which then stay cached from one pass over a resident batch to the next)
    auto issue_loads_p = [&](auto policy, uint32_t base, uint32_t pl) { // pieces [f0, f1) of plan word pl, for every quad's window
        constexpr int POL = decltype(policy)::value;
#pragma unroll
        for (int wi = 0; wi < TILE; wi++) {
            const uint32_t ba = (uint32_t)__builtin_amdgcn_readlane((int)base, 4 * wi);
            const uint32_t ff = (uint32_t)__builtin_amdgcn_readlane((int)pl, 4 * wi);
            const bool mine = (uint32_t)lane >= (ff & 0xffu) && (uint32_t)lane < ((ff >> 8) & 0xffu);
            R[wi] = __builtin_amdgcn_raw_buffer_load_b128(brsrc, mine ? ba + lane16 : 0xffffff00u, 0, POL);
        }
#pragma unroll
        for (int t = 0; t < TAILS; t++) {
            // tail load t: windows 8 (t & 1) .. +7, pieces 64 + 8 (t >> 1) .. +7
            const int src = (int)(tail_src + 128u * (uint32_t)(t & 1));
            const uint32_t ba = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)base);
            const uint32_t ff = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)pl);
            const uint32_t pc = 64u + 8u * (uint32_t)(t >> 1) + ((uint32_t)lane & 7u);
            const bool mine = pc >= (ff & 0xffu) && pc < ((ff >> 8) & 0xffu);
            R[TILE + t] = __builtin_amdgcn_raw_buffer_load_b128(brsrc, mine ? ba + 16u * pc : 0xffffff00u, 0, POL);
        }
    };
    auto issue_loads = [&](uint32_t base, uint32_t pl) {
        if (keep_cached) issue_loads_p(std::integral_constant<int, 0>(), base, pl);
        else issue_loads_p(std::integral_constant<int, 2>(), base, pl);
    };
    // staging registers -> slots: only the pieces that were requested (the rest of a slot keeps its bytes)
    auto write_slots = [&](uint32_t pl) {
#pragma unroll
        for (int wi = 0; wi < TILE; wi++) {
            const uint32_t ff = (uint32_t)__builtin_amdgcn_readlane((int)pl, 4 * wi);
            u32x4 *dst = (u32x4 *)(data + wi * S::STRIDE) + lane;
            if ((uint32_t)lane >= (ff & 0xffu) && (uint32_t)lane < ((ff >> 8) & 0xffu)) *dst = R[wi];
        }
#pragma unroll
        for (int t = 0; t < TAILS; t++) {
            const int src = (int)(tail_src + 128u * (uint32_t)(t & 1));
            const uint32_t ff = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)pl);
            const uint32_t pc = 64u + 8u * (uint32_t)(t >> 1) + ((uint32_t)lane & 7u);
            const uint32_t win = 8u * (uint32_t)(t & 1) + ((uint32_t)lane >> 3);
            u32x4 *dst = (u32x4 *)(data + win * S::STRIDE) + pc;
            if (pc >= (ff & 0xffu) && pc < ((ff >> 8) & 0xffu)) *dst = R[TILE + t];
        }
    };
    // length in bytes that the section whose first four bytes are h claims (type byte + u16, src/section.rs:252-271),
    // cut to what a section of this element type can use; 1 for anything that is no section
    auto claimed_len = [&](uint32_t h) -> uint32_t {
        const uint32_t b0 = h & 0xffu, n16 = (h >> 8) & 0xffffu;
        uint32_t nb = 1;
        if (b0 == K_CONST) nb = 1 + W;
        else if (b0 == K_NP) nb = n16 + 3u;
        else if (b0 == K_DELTA) nb = n16 + (uint32_t)DELTA_HDR;
        else if (b0 == K_XOR) nb = n16;
        return nb < SectLimits<W>::MAX_BYTES ? nb : SectLimits<W>::MAX_BYTES;
    };
    auto lds_read4 = [&](uint32_t a) -> uint32_t { // 4 bytes at LDS address a (any alignment)
        lds_u32p p = (lds_u32p)(uintptr_t)(a & ~3u);
        return __builtin_amdgcn_alignbit(p[1], p[0], (a & 3u) << 3);
    };
    // 12 bytes at LDS address a (any alignment) as three dwords
    auto lds_read12 = [&](uint32_t a, uint32_t &h0, uint32_t &h1, uint32_t &h2) {
        lds_u32p p = (lds_u32p)(uintptr_t)(a & ~3u);
        const uint32_t s = (a & 3u) << 3;
        const uint32_t d0 = p[0], d1 = p[1], d2 = p[2], d3 = p[3];
        h0 = __builtin_amdgcn_alignbit(d1, d0, s);
        h1 = __builtin_amdgcn_alignbit(d2, d1, s);
        h2 = __builtin_amdgcn_alignbit(d3, d2, s);
    };

    VS_PROBE_DECL
    for (;;) {
        VS_PROBE(7);
        // ---- (1) slide the windows: carried bytes to the front of the slot, then the fetched pieces behind them --------
        {
            const uint32_t c_n = plan >> 16;
            if (__ballot(c_n != 0) != 0) {
                // at most 12 pieces per quad, three per lane; all reads of the wave precede all writes (the ranges overlap)
                const u32x4 *src = (const u32x4 *)(data + slot_lds + (nwbase - wbase));
                u32x4 *dst = (u32x4 *)(data + slot_lds);
                u32x4 t0 = {0, 0, 0, 0}, t1 = t0, t2 = t0;
                if (q < c_n) t0 = src[q];
                if (q + 4u < c_n) t1 = src[q + 4u];
                if (q + 8u < c_n) t2 = src[q + 8u];
                __builtin_amdgcn_sched_barrier(0);
                if (q < c_n) dst[q] = t0;
                if (q + 4u < c_n) dst[q + 4u] = t1;
                if (q + 8u < c_n) dst[q + 8u] = t2;
            }
            VS_PROBE(6);
            write_slots(plan);
            const uint32_t f0 = plan & 0xffu, f1 = (plan >> 8) & 0xffu;
            if (f1 > f0) wend = nwbase + 16u * f1;
            else if (c_n == 0 && nwbase != wbase) wend = nwbase; // moved without bytes: an empty window
            wbase = nwbase;
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (mode_has_masks(MODE)) {
            // the masks of the last pass leave behind the wait for the staged bytes (a store in flight in front
            // of that wait would add its round trip to it)
            if (mk_pending) *(uint2 *)(masks + (uint64_t)mk_slot * 32 + 8u * q) = uint2{mks_lo, mks_hi};
            mk_pending = false;
        }
        VS_PROBE(0);
        // ---- (2) every quad parses at its cursor ---------------------------------------------------------------------
        bool hasS = false, topup = false;
        uint32_t S_pos = 0, S_nb = 0, S_len = 0, S_kind = K_SKIP;
        // The common case, straight-line: the quad is in the middle of a vector and the next section is a well-formed
        // octet-bearing one of a legal kind.  Everything else (vector boundaries, Null / Const sections, any error, a
        // header outside the window) goes through the general loop below.
        bool slow = true;
        if (vec != VS_NONE && st == 0 && k != VS_K_NEW && pos < vend && pos >= wbase && pos + 3u <= wend) {
            const uint32_t rem = vend - pos;
            const uint32_t h0 = lds_read4(lds_base + slot_lds + (pos - wbase));
            const uint32_t b0 = h0 & 0xffu, n16 = (h0 >> 8) & 0xffffu;
            const bool legal = ETYPE == CVEC_F32 ? b0 == K_XOR : (b0 == K_NP || b0 == K_DELTA);
            const uint32_t nb = b0 == K_NP ? n16 + 3u : (b0 == K_DELTA ? n16 + (uint32_t)DELTA_HDR : n16);
            const uint32_t len = nb < SectLimits<W>::MAX_BYTES ? nb : SectLimits<W>::MAX_BYTES;
            if (legal && rem >= 3u && nb <= rem && nb >= 3u && k < nslots && pos - wbase + len <= S::SLOT_BYTES) {
                S_len = len; S_pos = pos; S_nb = nb; S_kind = b0;
                topup = pos + len > wend;
                hasS = true;
                slow = false;
            }
        }
        if (__ballot(slow) != 0) {
            bool go = slow;
            int run = 0;
            // MODE_DECODE: runs of sections that are one value each (Null, Const) are noted here, up to two per quad
            // and pass (neighbouring runs of one value grow into one), and written by the whole wave behind the
            // walk: 1 KiB per store instruction instead of 64 bytes per quad.
            uint32_t fa_slot = 0, fa_n = 0, fa_y = 0, fb_slot = 0, fb_n = 0, fb_y = 0;
            while (go) {
                if (vec == VS_NONE) {
                    if (rvec == VS_NONE) { go = false; continue; }
                    const u32x4 e = *(const u32x4 *)me.table[rvec - tb];
                    vec = rvec; rvec = VS_NONE;
                    pos = e[0]; vend = e[1]; slot0 = e[2]; nslots = e[3];
                    k = VS_K_NEW; cnt = 0; st = 0; kvalid = 0;
                }
                if (st != 0 || (k != VS_K_NEW && pos >= vend)) {
                    // the vector ends here: the sections found must be the ones its header announced
                    if (st == 0 && k != nslots) { st = CVEC_ERR_SECTION_COUNT; kvalid |= 0x80000000u; }
                    if (q == 0) {
                        if (per_vec) per_vec[vec] = cnt;
                        if (status_out) status_out[vec] = (int32_t)st;
                        nvalid[vec] = kvalid;
                        if (cnt) atomicAdd(&me.acc_cnt, (unsigned long long)cnt);
                        if (st) { atomicAdd(&me.acc_nerr, 1u); atomicMax(&me.acc_max, st); }
                    }
                    vec = VS_NONE; st = 0;
                    continue;
                }
                if (k == VS_K_NEW) {
                    // VectorReader::try_new (src/vector.rs:404-414), scroll's pread errors included; pos = the vector's start
                    const uint32_t L = vend - pos;
                    uint32_t e = CVEC_OK;
                    if (L == 0) e = CVEC_BAD_OFFSET;
                    else if (L < 4) e = CVEC_NOT_ENOUGH_SPACE;
                    else if (L <= 5) e = CVEC_BAD_OFFSET;
                    else {
                        uint32_t h0, h1, h2;
                        lds_read12(lds_base + slot_lds + (pos - wbase), h0, h1, h2);
                        const uint32_t subtype = (h1 >> 8) & 0xffu;
                        if ((uint64_t)L < (uint64_t)h0 + 4u) e = CVEC_INPUT_TOO_SHORT;
                        else if (subtype != (uint32_t)ETYPE) e = CVEC_WRONG_VECTOR_TYPE;
                        else if (L < (uint32_t)HEADER_BYTES) e = CVEC_INPUT_TOO_SHORT;
                    }
                    k = 0;
                    if (e) { st = e; kvalid |= 0x80000000u; continue; }
                    pos += HEADER_BYTES;
                    continue;
                }
                // ---- section header at pos (FixedSectEnum::try_from, src/section.rs:252-271) ----------------------------
                if (pos >= wend || pos < wbase) { go = false; continue; } // not in the window: the next one starts here
                const uint32_t rem = vend - pos;
                uint32_t h0, h1, h2;
                lds_read12(lds_base + slot_lds + (pos - wbase), h0, h1, h2);
                const uint32_t b0 = h0 & 0xffu;
                const bool octets = b0 == K_NP || b0 == K_DELTA || b0 == K_XOR;
                const uint32_t hdr_need = b0 == K_CONST ? (rem < 1u + W ? 1u : 1u + W) : (octets && rem >= 3u ? 3u : 1u);
                if (pos + hdr_need > wend) { go = false; continue; }
                uint32_t nb = 0, e = CVEC_OK;
                if (b0 == K_NULL) {
                    nb = 1;
                } else if (b0 == K_CONST) {
                    if (rem < 1u + W) e = CVEC_INPUT_TOO_SHORT;
                    nb = 1 + W;
                } else if (octets) {
                    if (rem <= 1) e = CVEC_BAD_OFFSET;
                    else if (rem < 3) e = CVEC_NOT_ENOUGH_SPACE;
                    else {
                        const uint32_t n16 = (h0 >> 8) & 0xffffu;
                        nb = b0 == K_NP ? n16 + 3u : (b0 == K_DELTA ? n16 + (uint32_t)DELTA_HDR : n16);
                        if (nb > rem) e = CVEC_SCROLL_ERR;
                        else if (nb < 3) e = CVEC_INVALID_FORMAT;
                    }
                } else {
                    e = CVEC_INVALID_SECTION_TYPE;
                }
                if (e == CVEC_OK && k >= nslots) e = CVEC_ERR_SECTION_COUNT;
                if (e) { st = e; kvalid |= 0x80000000u; continue; }
                if (b0 == K_NULL || b0 == K_CONST) {
                    // settled without an octet loop: 256 times zero / the value behind the type byte
                    U v = 0;
                    if (b0 == K_CONST) {
                        if constexpr (W == 4) v = (h0 >> 8) | (h1 << 24);
                        else v = ((uint64_t)(h0 >> 8) | ((uint64_t)h1 << 24)) | ((uint64_t)h2 << 56);
                    }
                    // A run of Null sections (one zero byte each, src/section.rs:440-447: what append_nulls and sparse
                    // data leave behind) is settled at once: as many as the 12 bytes just read show, the vector, its
                    // announced section count and the window allow.
                    uint32_t r = 1;
                    if (b0 == K_NULL) {
                        const uint32_t zeros = h0 ? (uint32_t)__builtin_ctz(h0) >> 3
                                                  : (h1 ? 4u + ((uint32_t)__builtin_ctz(h1) >> 3) : (h2 ? 8u + ((uint32_t)__builtin_ctz(h2) >> 3) : 12u));
                        r = zeros;
                        r = r < rem ? r : rem;
                        r = r < nslots - k ? r : nslots - k;
                        r = r < wend - pos ? r : wend - pos;
                        r = r < (uint32_t)(VS_MAX_RUN - run) ? r : (uint32_t)(VS_MAX_RUN - run);
                    }
                    const uint32_t slot = slot0 + k;
                    if constexpr (MODE == MODE_DECODE) {
                        U y = v;
                        if (addc) {
                            // AddConstSink (src/sink.rs:304-317): a Null section forwards the base itself
                            if (b0 == K_NULL) y = (U)sa.base;
                            else if constexpr (ETYPE == CVEC_F32)
                                y = __float_as_uint(__uint_as_float((uint32_t)v) + __uint_as_float((uint32_t)sa.base));
                            else y = v + (U)sa.base;
                        }
                        u32x4 yv;
                        if constexpr (W == 4) yv = u32x4{(uint32_t)y, (uint32_t)y, (uint32_t)y, (uint32_t)y};
                        else yv = u32x4{(uint32_t)y, (uint32_t)((uint64_t)y >> 32), (uint32_t)y, (uint32_t)((uint64_t)y >> 32)};
                        bool noted = true;
                        if constexpr (W == 4) {
                            if (fa_n != 0 && fa_slot + fa_n == slot && fa_y == (uint32_t)y) fa_n += r;
                            else if (fa_n == 0) { fa_slot = slot; fa_n = r; fa_y = (uint32_t)y; }
                            else if (fb_n != 0 && fb_slot + fb_n == slot && fb_y == (uint32_t)y) fb_n += r;
                            else if (fb_n == 0) { fb_slot = slot; fb_n = r; fb_y = (uint32_t)y; }
                            else noted = false;
                        } else {
                            noted = false;
                        }
                        if (!noted) { // a third run of this pass: by the quad itself
                            const uint32_t o0 = slot * (uint32_t)(FIXED_LEN * W) + 16u * q;
                            for (uint32_t j = 0; j < r * (uint32_t)(FIXED_LEN * W / 64); j++)
                                __builtin_amdgcn_raw_buffer_store_b128(yv, orsrc, o0 + 64u * j, 0, 0);
                        }
                        cnt += r * (uint32_t)FIXED_LEN;
                    } else {
                        const bool all = b0 == K_NULL ? pa.match_zero != 0 : pred_match<ETYPE, NPRED>(pa, v);
                        cnt += all ? r * (uint32_t)FIXED_LEN : 0u;
                        if constexpr (mode_has_masks(MODE)) {
                            for (uint32_t j = 0; j < r; j++) {
                                *(uint2 *)(masks + (uint64_t)(slot + j) * 32 + 8u * q) = all ? uint2{~0u, ~0u} : uint2{0u, 0u};
                            }
                        }
                    }
                    pos += b0 == K_NULL ? r : nb;
                    k += r; kvalid += r;
                    run += (int)r;
                    if (run >= VS_MAX_RUN) go = false;
                    continue;
                }
                const bool legal = ETYPE == CVEC_F32 ? b0 == K_XOR : b0 != K_XOR;
                if (!legal) { st = CVEC_INVALID_FORMAT; continue; }
                // an octet-bearing section: the quad's work of this pass
                S_len = nb < SectLimits<W>::MAX_BYTES ? nb : SectLimits<W>::MAX_BYTES;
                S_pos = pos; S_nb = nb; S_kind = b0;
                // where it lies in the slot it may not fit (the window grew by appending): the next window starts at
                // it, and there it fits (a 32-bit section is at most 1100 bytes, a slot 1152)
                if (pos - wbase + S_len > S::SLOT_BYTES) { go = false; continue; }
                topup = pos + S_len > wend;
                hasS = true;
                go = false;
            }
            if constexpr (MODE == MODE_DECODE && W == 4) {
                // the noted runs, quad by quad: slot, length and value come from the quad's first lane
                auto fill_runs = [&](uint32_t f_slot, uint32_t f_n, uint32_t f_y) {
                    unsigned long long todo = __ballot(f_n != 0) & 0x1111111111111111ull;
                    while (todo) {
                        const int l = (int)__builtin_ctzll(todo);
                        todo &= todo - 1;
                        const uint32_t s0 = (uint32_t)__builtin_amdgcn_readlane((int)f_slot, l),
                                       nr = (uint32_t)__builtin_amdgcn_readlane((int)f_n, l),
                                       yy = (uint32_t)__builtin_amdgcn_readlane((int)f_y, l);
                        const u32x4 yv = u32x4{yy, yy, yy, yy};
                        uint32_t o = s0 * (uint32_t)(FIXED_LEN * W) + 16u * (uint32_t)lane;
                        for (uint32_t j = 0; j < nr; j++, o += (uint32_t)(FIXED_LEN * W))
                            __builtin_amdgcn_raw_buffer_store_b128(yv, orsrc, o, 0, 0);
                    }
                };
                fill_runs(fa_slot, fa_n, fa_y);
                fill_runs(fb_slot, fb_n, fb_y);
            }
        }
        VS_PROBE(1);
        // ---- (3) windows that are shorter than their section: fetch the rest -------------------------------------------------
        if (__ballot(topup) != 0) {
            // append up to the end of the section's last 128-byte line (or the end of the slot)
            uint32_t tp = 0;
            if (topup) {
                uint32_t fend = (S_pos + S_len + 3u + 127u) & ~127u;
                if (fend - wbase > S::SLOT_BYTES) fend = wbase + S::SLOT_BYTES;
                tp = mkplan((wend - wbase) >> 4, (fend - wbase) >> 4, 0);
            }
            issue_loads(wbase, tp);
            write_slots(tp);
            if (topup) wend = wbase + 16u * ((tp >> 8) & 0xffu);
        }
        VS_PROBE(2);
        // ---- (4) the next windows ---------------------------------------------------------------------------------------------
        bool want_new = false;
        nwbase = wbase; plan = 0;
        if (vec != VS_NONE && st == 0) {
            const uint32_t np = hasS ? S_pos + S_nb : pos; // where the next section header is
            if (hasS && np >= vend) want_new = rvec == VS_NONE;
            else {
                // what the next pass needs: the section at np whole, and the three header bytes behind it for the
                // look-ahead of the pass after that.  Its length is read from its header when that is in the window
                // already; otherwise guessed from the section before it / the vector's average.
                uint32_t nlen;
                if (np >= wbase && np + 3u <= wend) nlen = claimed_len(lds_read4(lds_base + slot_lds + (np - wbase)));
                else if (hasS) nlen = S_len + 48u;
                else nlen = (vend - np) / (nslots > k ? nslots - k : 1u) + 96u;
                uint32_t need_end = np + nlen + 3u;
                if (need_end > vend || need_end < np) need_end = vend;
                if (need_end > wend || np < wbase) {
                    uint32_t fend = (need_end + 127u) & ~127u;
                    if (np >= wbase && fend - wbase <= S::SLOT_BYTES) {
                        plan = mkplan((wend - wbase) >> 4, (fend - wbase) >> 4, 0);  // append behind what the slot holds
                    } else {
                        nwbase = np & ~15u;                                            // slide: the section moves to the front
                        uint32_t left = wend > nwbase && nwbase >= wbase ? (wend - nwbase) >> 4 : 0u;
                        if (left > 12u) left = 0;                                      // too much to carry: fetch it again
                        if (fend - nwbase > S::SLOT_BYTES) fend = nwbase + S::SLOT_BYTES;
                        uint32_t f1 = (fend - nwbase) >> 4;
                        if (f1 < left) f1 = left;
                        plan = mkplan(left, f1, left);
                    }
                }
            }
        } else {
            want_new = rvec == VS_NONE;
        }
        {
            const unsigned long long m4 = __ballot(want_new);
            const uint32_t total = (uint32_t)__builtin_popcountll(m4) >> 2;
            if (total != 0) {
                if (vnext - tb + TILE > 64u && tb + 64u < vB) { tb = vnext; load_table(); }
                const uint32_t rank = (uint32_t)__builtin_popcountll(m4 & ((1ull << (4u * w)) - 1ull)) >> 2;
                const uint32_t cand = vnext + rank;
                if (want_new && cand < vB && cand >= vnext) {
                    const u32x4 e = *(const u32x4 *)me.table[cand - tb];
                    rvec = cand;
                    const uint32_t e_start = e[0], L = e[1] - e[0], e_ns = e[3];
                    nwbase = e_start & ~15u;
                    uint32_t guess = HEADER_BYTES + 64u + (L > (uint32_t)HEADER_BYTES && e_ns ? (L - HEADER_BYTES) / e_ns : 0u);
                    guess = guess < L ? guess : L;
                    if (guess < 8u) guess = 8u; // the header checks look at the first 6 bytes wherever they fall
                    uint32_t fend = (e_start + guess + 127u) & ~127u;
                    if (fend - nwbase > S::SLOT_BYTES) fend = nwbase + S::SLOT_BYTES;
                    plan = mkplan(0, (fend - nwbase) >> 4, 0);
                }
                vnext = vB - vnext < total ? vB : vnext + total;
            }
        }
        VS_PROBE(5);
        issue_loads(nwbase, plan);
        VS_PROBE(3);
        // ---- (5) the octet loop over the quads' sections ---------------------------------------------------------------------
        uint32_t res_word = 0;
        const uint32_t pos_after = S_pos + S_nb;
#ifdef CVEC_PROBES
        if (__ballot(hasS) != 0 && sa._pad != 1) {
#else
        if (__ballot(hasS) != 0) {
#endif
            const uint32_t kind = hasS ? S_kind : K_SKIP;
            const uint32_t rel = slot_lds + (S_pos - wbase);
            const uint32_t ooff = (slot0 + k) * (uint32_t)(FIXED_LEN * W) + 2u * q * (uint32_t)W;
            const bool odd = process_chunk<ETYPE, MODE, NPRED, S::ZPAD>(data, lds_base, kind, rel, S_len, hasS, lane, pa, sa,
                                                                   res_word, mk_lo, mk_hi, orsrc, ooff);
            if constexpr (W == 4) {
                const unsigned long long odd_m = __ballot(odd); // a lane only sees the octets that store its own elements
                if (odd_m != 0) {
                    if ((odd_m >> ((uint32_t)lane & ~3u)) & 0xfull) section_exact<ETYPE, MODE, NPRED>(data, lds_base, kind, rel, S_len, lane, pa, sa, res_word, mk_lo,
                                                               mk_hi, orsrc, ooff);
                }
            }
        }
        VS_PROBE(4);
        // ---- (6) fold --------------------------------------------------------------------------------------------------------------
        if (hasS) {
            const uint32_t status = res_word >> 16;
            if (status) st = status;
            else {
                cnt += res_word & 0xffffu;
                kvalid++;
                if constexpr (mode_has_masks(MODE)) { mk_pending = true; mk_slot = slot0 + k; mks_lo = mk_lo; mks_hi = mk_hi; }
            }
            pos = pos_after;
            k++;
        }
        if (__ballot(vec != VS_NONE || rvec != VS_NONE) == 0) break;
        if (--passes_left == 0) {
            // never reached by a correct build; if it is, the call must not look successful
            if (lane == 0) { atomicAdd(&me.acc_nerr, 1u); atomicMax(&me.acc_max, (uint32_t)CVEC_ERR_WORKSPACE); }
            break;
        }
    }
    if constexpr (mode_has_masks(MODE)) {
        if (mk_pending) *(uint2 *)(masks + (uint64_t)mk_slot * 32 + 8u * q) = uint2{mks_lo, mks_hi};
    }
    VS_PROBE_FLUSH;
    // ---- the wave's totals ------------------------------------------------------------------------------------------------------
    if (lane == 0) {
        const unsigned long long c = me.acc_cnt;
        const uint32_t e = me.acc_nerr, m = me.acc_max;
        if (c) atomicAdd((unsigned long long *)&result[0], c);
        if (e) {
            atomicAdd((unsigned long long *)&result[1], (unsigned long long)e);
            atomicMax((unsigned long long *)&result[2], (unsigned long long)m);
        }
    }
}

// Which batches take the vector-streaming kernel: 32-bit elements (a section never outgrows a slot), offsets
// that fit the 32-bit buffer addressing of the kernel, and, unless forced, many vectors that consist mostly of
// Null / Const sections (on average at most 48 bytes per section): there the quads settle runs of such sections
// side by side without an octet loop (2.1 times the section-slot path on the reference's sparse vectors).  On dense
// data both kernels run the same octet loop and this one pays more per pass for its windows (measured 7 to 30 %
// slower), so dense batches stay on the section-slot path.
bool vstream_eligible(int mode, const cvec_dbatch_view &b, bool forced) {
    if (b.etype == CVEC_U64) return false;
    if (b.n_vectors == 0 || b.n_vectors >= (1ull << 31)) return false;
    if (b.n_bytes >= (1ull << 32) - (1ull << 20) || b.n_sections >= (1ull << 32) - 1) return false;
    if (mode == MODE_DECODE && b.n_sections * (uint64_t)FIXED_LEN * 4 >= (1ull << 32) - (1ull << 20)) return false;
    if (forced) return true;
    return b.n_vectors >= 16384 && b.n_sections <= 64 * b.n_vectors && b.n_bytes <= 48 * b.n_sections;
}

template <int ETYPE>
static void launch_vstream_t(int mode, const cvec_dbatch_view &b, const PredArgs &pa, const SinkArgs &sa, uint32_t *nvalid,
                             uint64_t *d_result, uint64_t *d_per_vector, int32_t *d_status, uint8_t *d_masks, void *d_out,
                             uint64_t *tag, hipStream_t stream) {
    const uint32_t n = (uint32_t)b.n_vectors;
    // persistent waves: 2 workgroups x 4 waves per CU, each wave 16 vectors at a time from its own range
    const uint32_t max_waves = 256u * 2u * SEC_WAVES;
    uint32_t nwaves = (n + TILE - 1) / TILE;
    if (nwaves > max_waves) nwaves = max_waves;
    const uint32_t per = (n + nwaves - 1) / nwaves;
    nwaves = (n + per - 1) / per;
    const uint32_t nwg = (nwaves + SEC_WAVES - 1) / SEC_WAVES;
    const uint32_t padded = (uint32_t)((b.n_bytes + 15) & ~15ull);
    const uint32_t out_bytes = mode == MODE_DECODE ? (uint32_t)(b.n_sections * FIXED_LEN * 4) : 0u;
    PredArgs pb = pa;
    uint32_t bucket = 1;
    while (bucket < pb.count) bucket *= 2;
    for (uint32_t i = pb.count; i < bucket && i < CVEC_MAX_PRED_VALUES; i++) pb.v[i] = pb.v[0];
    SinkArgs sb = sa;
    uint32_t keep_cached = mode != MODE_DECODE && b.n_bytes <= (640ull << 20) ? 1u : 0u; // as in launch_sections
#ifdef CVEC_PROBES
    static const int ablate = [] { const char *e = getenv("CVEC_ABLATE"); return e ? atoi(e) : 0; }();
    sb._pad = (uint32_t)ablate; // probe build: 1 = everything but the octet loop
    if (ablate == 6) keep_cached = 1;
    if (ablate == 7) keep_cached = 0;
    (void)hipMemsetAsync(tag + 8, 0, 64, stream);
#endif
#define CVEC_VS(M, N)                                                                                                      \
    k_vstream<ETYPE, M, N, VS_PIECES_LARGE><<<nwg, SEC_THREADS, 0, stream>>>(b.d_bytes, padded, b.d_byte_off, b.d_sec_base, n, \
                                                                            per, pb, sb, d_per_vector, d_status, d_masks,  \
                                                                            d_out, out_bytes, nvalid, d_result, tag, keep_cached)
#define CVEC_VS_BUCKET(M)                                                                                                  \
    switch (bucket) {                                                                                                      \
    case 1: CVEC_VS(M, 1); break;                                                                                          \
    case 2: CVEC_VS(M, 2); break;                                                                                          \
    case 4: CVEC_VS(M, 4); break;                                                                                          \
    case 8: CVEC_VS(M, 8); break;                                                                                          \
    default: CVEC_VS(M, 16); break;                                                                                        \
    }
    if (mode == MODE_DECODE) CVEC_VS(MODE_DECODE, 1);
    else if (bucket == 1) {
        // bit-pattern equality is IEEE == unless the predicate is a float zero or NaN
        const uint32_t fb = (uint32_t)pa.v[0];
        const bool bits_ok = ETYPE != CVEC_F32 || ((fb & 0x7fffffffu) != 0 && (fb & 0x7fffffffu) <= 0x7f800000u);
        if (d_masks) { if (bits_ok) CVEC_VS(MODE_FILTER_MASKS_BITS, 1); else CVEC_VS(MODE_FILTER_MASKS, 1); }
        else if (bits_ok) CVEC_VS(MODE_FILTER_BITS, 1);
        else CVEC_VS(MODE_FILTER, 1);
    }
    else if (d_masks) { CVEC_VS_BUCKET(MODE_FILTER_MASKS); }
    else {
        CVEC_VS_BUCKET(MODE_FILTER);
    }
#undef CVEC_VS_BUCKET
#undef CVEC_VS
}

// One operation = a 32-byte memset of the result block + the kernel.
hipError_t launch_vstream(int mode, const cvec_dbatch_view &b, const PredArgs &pa, const SinkArgs &sa, uint32_t *nvalid,
                          uint64_t *d_result, uint64_t *d_per_vector, int32_t *d_status, uint8_t *d_masks, void *d_out,
                          uint64_t *tag, hipStream_t stream) {
    if (b.etype == CVEC_U32)
        launch_vstream_t<CVEC_U32>(mode, b, pa, sa, nvalid, d_result, d_per_vector, d_status, d_masks, d_out, tag, stream);
    else if (b.etype == CVEC_F32)
        launch_vstream_t<CVEC_F32>(mode, b, pa, sa, nvalid, d_result, d_per_vector, d_status, d_masks, d_out, tag, stream);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

} // namespace cvec
