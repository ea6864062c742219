// cvec_encode.hip — gfx950 kernels for the write side: Vector{U32,U64,F32Xor}Appender::encode_all
// (reference src/vector.rs:238-246) for a batch of vectors.
//
// One wavefront encodes one 256-element section at a time.  Lane L holds elements 4L..4L+3, so the
// lane pair (2o, 2o+1) holds octet o and the octet-level quantities of NibblePack (reference
// src/nibblepacking.rs:108-153) are computed once per section for all 32 octets:
//   nonzero bitmask      = the lane's four nonzero bits joined with its partner's (one DPP swap)
//   nibble word          = clz/ctz of the OR of the lane's values and its partner's
//   octet byte offsets   = wavefront scan of the 32 octet sizes (DPP row shifts + row broadcasts)
//   bitstream            = every lane ORs its (adjacent) fields into a zeroed LDS image of the section
// Because a section's byte offset depends on every section before it, the work is two passes over
// the values: sizes → exclusive scan of section sizes → emit.
#include "cvec_device.hpp"

namespace cvec {

hipError_t launch_exclusive_scan(const uint64_t *d_in, uint64_t n, uint64_t *d_out, uint64_t *d_scratch,
                                 hipStream_t stream);
uint64_t scan_scratch_words(uint64_t n);

constexpr int ENC_THREADS = 256; // 4 waves = 4 sections per workgroup

template <int ETYPE> struct EncT {
    static constexpr int W = ETYPE == CVEC_U64 ? 8 : 4;
    using U = typename Elem<W>::U;
};

// Cross-lane moves as DPP (register-to-register, no LDS round trip): CTRL is the DPP control word,
// ROW_MASK selects the 16-lane rows that are written (the others keep `old`), BOUND zero-fills lanes
// whose source lies outside the row.
template <int CTRL, int ROW_MASK, bool BOUND> __device__ inline uint32_t dpp_mov(uint32_t old, uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)v, CTRL, ROW_MASK, 0xf, BOUND);
}
template <int CTRL, int ROW_MASK, bool BOUND> __device__ inline uint64_t dpp_mov(uint64_t old, uint64_t v) {
    return (uint64_t)dpp_mov<CTRL, ROW_MASK, BOUND>((uint32_t)old, (uint32_t)v) |
           ((uint64_t)dpp_mov<CTRL, ROW_MASK, BOUND>((uint32_t)(old >> 32), (uint32_t)(v >> 32)) << 32);
}
template <int CTRL, int ROW_MASK, bool BOUND> __device__ inline float dpp_mov(float old, float v) {
    return __uint_as_float(dpp_mov<CTRL, ROW_MASK, BOUND>(__float_as_uint(old), __float_as_uint(v)));
}
constexpr int DPP_QUAD_1032 = 0xb1, DPP_QUAD_2301 = 0x4e, DPP_ROW_HALF_MIRROR = 0x141, DPP_ROW_MIRROR = 0x140,
              DPP_ROW_SHR2 = 0x112, DPP_ROW_SHR4 = 0x114, DPP_ROW_SHR8 = 0x118, DPP_ROW_BCAST15 = 0x142, DPP_ROW_BCAST31 = 0x143;

__device__ inline uint32_t lane63(uint32_t x) { return (uint32_t)__builtin_amdgcn_readlane((int)x, 63); }
__device__ inline uint64_t lane63(uint64_t x) { return (uint64_t)lane63((uint32_t)x) | ((uint64_t)lane63((uint32_t)(x >> 32)) << 32); }
__device__ inline float lane63(float x) { return __uint_as_float(lane63(__float_as_uint(x))); }

// Value of lane r (uniform r), the same in every lane.
__device__ inline uint32_t lane_val(uint32_t x, int r) { return (uint32_t)__builtin_amdgcn_readlane((int)x, r); }
__device__ inline uint64_t lane_val(uint64_t x, int r) {
    return (uint64_t)lane_val((uint32_t)x, r) | ((uint64_t)lane_val((uint32_t)(x >> 32), r) << 32);
}

// Wave-wide reduction with an idempotent-safe combine (min / max / or: combining a lane with itself
// is harmless, which the masked row broadcasts rely on); the result is uniform (read from lane 63).
template <class T, class Op> __device__ inline T wave_reduce(T x, Op op) {
    x = op(x, dpp_mov<DPP_QUAD_1032, 0xf, false>(x, x));
    x = op(x, dpp_mov<DPP_QUAD_2301, 0xf, false>(x, x));
    x = op(x, dpp_mov<DPP_ROW_HALF_MIRROR, 0xf, false>(x, x));
    x = op(x, dpp_mov<DPP_ROW_MIRROR, 0xf, false>(x, x));        // every lane: its row's result
    x = op(x, dpp_mov<DPP_ROW_BCAST15, 0xa, false>(x, x));       // rows 1, 3 <- rows 0, 2
    x = op(x, dpp_mov<DPP_ROW_BCAST31, 0xc, false>(x, x));       // rows 2, 3 <- rows 0..1
    return lane63(x);
}
template <class U> __device__ inline U wave_min_u(U x) {
    return wave_reduce<U>(x, [](U a, U b) { return b < a ? b : a; });
}
template <class U> __device__ inline U wave_max_u(U x) {
    return wave_reduce<U>(x, [](U a, U b) { return b > a ? b : a; });
}

// Minimum of `mn` and maximum of `mx` over the wave, both at once, with the lane exchange fused into
// the min / max instruction (the compiler keeps a separate DPP move and a copy per step).  A VGPR
// written by a VALU instruction may be read through DPP two wait states later: the two chains
// alternate and an s_nop fills the second state.  Rows that a masked row broadcast does not write
// keep their value.  For floats no NaN may come in.
#define CVEC_MINMAX_STEP(OPMIN, OPMAX, CTRL)                                                                          \
    OPMIN " %0, %0, %0 " CTRL " bank_mask:0xf\n\t" OPMAX " %1, %1, %1 " CTRL " bank_mask:0xf\n\ts_nop 0\n\t"
#define CVEC_MINMAX_WAVE(OPMIN, OPMAX)                                                                                \
    "s_nop 1\n\t" CVEC_MINMAX_STEP(OPMIN, OPMAX, "quad_perm:[1,0,3,2] row_mask:0xf")                                \
        CVEC_MINMAX_STEP(OPMIN, OPMAX, "quad_perm:[2,3,0,1] row_mask:0xf")                                            \
        CVEC_MINMAX_STEP(OPMIN, OPMAX, "row_half_mirror row_mask:0xf")                                                \
        CVEC_MINMAX_STEP(OPMIN, OPMAX, "row_mirror row_mask:0xf")                                                     \
        CVEC_MINMAX_STEP(OPMIN, OPMAX, "row_bcast:15 row_mask:0xa")                                                   \
        CVEC_MINMAX_STEP(OPMIN, OPMAX, "row_bcast:31 row_mask:0xc") "s_nop 0"
__device__ inline void wave_minmax(float &mn, float &mx) {
    asm("; wave min/max f32\n\t" CVEC_MINMAX_WAVE("v_min_f32_dpp", "v_max_f32_dpp") : "+v"(mn), "+v"(mx));
    mn = lane63(mn); mx = lane63(mx);
}
__device__ inline void wave_minmax(uint32_t &mn, uint32_t &mx) {
    asm("; wave min/max u32\n\t" CVEC_MINMAX_WAVE("v_min_u32_dpp", "v_max_u32_dpp") : "+v"(mn), "+v"(mx));
    mn = lane63(mn); mx = lane63(mx);
}
__device__ inline void wave_minmax(uint64_t &mn, uint64_t &mx) {
    mn = wave_min_u<uint64_t>(mn);
    mx = wave_max_u<uint64_t>(mx);
}
__device__ inline uint32_t bitlen_u(uint32_t x) { return x ? 32u - (uint32_t)__clz(x) : 0u; }
__device__ inline uint32_t bitlen_u(uint64_t x) { return x ? 64u - (uint32_t)__clzll((long long)x) : 0u; }

// Section plan, identical in every lane of the wave.
template <class U> struct SecPlan {
    uint32_t kind;
    U minv;      // Delta base
    U maxv;
    U first;     // values[0] (Const payload)
};

// SectionWriterStats::from_vect + writer choice (reference src/section.rs:473-480, :719-732,
// :823-845).  For f32 the fold keeps the first minimum / last maximum and treats an unordered
// compare as Equal: min is NaN iff values[0] is NaN (NaNs after it are skipped); max restarts after
// every NaN, so it is NaN iff the last value is, else the maximum of the values after the last NaN.
template <int ETYPE>
__device__ inline SecPlan<typename EncT<ETYPE>::U> plan_section(const typename EncT<ETYPE>::U x[4], int lane) {
    using U = typename EncT<ETYPE>::U;
    SecPlan<U> p;
    p.first = lane_val(x[0], 0);
    if constexpr (ETYPE == CVEC_F32) {
        float f[4];
#pragma unroll
        for (int i = 0; i < 4; i++) f[i] = __uint_as_float(x[i]);
        const float inf = __builtin_inff();
        float mn, mx;
        int q = -1; // index of the last NaN
        const unsigned long long nanb = __ballot(__builtin_isunordered(f[0], f[1]) || __builtin_isunordered(f[2], f[3]));
        if (nanb == 0) {
            // No NaN in the section (the usual case): minimum == maximum exactly when every value compares equal to
            // the first one, and nothing else of the two is used for f32 (no Delta sections; a Const section stores
            // values[0]).  Four compares and a ballot instead of two reductions over the wave.
            const float f0 = __uint_as_float(p.first);
            const bool differs = (f[0] != f0) | (f[1] != f0) | (f[2] != f0) | (f[3] != f0);
            p.minv = p.maxv = p.first;
            p.kind = __ballot(differs) != 0 ? (uint32_t)K_XOR : (f0 == 0.0f ? (uint32_t)K_NULL : (uint32_t)K_CONST);
            return p;
        } else {
            uint32_t nan4 = 0;
#pragma unroll
            for (int i = 0; i < 4; i++) nan4 |= (uint32_t)(f[i] != f[i]) << i;
            const int lq = 63 - __builtin_clzll(nanb);
            q = 4 * lq + 31 - __clz((int)__builtin_amdgcn_readlane((int)nan4, lq));
            mn = inf; mx = -inf;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                mn = f[i] < mn ? f[i] : mn; // false for a NaN
                mx = (4 * lane + i > q && f[i] > mx) ? f[i] : mx;
            }
        }
        // no NaN reaches the reductions, and which zero survives does not matter below
        wave_minmax(mn, mx);
        const float first = __uint_as_float(p.first);
        if (first != first) mn = first;                  // min fold is stuck on a leading NaN
        if (q == FIXED_LEN - 1) mx = __builtin_nanf(""); // max fold ends on a trailing NaN
        p.minv = __float_as_uint(mn);
        p.maxv = __float_as_uint(mx);
        if (mn == mx) p.kind = (mn == 0.0f) ? K_NULL : K_CONST;
        else p.kind = K_XOR;
    } else {
        U mn = x[0], mx = x[0];
#pragma unroll
        for (int i = 1; i < 4; i++) { mn = x[i] < mn ? x[i] : mn; mx = x[i] > mx ? x[i] : mx; }
        wave_minmax(mn, mx);
        p.minv = mn;
        p.maxv = mx;
        if (mn == mx) p.kind = (mn == 0) ? K_NULL : K_CONST;
        else {
            const uint32_t reg = (bitlen_u(mx) + 3) / 4, rng = (bitlen_u((U)(mx - mn)) + 3) / 4;
            p.kind = rng < reg ? K_DELTA : K_NP;
        }
    }
    return p;
}

// What NibblePack sees: XOR against the previous octet (f32, reference src/section.rs:738-743),
// value - min (Delta, :656) or the value itself.
template <int ETYPE>
__device__ inline void transform_section(typename EncT<ETYPE>::U x[4], uint32_t kind,
                                         typename EncT<ETYPE>::U minv, int lane) {
    using U = typename EncT<ETYPE>::U;
    if constexpr (ETYPE == CVEC_F32) {
        // element e-8 is the same register two lanes down
        U prev[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const U below = __shfl(x[i], (lane + 62) & 63, 64);
            prev[i] = lane >= 2 ? below : (U)0;
        }
#pragma unroll
        for (int i = 0; i < 4; i++) x[i] ^= prev[i];
    } else {
        if (kind == K_DELTA) {
#pragma unroll
            for (int i = 0; i < 4; i++) x[i] -= minv;
        }
    }
}

// Inclusive scan over the 32 octets of a wave (both lanes of pair o hold the value of octet o) and
// its total.  With the odd lanes counted as 0, the first step of a lane-wise scan (shift by one)
// reproduces the input, so the scan starts at shift two: row shifts, then row broadcasts.
struct PairScan { uint32_t incl, total; };
__device__ inline PairScan pair_scan_total(uint32_t v) {
    PairScan r;
    uint32_t x = v + dpp_mov<DPP_ROW_SHR2, 0xf, true>(0u, v);
    x += dpp_mov<DPP_ROW_SHR4, 0xf, true>(0u, x);
    x += dpp_mov<DPP_ROW_SHR8, 0xf, true>(0u, x);
    x += dpp_mov<DPP_ROW_BCAST15, 0xa, false>(0u, x);
    x += dpp_mov<DPP_ROW_BCAST31, 0xc, false>(0u, x);
    r.incl = x;
    r.total = (uint32_t)__builtin_amdgcn_readlane((int)x, 63);
    return r;
}

struct OctInfo {
    uint32_t mask;   // nonzero bitmask of this lane's octet
    uint32_t m4;     // nonzero bits of this lane's four values
    uint32_t n4, t4; // bits kept per value, trailing zero bits dropped (both multiples of 4)
    uint32_t size;   // octet bytes
};

template <class U> __device__ inline OctInfo octet_info(const U x[4], int lane) {
    OctInfo o;
    U r = (x[0] | x[1]) | (x[2] | x[3]);
    r |= dpp_mov<DPP_QUAD_1032, 0xf, false>(r, r);
    o.m4 = (uint32_t)(x[0] != 0) | (uint32_t)(x[1] != 0) << 1 | (uint32_t)(x[2] != 0) << 2 | (uint32_t)(x[3] != 0) << 3;
    const uint32_t pm = dpp_mov<DPP_QUAD_1032, 0xf, false>(o.m4, o.m4);
    o.mask = (lane & 1) ? (pm | o.m4 << 4) : (o.m4 | pm << 4);
    if (o.mask) {
        uint32_t lz, tz;
        if constexpr (sizeof(U) == 8) { lz = (uint32_t)__clzll((long long)r); tz = (uint32_t)__ffsll((long long)r) - 1; }
        else { lz = 32u + (uint32_t)__clz((int)r); tz = (uint32_t)__ffs((int)r) - 1; }
        const uint32_t t = tz >> 2, n = 16 - (lz >> 2) - t;
        o.t4 = 4 * t;
        o.n4 = 4 * n;
        o.size = 2 + ((n * __popc(o.mask) + 1) >> 1);
    } else {
        o.t4 = 0; o.n4 = 0; o.size = 1;
    }
    return o;
}

// Values of one section through a buffer descriptor of `avail` elements (what is left of the value
// array, at most one section): 16-byte loads without a branch, and avail == 0 is a harmless no-op
// load.  Branch-free loads let the compiler count them, so the loads of the next section stay in
// flight under the work on this one.  Elements past `valid` become the zero padding of finish()
// (src/vector.rs:352-355).
typedef uint32_t __attribute__((__vector_size__(16))) u32x4_t;
// Values are read once per pass and the encoded bytes written once: non-temporal loads and stores
// (measured together: +3.5 % on f32, +2 % on u32, -2 % on u64 delta against the default policy).
#ifndef CVEC_ENC_LD_AUX
#define CVEC_ENC_LD_AUX 2
#endif
template <int ETYPE>
__device__ inline void load_section(const typename EncT<ETYPE>::U *__restrict__ values, uint64_t e0, uint32_t valid,
                                    uint32_t avail, int lane, typename EncT<ETYPE>::U x[4]) {
    constexpr int W = EncT<ETYPE>::W;
    typedef uint32_t __attribute__((__vector_size__(16))) u32x4;
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(values + e0), 0, (int)(avail * W), 0x00020000);
    if constexpr (W == 4) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane * 16, 0, CVEC_ENC_LD_AUX);
#pragma unroll
        for (int i = 0; i < 4; i++) x[i] = v[i];
    } else {
        const u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane * 32, 0, CVEC_ENC_LD_AUX);
        const u32x4 b = __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane * 32, 16, CVEC_ENC_LD_AUX);
        x[0] = (uint64_t)a[0] | ((uint64_t)a[1] << 32); x[1] = (uint64_t)a[2] | ((uint64_t)a[3] << 32);
        x[2] = (uint64_t)b[0] | ((uint64_t)b[1] << 32); x[3] = (uint64_t)b[2] | ((uint64_t)b[3] << 32);
    }
#pragma unroll
    for (int i = 0; i < 4; i++) x[i] = (uint32_t)(4 * lane + i) < valid ? x[i] : 0;
}

// ---- vector → section bookkeeping -------------------------------------------------------------
__global__ void k_enc_nsec(const uint64_t *__restrict__ elem_off, uint64_t n, uint64_t *__restrict__ nsec) {
    uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n) return;
    nsec[v] = (elem_off[v + 1] - elem_off[v] + FIXED_LEN - 1) / FIXED_LEN;
}

__global__ void k_enc_map(const uint64_t *__restrict__ sec_base, uint64_t n, uint32_t *__restrict__ sec_vec) {
    uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n) return;
    for (uint64_t s = sec_base[v]; s < sec_base[v + 1]; s++) sec_vec[s] = (uint32_t)v;
}

// ---- runs of sections ---------------------------------------------------------------------------------
// A wave encodes a run of `run` consecutive section slots.  Lane l looks up where slot0 + l lives
// (three dependent loads, paid once per run instead of once per section); the loop reads the
// description of section r with readlane and loads the values of section r + 1 before it works on
// section r, so every wave keeps two sections of loads in flight.
constexpr int ENC_MAX_RUN = 16;

struct RunMeta {
    uint64_t e0;    // first element of the section in `values`
    uint32_t valid; // elements that exist (the rest of the section is zero padding); 0 = no section
    uint32_t avail; // elements of the value array from e0 on, at most one section
    uint32_t vec;   // the vector the section belongs to
};

__device__ inline RunMeta run_meta(const uint64_t *__restrict__ elem_off, const uint64_t *__restrict__ sec_base,
                                   const uint32_t *__restrict__ sec_vec, uint64_t n_vectors, uint64_t slot, bool active) {
    RunMeta m;
    m.e0 = 0; m.valid = 0; m.avail = 0; m.vec = 0;
    const uint64_t n_slots = sec_base[n_vectors], n_values = elem_off[n_vectors];
    if (active && slot < n_slots) {
        const uint32_t v = sec_vec[slot];
        m.vec = v;
        const uint64_t k = slot - sec_base[v];
        m.e0 = elem_off[v] + k * FIXED_LEN;
        const uint64_t left = elem_off[v + 1] - m.e0;
        m.valid = (uint32_t)(left < FIXED_LEN ? left : FIXED_LEN);
        const uint64_t room = n_values - m.e0;
        m.avail = (uint32_t)(room < FIXED_LEN ? room : FIXED_LEN);
    }
    return m;
}
__device__ inline uint64_t wave_slot0(uint32_t run) {
    const uint32_t w = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    return ((uint64_t)blockIdx.x * (ENC_THREADS / 64) + w) * run;
}

// ---- pass A: section sizes ------------------------------------------------------------------------
template <int ETYPE>
__global__ __launch_bounds__(ENC_THREADS) void k_enc_sizes(const typename EncT<ETYPE>::U *__restrict__ values,
                                                            const uint64_t *__restrict__ elem_off,
                                                            const uint64_t *__restrict__ sec_base, uint64_t n_vectors,
                                                            const uint32_t *__restrict__ sec_vec, uint64_t max_slots,
                                                            uint32_t run, uint64_t *__restrict__ sec_size) {
    using U = typename EncT<ETYPE>::U;
    constexpr int W = EncT<ETYPE>::W;
    const int lane = threadIdx.x & 63;
    const uint64_t slot0 = wave_slot0(run);
    if (slot0 >= max_slots) return;
    const uint64_t slot = slot0 + lane;
    const RunMeta m = run_meta(elem_off, sec_base, sec_vec, n_vectors, slot, (uint32_t)lane < run);
    const int cnt = __popcll(__ballot(m.valid != 0));
    uint32_t my_size = 0; // slots past the last section get size 0
    U xa[4], xn[4];
    load_section<ETYPE>(values, lane_val(m.e0, 0), lane_val(m.valid, 0), lane_val(m.avail, 0), lane, xa);
    for (int r = 0; r < cnt; r++) {
        load_section<ETYPE>(values, lane_val(m.e0, r + 1), lane_val(m.valid, r + 1), lane_val(m.avail, r + 1), lane, xn); // lane cnt: nothing to load
        const SecPlan<U> p = plan_section<ETYPE>(xa, lane);
        uint32_t size;
        if (p.kind == K_NULL) size = 1;
        else if (p.kind == K_CONST) size = 1 + W;
        else {
            transform_section<ETYPE>(xa, p.kind, p.minv, lane);
            size = (p.kind == K_DELTA ? DELTA_HDR : 3) + pair_scan_total(octet_info<U>(xa, lane).size).total;
        }
        if (lane == r) my_size = size;
#pragma unroll
        for (int i = 0; i < 4; i++) xa[i] = xn[i];
    }
    if ((uint32_t)lane < run && slot < max_slots) sec_size[slot] = my_size;
}

// ---- headers + byte_off -----------------------------------------------------------------------------
__global__ void k_enc_headers(int32_t etype, const uint64_t *__restrict__ elem_off, const uint64_t *__restrict__ sec_base,
                              const uint64_t *__restrict__ sec_scan, uint64_t n, uint8_t *__restrict__ bytes,
                              uint64_t *__restrict__ byte_off, uint64_t *__restrict__ result) {
    uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v > n) return;
    // vector v starts after v headers and all sections of the vectors before it
    const uint64_t start = HEADER_BYTES * v + sec_scan[sec_base[v]];
    byte_off[v] = start;
    if (v == n) {
        result[0] = start; result[1] = 0; result[2] = 0; result[3] = sec_base[n];
        return;
    }
    const uint64_t end = HEADER_BYTES * (v + 1) + sec_scan[sec_base[v + 1]];
    const uint32_t num_bytes = (uint32_t)(end - start - 4);            // BinaryVector.num_bytes (src/vector.rs:54, :139)
    const uint32_t num_elements = (uint32_t)(elem_off[v + 1] - elem_off[v]);
    uint8_t *h = bytes + start;
    *(u32u *)(h + 0) = num_bytes;
    h[4] = 0x10;                                                        // VectorType::FixedSection256
    h[5] = (uint8_t)etype;                                              // VectorSubType
    h[6] = 0; h[7] = 0;
    *(u32u *)(h + 8) = num_elements;                                    // FixedSectStats (src/vector.rs:163-186)
    h[12] = 0; h[13] = 0;                                               // num_null_sections: 0 for encode_all
    h[14] = 0; h[15] = 0;
}

// ---- LDS image of a run of octets -----------------------------------------------------------------
// The image of a section (or of a block of a legacy stream) is built in LDS at the same 16-byte
// phase as its destination, so that the copy-out is aligned 16-byte traffic on both sides (a
// misaligned LDS read costs ~16x an aligned one on gfx950).

// ORs `bits` into the image at bit position `bitpos`; words that receive nothing are ORed with 0,
// which is cheaper than a divergent branch around an LDS instruction.  The image has 16 spare bytes
// behind the largest section for the words this touches past the end.
__device__ inline void image_or(uint32_t *buf, uint32_t bitpos, uint32_t bits) {
    const uint32_t d = bitpos >> 5, sh = bitpos & 31;
    const uint64_t w = (uint64_t)bits << sh;
    atomicOr(&buf[d], (uint32_t)w);
    atomicOr(&buf[d + 1], (uint32_t)(w >> 32));
}
__device__ inline void image_or(uint32_t *buf, uint32_t bitpos, uint64_t bits) {
    const uint32_t d = bitpos >> 5, sh = bitpos & 31;
    const uint64_t lo = bits << sh;
    const uint32_t hi = (uint32_t)((bits >> 32) >> 1 >> (31 - sh));
    atomicOr(&buf[d], (uint32_t)lo);
    atomicOr(&buf[d + 1], (uint32_t)(lo >> 32));
    atomicOr(&buf[d + 2], hi);
}

// Octets o = lane / 2 with o < noct: the even lane of a pair ORs the two header bytes, every lane ORs
// its fields into the zeroed image.  `first` = byte offset of octet 0 in the section; `phase` =
// position of section byte 0 in the image.  Values of octets >= noct must be zero.
template <class U>
__device__ inline void image_octets(const U x[4], int lane, uint32_t *buf, uint32_t phase, uint32_t first,
                                    uint32_t noct) {
    constexpr int W = sizeof(U);
    const OctInfo o = octet_info<U>(x, lane);
    const bool exists = (uint32_t)(lane >> 1) < noct, odd = lane & 1;
    const uint32_t osize = exists ? o.size : 0u;
    const PairScan ps = pair_scan_total(osize);
    const uint32_t off = phase + first + ps.incl - osize; // image byte of the octet's mask byte
    // mask byte, then ((n - 1) << 4) | t when any value is nonzero (n4 = 4n, t4 = 4t)
    const uint32_t word = o.mask ? (((o.n4 - 4) << 2) | (o.t4 >> 2)) << 8 : 0u;
    image_or(buf, off * 8, (!odd && exists) ? (o.mask | word) : 0u);
    // the lane's nonzero values are neighbours in the bitstream: they follow the nonzero values of
    // the even lane (odd lanes) or start the stream (even lanes)
    const uint32_t bit0 = (off + 2) * 8 + (odd ? (uint32_t)__popc(o.mask & 0xfu) * o.n4 : 0u);
    if constexpr (W == 4) {
        // two values at a time: at most 64 bits, the second slides down when the first is zero
        const uint32_t f0 = x[0] >> o.t4, f1 = x[1] >> o.t4, f2 = x[2] >> o.t4, f3 = x[3] >> o.t4;
        const uint64_t c01 = (o.m4 & 1) ? (uint64_t)f0 | ((uint64_t)f1 << o.n4) : (uint64_t)f1;
        const uint64_t c23 = (o.m4 & 4) ? (uint64_t)f2 | ((uint64_t)f3 << o.n4) : (uint64_t)f3;
        image_or(buf, bit0, c01);
        image_or(buf, bit0 + (uint32_t)__popc(o.m4 & 3u) * o.n4, c23);
    } else {
#pragma unroll
        for (int i = 0; i < 4; i++)
            image_or(buf, bit0 + (uint32_t)__popc(o.m4 & ((1u << i) - 1u)) * o.n4, (uint64_t)(x[i] >> o.t4));
    }
    // the ORs above are LDS operations of ONE wave: they complete in order, so the image is final
    // for this wave's reads without a barrier.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
}

// Whole 16-byte pieces as aligned dwordx4 by all lanes; the partial first and last piece one byte per
// lane (lanes 0..15 the first, 16..31 the last), so no lane loops over bytes.
__device__ inline void image_copy_out(const uint32_t *buf, uint32_t phase, uint32_t size, uint8_t *dst, int lane) {
    const uint8_t *img = (const uint8_t *)buf;
    uint8_t *gdst = dst - phase; // 16-byte aligned
    const uint32_t end = phase + size;
    const uint32_t npieces = (end + 15) >> 4;
    for (uint32_t d = lane; d < npieces; d += 64) {
        const uint32_t lo = 16 * d;
        if (lo >= phase && lo + 16 <= end) __builtin_nontemporal_store(((const u32x4_t *)buf)[d], (u32x4_t *)(gdst + lo));
    }
    if (lane < 32) {
        const bool last = lane >= 16;
        if (!last || npieces > 1) {
            const uint32_t b = 16 * (last ? npieces - 1 : 0u) + (lane & 15);
            const bool full = last ? (end & 15) == 0 : (phase == 0 && end >= 16);
            if (!full && b >= phase && b < end) gdst[b] = img[b];
        }
    }
}

// ---- pass B: emit ---------------------------------------------------------------------------------------
template <int W> struct EncBuf { static constexpr int DWORDS = ((SectLimits<W>::MAX_BYTES + 15 + 15) / 16 + 1) * 4; };

template <int ETYPE>
__global__ __launch_bounds__(ENC_THREADS) void k_enc_emit(const typename EncT<ETYPE>::U *__restrict__ values,
                                                           const uint64_t *__restrict__ elem_off,
                                                           const uint64_t *__restrict__ sec_base, uint64_t n_vectors,
                                                           const uint32_t *__restrict__ sec_vec, uint64_t max_slots,
                                                           uint32_t run, const uint64_t *__restrict__ sec_scan,
                                                           uint8_t *__restrict__ bytes) {
    using U = typename EncT<ETYPE>::U;
    constexpr int W = EncT<ETYPE>::W;
    __shared__ __attribute__((aligned(16))) uint32_t s_buf[ENC_THREADS / 64][EncBuf<W>::DWORDS];
    const int lane = threadIdx.x & 63;
    uint32_t *buf = s_buf[threadIdx.x >> 6];
    const uint64_t slot0 = wave_slot0(run);
    if (slot0 >= max_slots) return;
    const uint64_t slot = slot0 + lane;
    const RunMeta m = run_meta(elem_off, sec_base, sec_vec, n_vectors, slot, (uint32_t)lane < run);
    uint64_t m_dst = 0;
    uint32_t m_size = 0;
    if (m.valid) {
        const uint64_t at = sec_scan[slot];
        m_dst = HEADER_BYTES * ((uint64_t)sec_vec[slot] + 1) + at;
        m_size = (uint32_t)(sec_scan[slot + 1] - at);
    }
    // A section of one byte is a Null section (every other kind takes three bytes or more): its values are all
    // zero and are not read again.  Every lane writes the byte of its own Null section here
    // (NullFixedSect::write, src/section.rs:440-443), and the loop below takes the sections with octets only.
    // Likewise a section of 1 + W bytes is a Const section (ConstFixedSect::write, src/section.rs:795-803: the type
    // byte and values[0]; sections with octets take more than 30 bytes), for which one value is read.
    const bool own_null = m.valid != 0 && m_size == 1, own_const = m.valid != 0 && m_size == 1 + W;
    if (own_null) bytes[m_dst] = (uint8_t)K_NULL;
    if (own_const) {
        const U first = values[m.e0];
        bytes[m_dst] = (uint8_t)K_CONST;
        if constexpr (W == 4) *(u32u *)(bytes + m_dst + 1) = first; else *(u64u *)(bytes + m_dst + 1) = first;
    }
    uint64_t todo = __ballot(m.valid != 0 && !own_null && !own_const);
    if (todo == 0) return;
    int r = __builtin_ctzll(todo);
    todo &= todo - 1;
    U x[4], xn[4];
    load_section<ETYPE>(values, lane_val(m.e0, r), lane_val(m.valid, r), lane_val(m.avail, r), lane, x);
    for (;;) {
        // the next section that has bytes of its own; behind the last one a descriptor of no elements (a no-op load)
        const bool more = todo != 0;
        const int rn = more ? __builtin_ctzll(todo) : r;
        todo &= todo - 1;
        load_section<ETYPE>(values, lane_val(m.e0, rn), lane_val(m.valid, rn), more ? lane_val(m.avail, rn) : 0u, lane, xn);
        const uint64_t dst_off = lane_val(m_dst, r);
        const uint32_t size = lane_val(m_size, r);
        uint8_t *dst = bytes + dst_off;
        const SecPlan<U> p = plan_section<ETYPE>(x, lane);
        if (p.kind == K_NULL) { // NullFixedSect::write (src/section.rs:440-443)
            if (lane == 0) dst[0] = (uint8_t)K_NULL;
        } else if (p.kind == K_CONST) { // ConstFixedSect::write (src/section.rs:795-803): values[0], not min
            if (lane == 0) {
                dst[0] = (uint8_t)K_CONST;
                if constexpr (W == 4) *(u32u *)(dst + 1) = p.first; else *(u64u *)(dst + 1) = p.first;
            }
        } else {
            const uint32_t phase = (uint32_t)(dst_off & 15);   // bytes is 16-byte aligned: the phase of dst itself
            uint8_t *bufb = (uint8_t *)buf + phase;            // section byte 0
            const uint32_t npieces = (phase + size + 15) >> 4; // 16-byte pieces of the image
            // the copy-out reads of the previous section of this wave were issued before these writes
            // and LDS serves one wave in order
            for (uint32_t d = lane; d < npieces; d += 64) ((uint4 *)buf)[d] = make_uint4(0, 0, 0, 0);
            const uint32_t hdr = p.kind == K_DELTA ? DELTA_HDR : 3;
            if (lane == 0) {
                bufb[0] = (uint8_t)p.kind;
                // NP / Delta: bytes after the header; Xor: total including the 3 header bytes
                // (src/section.rs:579-581, :659-661, :745-748)
                const uint32_t len16 = p.kind == K_XOR ? size : size - hdr;
                bufb[1] = (uint8_t)len16; bufb[2] = (uint8_t)(len16 >> 8);
                if (p.kind == K_DELTA) {
                    bufb[3] = (uint8_t)bitlen_u((U)(p.maxv - p.minv));          // src/section.rs:663
                    for (int i = 0; i < W; i++) bufb[4 + i] = (uint8_t)(p.minv >> (8 * i)); // base; bytes 4+W..11 stay 0
                }
            }
            transform_section<ETYPE>(x, p.kind, p.minv, lane);
            image_octets<U>(x, lane, buf, phase, hdr, 32);
            // The next section's values are waited for here, ahead of this section's stores: a wait for
            // loads also drains every store in flight (the counter cannot tell them apart), and at this
            // point the only stores in flight are the previous section's, a whole section old.
#pragma unroll
            for (int i = 0; i < 4; i++) asm volatile("" ::"v"(xn[i]));
            image_copy_out(buf, phase, size, dst, lane);
        }
#pragma unroll
        for (int i = 0; i < 4; i++) x[i] = xn[i];
        if (!more) break;
        r = rn;
    }
}

// ---- one pass: sizes, offsets and bytes of a tile of sections in one kernel -----------------------------------------
// The two passes above read every value twice, because a section's place in the blob depends on the sizes of all
// sections before it.  Here a workgroup owns a tile of consecutive section slots and reads its values once:
//   1. every wave loads its share of the tile's sections, plans and transforms them, computes their sizes and parks
//      the transformed values in LDS (a section is 1 or 2 KiB there; kind, base and size stay in the lane that has
//      the section's number);
//   2. the tile's total goes into the tile's state word (flag | value in one 64-bit store), and the first wave looks
//      back over the state words of the tiles in front of it: it adds up their totals until it meets a tile whose
//      offset is already known, then publishes its own (decoupled look-back; tiles take their numbers from a counter,
//      so every tile in front of a running tile is running or done and the wait always ends);
//   3. every wave builds the bytes of its sections from the parked values and copies them out.
// The scan of the section sizes falls out of the same pass (sec_scan, which the header kernel needs).
//
// Built with -DCVEC_ENC_ONE_PASS only.  It is bit-exact (the whole GPU suite passes on it) and moves a third less
// memory, and it is slower than the two passes: 0.98 ms against 0.19 + 0.45 ms per GiB of incompressible f32, 0.85
// against 0.18 + 0.30 ms on dense u32.  The work on a section is a chain of dependent cross-lane steps (the min / max
// reduction, the octet scan, the ORs into the image) that takes a wave about 1.5 us whatever else the CU does, so the
// section rate is set by the number of resident waves: the two kernels run 32 waves per CU, this one 16, because it
// holds the state of both passes in registers (88 to 99 VGPRs: one workgroup of 16 waves per CU) next to 64 KiB of
// parked values.  Forced to 64 VGPRs it spills and takes 1.29 ms; with 4 waves per workgroup (two workgroups per
// CU) 1.0 ms.  The look-back itself costs little (the kernel without it, with garbage offsets, takes as long).
// A tile is 64 KiB of values (64 sections of 32-bit elements, 32 of 64-bit ones) in the hands of 16 waves: the
// section work is chains of dependent cross-lane operations, which need many waves per SIMD to fill the issue slots
// (two workgroups of 16 waves per CU = 8 per SIMD; with 4 waves per workgroup the same kernel ran 40 % slower than
// the two passes it replaces).
constexpr int ENC_TILE_THREADS = 1024;
template <int W> struct EncTile {
    static constexpr int WAVES = ENC_TILE_THREADS / 64;
    static constexpr int PER_WAVE = W == 4 ? 4 : 2;
    static constexpr int SECTIONS = WAVES * PER_WAVE;
};
constexpr uint64_t TILE_FLAG_TOTAL = 1ull << 62, TILE_FLAG_OFFSET = 2ull << 62, TILE_VALUE = (1ull << 62) - 1;

__device__ inline uint64_t wave_sum_u64(uint64_t x) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) x += __shfl_xor(x, d, 64);
    return x;
}

template <int ETYPE>
__global__ __launch_bounds__(ENC_TILE_THREADS) void k_enc_tile(const typename EncT<ETYPE>::U *__restrict__ values,
                                                           const uint64_t *__restrict__ elem_off,
                                                           const uint64_t *__restrict__ sec_base, uint64_t n_vectors,
                                                           const uint32_t *__restrict__ sec_vec, uint64_t max_slots,
                                                           uint32_t n_tiles, uint64_t *__restrict__ tile_state,
                                                           uint32_t *__restrict__ ticket, uint64_t *__restrict__ sec_scan,
                                                           uint8_t *__restrict__ bytes) {
    using U = typename EncT<ETYPE>::U;
    constexpr int W = EncT<ETYPE>::W;
    constexpr int PER_WAVE = EncTile<W>::PER_WAVE, WAVES = EncTile<W>::WAVES;
    typedef uint32_t __attribute__((__vector_size__(16))) u32x4;
    // a slot per section: first the parked values (1 or 2 KiB), later, in the same place, the image of its bytes
    static_assert(EncBuf<W>::DWORDS * 4 >= FIXED_LEN * W, "a slot holds a section's values");
    __shared__ __attribute__((aligned(16))) uint32_t s_slot[WAVES][PER_WAVE][EncBuf<W>::DWORDS];
    // what the waves tell each other, in two copies: a wave can be one barrier ahead of another
    __shared__ uint64_t s_wave_total[2][WAVES];
    __shared__ uint64_t s_tile_base[2];
    __shared__ uint32_t s_next[2];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

    // The persistent workgroup takes its tiles from a counter, the next one while it works on the current one, and
    // has the next tile's values in flight (in registers) while it waits for its offset and writes the current
    // tile's bytes: a workgroup holds 64 KiB of LDS, so a CU runs two of them and every wave has to keep many
    // loads in flight.
    U x[PER_WAVE][4];
    RunMeta m;
    auto fetch = [&](uint32_t tile) {
        const uint64_t slot = ((uint64_t)tile * WAVES + wave) * PER_WAVE + lane;
        m = run_meta(elem_off, sec_base, sec_vec, n_vectors, slot, tile < n_tiles && lane < PER_WAVE && slot < max_slots);
#pragma unroll
        for (int r = 0; r < PER_WAVE; r++)
            load_section<ETYPE>(values, lane_val(m.e0, r), lane_val(m.valid, r), lane_val(m.avail, r), lane, x[r]);
    };
    if (threadIdx.x == 0) s_next[0] = atomicAdd(ticket, 1u);
    __syncthreads();
    uint32_t tile = s_next[0];
    fetch(tile);
    for (int it = 0; tile < n_tiles; it ^= 1) {
        if (threadIdx.x == 0) s_next[it ^ 1] = atomicAdd(ticket, 1u);
        const uint64_t slot = ((uint64_t)tile * WAVES + wave) * PER_WAVE + lane;
        const int cnt = __popcll(__ballot(m.valid != 0));
        const uint32_t my_vec = m.vec;
        // ---- 1: sizes; the transformed values go to LDS, what else the emit needs stays in lane r ---------------------
        uint32_t my_size = 0, my_kind = K_NULL;
        U my_min = 0, my_max = 0, my_first = 0;
#pragma unroll
        for (int r = 0; r < PER_WAVE; r++) {
            if (r < cnt) {
                const SecPlan<U> p = plan_section<ETYPE>(x[r], lane);
                uint32_t size;
                if (p.kind == K_NULL) size = 1;
                else if (p.kind == K_CONST) size = 1 + W;
                else {
                    transform_section<ETYPE>(x[r], p.kind, p.minv, lane);
                    size = (p.kind == K_DELTA ? DELTA_HDR : 3) + pair_scan_total(octet_info<U>(x[r], lane).size).total;
                    u32x4 *dst = (u32x4 *)s_slot[wave][r];
                    if constexpr (W == 4) {
                        dst[lane] = u32x4{x[r][0], x[r][1], x[r][2], x[r][3]};
                    } else {
                        dst[lane] = u32x4{(uint32_t)x[r][0], (uint32_t)(x[r][0] >> 32), (uint32_t)x[r][1], (uint32_t)(x[r][1] >> 32)};
                        dst[64 + lane] = u32x4{(uint32_t)x[r][2], (uint32_t)(x[r][2] >> 32), (uint32_t)x[r][3], (uint32_t)(x[r][3] >> 32)};
                    }
                }
                if (lane == r) { my_size = size; my_kind = p.kind; my_min = p.minv; my_max = p.maxv; my_first = p.first; }
            }
        }
        // ---- 2: the tile's place in the blob --------------------------------------------------------------------------
        // (offsets count section bytes only; the 16-byte vector headers are added by vector number)
        uint64_t incl = my_size; // inclusive scan over the wave's lanes (sizes sit in lanes 0 .. PER_WAVE - 1, zeros above)
#pragma unroll
        for (int d = 1; d < PER_WAVE; d <<= 1) {
            const uint64_t y = __shfl_up(incl, d, 64);
            if (lane >= d) incl += y;
        }
        if (lane == PER_WAVE - 1) s_wave_total[it][wave] = incl;
        __syncthreads();
        // the next tile's values: requested now, used in the next pass of the loop
        const uint32_t next = s_next[it ^ 1];
        fetch(next);
        if (wave == 0) {
            uint64_t tile_total = 0;
#pragma unroll
            for (int w = 0; w < WAVES; w++) tile_total += s_wave_total[it][w];
            uint64_t before = 0;
            if (tile == 0) {
                if (lane == 0) __hip_atomic_store(&tile_state[0], TILE_FLAG_OFFSET | tile_total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                if (lane == 0) __hip_atomic_store(&tile_state[tile], TILE_FLAG_TOTAL | tile_total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                // lane l looks at the l-th tile in front of the window's start; the virtual tile in front of tile 0 has offset 0
                int64_t at = (int64_t)tile - 1;
                for (;;) {
                    const int64_t t = at - lane;
                    const uint64_t st = t >= 0 ? __hip_atomic_load(&tile_state[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : TILE_FLAG_OFFSET;
                    const unsigned long long known = __ballot((st >> 62) == 2), blank = __ballot((st >> 62) == 0);
                    const int first_known = known ? (int)__builtin_ctzll(known) : 64;
                    // every tile nearer than the first one with a known offset has to have its total out
                    const unsigned long long nearer = first_known >= 64 ? ~0ull : ((1ull << first_known) - 1ull);
                    if (blank & nearer) { __builtin_amdgcn_s_sleep(2); continue; }
                    const uint64_t take = lane <= first_known ? (st & TILE_VALUE) : 0ull;
                    before += wave_sum_u64(take);
                    if (first_known < 64) break;
                    at -= 64;
                }
                if (lane == 0) __hip_atomic_store(&tile_state[tile], TILE_FLAG_OFFSET | (before + tile_total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (lane == 0) s_tile_base[it] = before;
        }
        __syncthreads();
        uint64_t my_off = s_tile_base[it] + incl - my_size;
#pragma unroll
        for (int w = 0; w < WAVES; w++) my_off += w < wave ? s_wave_total[it][w] : 0ull;
        if (lane < PER_WAVE && slot < max_slots) {
            sec_scan[slot] = my_off;
            if (slot == max_slots - 1) sec_scan[max_slots] = my_off + my_size;
        }
        const uint64_t my_dst = HEADER_BYTES * ((uint64_t)my_vec + 1) + my_off;
        // ---- 3: bytes -------------------------------------------------------------------------------------------------
        for (int r = 0; r < cnt; r++) {
            const uint64_t dst_off = lane_val(my_dst, r);
            const uint32_t size = lane_val(my_size, r), kind = lane_val(my_kind, r);
            uint8_t *dst = bytes + dst_off;
            if (kind == K_NULL) { // NullFixedSect::write (src/section.rs:440-443)
                if (lane == 0) dst[0] = (uint8_t)K_NULL;
            } else if (kind == K_CONST) { // ConstFixedSect::write (src/section.rs:795-803): values[0], not min
                const U first = lane_val(my_first, r);
                if (lane == 0) {
                    dst[0] = (uint8_t)K_CONST;
                    if constexpr (W == 4) *(u32u *)(dst + 1) = first; else *(u64u *)(dst + 1) = first;
                }
            } else {
                const U minv = lane_val(my_min, r), maxv = lane_val(my_max, r);
                U y[4];
                uint32_t *buf = s_slot[wave][r];
                const u32x4 *src = (const u32x4 *)buf;
                if constexpr (W == 4) {
                    const u32x4 v = src[lane];
#pragma unroll
                    for (int i = 0; i < 4; i++) y[i] = v[i];
                } else {
                    const u32x4 a = src[lane], b = src[64 + lane];
                    y[0] = (uint64_t)a[0] | ((uint64_t)a[1] << 32); y[1] = (uint64_t)a[2] | ((uint64_t)a[3] << 32);
                    y[2] = (uint64_t)b[0] | ((uint64_t)b[1] << 32); y[3] = (uint64_t)b[2] | ((uint64_t)b[3] << 32);
                }
                // (the values are in registers now: LDS serves a wave in order, so the zeroes below land behind the reads)
                const uint32_t phase = (uint32_t)(dst_off & 15);
                uint8_t *bufb = (uint8_t *)buf + phase;
                const uint32_t npieces = (phase + size + 15) >> 4;
                for (uint32_t d = lane; d < npieces; d += 64) ((uint4 *)buf)[d] = make_uint4(0, 0, 0, 0);
                const uint32_t hdr = kind == K_DELTA ? DELTA_HDR : 3;
                if (lane == 0) {
                    bufb[0] = (uint8_t)kind;
                    const uint32_t len16 = kind == K_XOR ? size : size - hdr; // src/section.rs:579-581, :659-661, :745-748
                    bufb[1] = (uint8_t)len16; bufb[2] = (uint8_t)(len16 >> 8);
                    if (kind == K_DELTA) {
                        bufb[3] = (uint8_t)bitlen_u((U)(maxv - minv));                    // src/section.rs:663
                        for (int i = 0; i < W; i++) bufb[4 + i] = (uint8_t)(minv >> (8 * i)); // base; bytes 4+W..11 stay 0
                    }
                }
                image_octets<U>(y, lane, buf, phase, hdr, 32);
                image_copy_out(buf, phase, size, dst, lane);
            }
        }
        tile = next;
    }
}

// ---- legacy stream codecs: pack_u64_delta / pack_f64_xor (reference src/nibblepacking.rs:15-45) ----
// A stream is cut into blocks of 32 octets (256 packed values), one wave per block, exactly like a
// section without header.  Packed value i of a stream: delta codec = v[i] - v[i-1], clipped at 0
// (saturating_sub, :18), v[-1] = 0; XOR codec = bits(v[i+1]) ^ bits(v[i]) (the first value is
// stored raw in front, :31-44).  pack_u64 zero-pads the last octet (:85-91).
template <int KIND>
__device__ inline void load_stream_block(const uint64_t *__restrict__ v, uint64_t npacked, uint64_t i0, int lane,
                                         uint64_t x[4]) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const uint64_t idx = i0 + 4 * lane + i;
        uint64_t y = 0;
        if (idx < npacked) {
            if (KIND == CVEC_STREAM_F64_XOR) {
                y = v[idx + 1] ^ v[idx];
            } else if (KIND == CVEC_STREAM_U64_PLAIN) {
                y = v[idx];
            } else {
                const uint64_t cur = v[idx], prev = idx ? v[idx - 1] : 0;
                y = cur >= prev ? cur - prev : 0;
            }
        }
        x[i] = y;
    }
}

__device__ inline uint64_t stream_packed_values(int kind, uint64_t n) {
    return kind == CVEC_STREAM_F64_XOR ? (n ? n - 1 : 0) : n;
}

template <int KIND>
__global__ __launch_bounds__(ENC_THREADS) void k_stream_sizes(const uint64_t *__restrict__ values,
                                                               const uint64_t *__restrict__ elem_off,
                                                               const uint64_t *__restrict__ blk_base,
                                                               const uint32_t *__restrict__ blk_stream, uint64_t n,
                                                               uint64_t max_blocks, uint64_t *__restrict__ blk_size) {
    const int lane = threadIdx.x & 63;
    const uint64_t blk = (uint64_t)blockIdx.x * (ENC_THREADS / 64) + (threadIdx.x >> 6);
    if (blk >= max_blocks) return;
    if (blk >= blk_base[n]) { if (lane == 0) blk_size[blk] = 0; return; }
    const uint64_t s = blk_stream[blk];
    const uint64_t npacked = stream_packed_values(KIND, elem_off[s + 1] - elem_off[s]);
    const uint64_t i0 = (blk - blk_base[s]) * FIXED_LEN;
    const uint32_t noct = (uint32_t)(((npacked - i0 < FIXED_LEN ? npacked - i0 : FIXED_LEN) + 7) / 8);
    uint64_t x[4];
    load_stream_block<KIND>(values + elem_off[s], npacked, i0, lane, x);
    const uint32_t sz = (uint32_t)(lane >> 1) < noct ? octet_info<uint64_t>(x, lane).size : 0u;
    const uint32_t total = pair_scan_total(sz).total;
    if (lane == 0) blk_size[blk] = total;
}

__global__ void k_stream_ext(int kind, const uint64_t *__restrict__ elem_off, uint64_t n, uint64_t *__restrict__ ext) {
    const uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    ext[s] = (kind == CVEC_STREAM_F64_XOR && elem_off[s + 1] > elem_off[s]) ? 8 : 0; // the raw first value
}

__global__ void k_stream_offsets(int kind, const uint64_t *__restrict__ values, const uint64_t *__restrict__ elem_off,
                                 const uint64_t *__restrict__ blk_base, const uint64_t *__restrict__ blk_scan,
                                 const uint64_t *__restrict__ ext_scan, uint64_t n, uint8_t *__restrict__ bytes,
                                 uint64_t *__restrict__ byte_off, int32_t *__restrict__ status,
                                 uint64_t *__restrict__ result) {
    const uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s > n) return;
    const uint64_t start = blk_scan[blk_base[s]] + ext_scan[s];
    byte_off[s] = start;
    if (s == n) { result[0] = start; return; }
    int st = CVEC_OK;
    if (kind == CVEC_STREAM_F64_XOR) {
        if (elem_off[s + 1] == elem_off[s]) st = CVEC_INPUT_TOO_SHORT; // pack_f64_xor needs one value (:37)
        else *(u64u *)(bytes + start) = values[elem_off[s]];
    }
    if (status) status[s] = st;
    if (st) { atomicAdd((unsigned long long *)&result[1], 1ull); atomicMax((unsigned long long *)&result[2], (unsigned long long)st); }
}

template <int KIND>
__global__ __launch_bounds__(ENC_THREADS) void k_stream_emit(const uint64_t *__restrict__ values,
                                                              const uint64_t *__restrict__ elem_off,
                                                              const uint64_t *__restrict__ blk_base,
                                                              const uint32_t *__restrict__ blk_stream, uint64_t n,
                                                              uint64_t max_blocks, const uint64_t *__restrict__ blk_scan,
                                                              const uint64_t *__restrict__ byte_off,
                                                              uint8_t *__restrict__ bytes) {
    __shared__ __attribute__((aligned(16))) uint32_t s_buf[ENC_THREADS / 64][EncBuf<8>::DWORDS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t blk = (uint64_t)blockIdx.x * (ENC_THREADS / 64) + wave;
    if (blk >= max_blocks || blk >= blk_base[n]) return;
    const uint64_t s = blk_stream[blk];
    const uint64_t npacked = stream_packed_values(KIND, elem_off[s + 1] - elem_off[s]);
    const uint64_t i0 = (blk - blk_base[s]) * FIXED_LEN;
    const uint32_t noct = (uint32_t)(((npacked - i0 < FIXED_LEN ? npacked - i0 : FIXED_LEN) + 7) / 8);
    const uint32_t size = (uint32_t)(blk_scan[blk + 1] - blk_scan[blk]);
    const uint64_t dst_off = byte_off[s] + (KIND == CVEC_STREAM_F64_XOR ? 8 : 0) + (blk_scan[blk] - blk_scan[blk_base[s]]);
    uint64_t x[4];
    load_stream_block<KIND>(values + elem_off[s], npacked, i0, lane, x);
    uint32_t *buf = s_buf[wave];
    const uint32_t phase = (uint32_t)(dst_off & 15);
    const uint32_t npieces = (phase + size + 15) >> 4;
    for (uint32_t d = lane; d < npieces; d += 64) ((uint4 *)buf)[d] = make_uint4(0, 0, 0, 0);
    image_octets<uint64_t>(x, lane, buf, phase, 0, noct);
    image_copy_out(buf, phase, size, bytes + dst_off, lane);
}

struct StreamPackWs {
    uint64_t *nblk, *blk_base, *ext, *ext_scan, *scan_scratch, *blk_size, *blk_scan, *scan_scratch2;
    uint32_t *blk_stream;
    size_t total;
};
static StreamPackWs carve_stream_pack(void *base, uint64_t n, uint64_t max_blocks) {
    StreamPackWs w;
    uint8_t *b = (uint8_t *)base;
    size_t off = 0;
    auto take = [&](size_t bytes) { uint8_t *p = b + off; off += (bytes + 255) / 256 * 256; return p; };
    w.nblk = (uint64_t *)take((n + 1) * 8);
    w.blk_base = (uint64_t *)take((n + 1) * 8);
    w.ext = (uint64_t *)take((n + 1) * 8);
    w.ext_scan = (uint64_t *)take((n + 1) * 8);
    w.scan_scratch = (uint64_t *)take(scan_scratch_words(n) * 8);
    w.blk_size = (uint64_t *)take((max_blocks + 1) * 8);
    w.blk_scan = (uint64_t *)take((max_blocks + 1) * 8);
    w.scan_scratch2 = (uint64_t *)take(scan_scratch_words(max_blocks) * 8);
    w.blk_stream = (uint32_t *)take((max_blocks + 1) * 4);
    w.total = off;
    return w;
}
size_t stream_pack_workspace_bytes(uint64_t n_streams, uint64_t n_values_total) {
    return carve_stream_pack(nullptr, n_streams, n_values_total / FIXED_LEN + n_streams).total;
}

__global__ void k_stream_nblk_pack(int kind, const uint64_t *__restrict__ elem_off, uint64_t n, uint64_t *__restrict__ nblk) {
    const uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    nblk[s] = (stream_packed_values(kind, elem_off[s + 1] - elem_off[s]) + FIXED_LEN - 1) / FIXED_LEN;
}

template <int KIND>
static hipError_t stream_pack_t(const uint64_t *d_values, const uint64_t *d_elem_off, uint64_t n, uint64_t n_values_total,
                                uint8_t *d_bytes, uint64_t *d_byte_off, int32_t *d_status, uint64_t *d_result,
                                void *d_ws, hipStream_t stream) {
    const uint64_t maxb = n_values_total / FIXED_LEN + n;
    StreamPackWs w = carve_stream_pack(d_ws, n, maxb);
    hipError_t e = hipMemsetAsync(d_result, 0, CVEC_RESULT_WORDS * sizeof(uint64_t), stream);
    if (e != hipSuccess) return e;
    const uint32_t g = (uint32_t)((n + 256) / 256);
    if (n) {
        k_stream_nblk_pack<<<g, 256, 0, stream>>>(KIND, d_elem_off, n, w.nblk);
        k_stream_ext<<<g, 256, 0, stream>>>(KIND, d_elem_off, n, w.ext);
    }
    e = launch_exclusive_scan(w.nblk, n, w.blk_base, w.scan_scratch, stream);
    if (e != hipSuccess) return e;
    e = launch_exclusive_scan(w.ext, n, w.ext_scan, w.scan_scratch, stream);
    if (e != hipSuccess) return e;
    const uint32_t gb = (uint32_t)((maxb + ENC_THREADS / 64 - 1) / (ENC_THREADS / 64));
    if (n) k_enc_map<<<g, 256, 0, stream>>>(w.blk_base, n, w.blk_stream); // block -> stream
    if (maxb) k_stream_sizes<KIND><<<gb, ENC_THREADS, 0, stream>>>(d_values, d_elem_off, w.blk_base, w.blk_stream, n, maxb, w.blk_size);
    e = launch_exclusive_scan(w.blk_size, maxb, w.blk_scan, w.scan_scratch2, stream);
    if (e != hipSuccess) return e;
    k_stream_offsets<<<g, 256, 0, stream>>>(KIND, d_values, d_elem_off, w.blk_base, w.blk_scan, w.ext_scan, n, d_bytes,
                                            d_byte_off, d_status, d_result);
    if (maxb) k_stream_emit<KIND><<<gb, ENC_THREADS, 0, stream>>>(d_values, d_elem_off, w.blk_base, w.blk_stream, n, maxb, w.blk_scan,
                                                                  d_byte_off, d_bytes);
    return hipGetLastError();
}

hipError_t launch_stream_pack(int kind, const uint64_t *d_values, const uint64_t *d_elem_off, uint64_t n,
                              uint64_t n_values_total, uint8_t *d_bytes, uint64_t *d_byte_off, int32_t *d_status,
                              uint64_t *d_result, void *d_ws, hipStream_t stream) {
    if (kind == CVEC_STREAM_F64_XOR)
        return stream_pack_t<CVEC_STREAM_F64_XOR>(d_values, d_elem_off, n, n_values_total, d_bytes, d_byte_off, d_status,
                                                  d_result, d_ws, stream);
    if (kind == CVEC_STREAM_U64_PLAIN)
        return stream_pack_t<CVEC_STREAM_U64_PLAIN>(d_values, d_elem_off, n, n_values_total, d_bytes, d_byte_off, d_status,
                                                    d_result, d_ws, stream);
    return stream_pack_t<CVEC_STREAM_U64_DELTA>(d_values, d_elem_off, n, n_values_total, d_bytes, d_byte_off, d_status,
                                                d_result, d_ws, stream);
}

// ---- histograms: DeltaDiffPackSink (reference src/histogram.rs:92-171) -----------------------------
// A histogram arrives as a NibblePack stream of num_buckets values (bucket-to-bucket deltas).  The
// sink re-packs it as the difference to the previous histogram of the same series, bucket by bucket;
// a bucket that fell below its predecessor is passed through unchanged and flags the histogram
// (:138-143).  Because last_hist_deltas is simply the previous histogram's values (:146-148) there is
// no chain: every histogram is independent given its predecessor's unpacked values.
// `raw` holds the unpacked values in blocks of 256: histogram h, value i at raw[(h * in_blocks) * 256 + i].
// finish() packs the last octet a second time (:117-122: `i` is already past it); that octet is part
// of the reference's output and is reproduced: value index i >= 8 * noct stands for i - 8.

struct HistShape {
    uint32_t num_buckets;
    uint32_t noct;       // octets of one histogram: ceil(num_buckets / 8)
    uint32_t in_blocks;  // 256-value blocks per unpacked histogram
    uint32_t out_blocks; // 32-octet blocks per re-packed histogram (noct + 1 octets)
};

// One lane per series: a histogram is usable when it and every earlier histogram of its series
// unpacked (the reference's sink state is undefined after a failed unpack); the first failure's code
// is handed down the series.  Also marks the first histogram of each series.
__global__ void k_hist_status(const uint64_t *__restrict__ series_off, uint64_t n_series, HistShape sh,
                              const uint32_t *__restrict__ sec_res, int32_t *__restrict__ status,
                              uint32_t *__restrict__ first, uint32_t *__restrict__ dropped) {
    const uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_series) return;
    int carried = CVEC_OK;
    for (uint64_t h = series_off[s]; h < series_off[s + 1]; h++) {
        int st = status[h];
        for (uint32_t b = 0; b < sh.in_blocks && st == CVEC_OK; b++) {
            const uint32_t r = sec_res[h * sh.in_blocks + b];
            if (r >> 16) st = (int)(r >> 16);
        }
        if (carried == CVEC_OK) carried = st;
        status[h] = carried;
        first[h] = h == series_off[s];
        dropped[h] = 0;
    }
}

// Values idx0 + 4 * lane + 0..3 of re-packed histogram h; `low` collects "a bucket fell".
__device__ inline void load_hist_block(const uint64_t *__restrict__ raw, uint64_t h, bool is_first, bool ok, HistShape sh,
                                       uint32_t idx0, int lane, uint64_t x[4], bool &low) {
    const uint64_t *cur = raw + h * sh.in_blocks * FIXED_LEN;
    const uint64_t *prev = cur - (uint64_t)sh.in_blocks * FIXED_LEN; // only read when !is_first
#pragma unroll
    for (int i = 0; i < 4; i++) {
        uint32_t idx = idx0 + 4 * lane + i;
        uint64_t y = 0;
        if (ok && idx < 8 * (sh.noct + 1)) {
            if (idx >= 8 * sh.noct) idx -= 8; // the octet that finish() repeats
            if (idx < sh.num_buckets) {
                const uint64_t c = cur[idx], p = is_first ? 0 : prev[idx];
                if (c < p) { low = true; y = c; } else y = c - p;
            }
        }
        x[i] = y;
    }
}

__global__ __launch_bounds__(ENC_THREADS) void k_hist_sizes(const uint64_t *__restrict__ raw, uint64_t n_hist, HistShape sh,
                                                             const int32_t *__restrict__ status,
                                                             const uint32_t *__restrict__ first,
                                                             uint64_t *__restrict__ blk_size,
                                                             uint32_t *__restrict__ dropped) {
    const int lane = threadIdx.x & 63;
    const uint64_t blk = (uint64_t)blockIdx.x * (ENC_THREADS / 64) + (threadIdx.x >> 6);
    if (blk >= n_hist * sh.out_blocks) return;
    const uint64_t h = blk / sh.out_blocks;
    const uint32_t j = (uint32_t)(blk % sh.out_blocks);
    const bool ok = status[h] == CVEC_OK;
    const uint32_t left = sh.noct + 1 - 32 * j, noct = ok ? (left < 32 ? left : 32) : 0;
    uint64_t x[4];
    bool low = false;
    load_hist_block(raw, h, first[h] != 0, ok, sh, j * FIXED_LEN, lane, x, low);
    const uint32_t sz = (uint32_t)(lane >> 1) < noct ? octet_info<uint64_t>(x, lane).size : 0u;
    const uint32_t total = pair_scan_total(sz).total;
    if (lane == 0) blk_size[blk] = total;
    if (__ballot(low) && lane == 0) atomicOr(&dropped[h], 1u);
}

__global__ void k_hist_offsets(uint64_t n_hist, HistShape sh, const uint64_t *__restrict__ blk_scan,
                               const int32_t *__restrict__ status, uint64_t *__restrict__ out_off,
                               uint64_t *__restrict__ result) {
    const uint64_t h = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (h > n_hist) return;
    out_off[h] = blk_scan[h * sh.out_blocks];
    if (h == n_hist) { result[0] = out_off[h]; return; }
    const int st = status[h];
    if (st) { atomicAdd((unsigned long long *)&result[1], 1ull); atomicMax((unsigned long long *)&result[2], (unsigned long long)st); }
}

__global__ __launch_bounds__(ENC_THREADS) void k_hist_emit(const uint64_t *__restrict__ raw, uint64_t n_hist, HistShape sh,
                                                            const int32_t *__restrict__ status,
                                                            const uint32_t *__restrict__ first,
                                                            const uint64_t *__restrict__ blk_scan,
                                                            uint8_t *__restrict__ bytes) {
    __shared__ __attribute__((aligned(16))) uint32_t s_buf[ENC_THREADS / 64][EncBuf<8>::DWORDS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t blk = (uint64_t)blockIdx.x * (ENC_THREADS / 64) + wave;
    if (blk >= n_hist * sh.out_blocks) return;
    const uint64_t h = blk / sh.out_blocks;
    const uint32_t j = (uint32_t)(blk % sh.out_blocks);
    if (status[h] != CVEC_OK) return;
    const uint32_t left = sh.noct + 1 - 32 * j, noct = left < 32 ? left : 32;
    const uint64_t dst_off = blk_scan[blk];
    const uint32_t size = (uint32_t)(blk_scan[blk + 1] - dst_off);
    uint64_t x[4];
    bool low = false;
    load_hist_block(raw, h, first[h] != 0, true, sh, j * FIXED_LEN, lane, x, low);
    uint32_t *buf = s_buf[wave];
    const uint32_t phase = (uint32_t)(dst_off & 15);
    const uint32_t npieces = (phase + size + 15) >> 4;
    for (uint32_t d = lane; d < npieces; d += 64) ((uint4 *)buf)[d] = make_uint4(0, 0, 0, 0);
    image_octets<uint64_t>(x, lane, buf, phase, 0, noct);
    image_copy_out(buf, phase, size, bytes + dst_off, lane);
}

HistShape hist_shape(uint32_t num_buckets) {
    HistShape sh;
    sh.num_buckets = num_buckets;
    sh.noct = (num_buckets + 7) / 8;
    sh.in_blocks = (num_buckets + FIXED_LEN - 1) / FIXED_LEN;
    sh.out_blocks = sh.noct ? (sh.noct + 1 + 31) / 32 : 0;
    return sh;
}
// workspace of the re-pack half: blk_size, blk_scan, scan scratch
size_t hist_pack_workspace_bytes(uint64_t n_hist, uint32_t num_buckets) {
    const uint64_t nb = n_hist * hist_shape(num_buckets).out_blocks;
    return ((nb + 1) * 8 + 255) / 256 * 256 * 2 + (scan_scratch_words(nb) * 8 + 255) / 256 * 256;
}
hipError_t launch_hist_pack(const uint64_t *raw, const uint32_t *sec_res, const uint64_t *d_series_off, uint64_t n_series,
                            uint64_t n_hist, uint32_t num_buckets, int32_t *d_status, uint32_t *d_first,
                            uint8_t *d_out, uint64_t *d_out_off, uint32_t *d_dropped, uint64_t *d_result, void *d_ws,
                            hipStream_t stream) {
    const HistShape sh = hist_shape(num_buckets);
    const uint64_t nb = n_hist * sh.out_blocks;
    uint8_t *b = (uint8_t *)d_ws;
    const size_t step = ((nb + 1) * 8 + 255) / 256 * 256;
    uint64_t *blk_size = (uint64_t *)b, *blk_scan = (uint64_t *)(b + step), *scratch = (uint64_t *)(b + 2 * step);
    if (n_series) k_hist_status<<<(uint32_t)((n_series + 255) / 256), 256, 0, stream>>>(d_series_off, n_series, sh, sec_res,
                                                                                     d_status, d_first, d_dropped);
    const uint32_t gb = (uint32_t)((nb + ENC_THREADS / 64 - 1) / (ENC_THREADS / 64));
    if (nb) k_hist_sizes<<<gb, ENC_THREADS, 0, stream>>>(raw, n_hist, sh, d_status, d_first, blk_size, d_dropped);
    hipError_t e = launch_exclusive_scan(blk_size, nb, blk_scan, scratch, stream);
    if (e != hipSuccess) return e;
    k_hist_offsets<<<(uint32_t)((n_hist + 256) / 256), 256, 0, stream>>>(n_hist, sh, blk_scan, d_status, d_out_off, d_result);
    if (nb) k_hist_emit<<<gb, ENC_THREADS, 0, stream>>>(raw, n_hist, sh, d_status, d_first, blk_scan, d_out);
    return hipGetLastError();
}

// ---- launcher --------------------------------------------------------------------------------------------
struct EncWs {
    uint32_t *sec_vec;
    uint64_t *sec_size;
    uint64_t *sec_scan;
    uint64_t *scan_scratch;
    uint64_t *tile_state; // one-pass kernel: a state word per tile, and behind them the counter that numbers the tiles
    size_t tile_bytes;
    size_t total;
};
static inline size_t al(size_t x) { return (x + 255) / 256 * 256; }
static EncWs carve_enc_ws(void *base, uint64_t n_vectors, uint64_t n_el) {
    const uint64_t max_slots = n_el / FIXED_LEN + n_vectors;
    EncWs w;
    size_t off = 0;
    uint8_t *b = (uint8_t *)base;
    w.sec_vec = (uint32_t *)(b + off); off += al((max_slots + 1) * 4);
    w.sec_size = (uint64_t *)(b + off); off += al((max_slots + 1) * 8);
    w.sec_scan = (uint64_t *)(b + off); off += al((max_slots + 2) * 8);
    const uint64_t m = max_slots > n_vectors ? max_slots : n_vectors;
    w.scan_scratch = (uint64_t *)(b + off); off += al(scan_scratch_words(m) * 8);
    w.tile_bytes = al((max_slots / EncTile<8>::SECTIONS + 2) * 8 + 8);
    w.tile_state = (uint64_t *)(b + off); off += w.tile_bytes;
    w.total = off;
    return w;
}
size_t encode_workspace_bytes(uint64_t n_vectors, uint64_t n_el) { return carve_enc_ws(nullptr, n_vectors, n_el).total; }

template <int ETYPE>
static void launch_encode_t(const void *d_values, const uint64_t *d_elem_off, uint64_t n, uint64_t max_slots,
                            uint8_t *d_bytes, uint64_t *d_byte_off, uint64_t *d_sec_base, uint64_t *d_result,
                            const EncWs &w, hipStream_t stream) {
    using U = typename EncT<ETYPE>::U;
    const uint32_t vblocks = (uint32_t)((n + 1 + 255) / 256);
    // sections per vector → sec_base (exclusive scan, n+1 entries)
    k_enc_nsec<<<vblocks, 256, 0, stream>>>(d_elem_off, n, d_sec_base);
    (void)launch_exclusive_scan(d_sec_base, n, d_sec_base, w.scan_scratch, stream);
    if (max_slots > 0) {
        k_enc_map<<<vblocks, 256, 0, stream>>>(d_sec_base, n, w.sec_vec);
        // sections per wave: long runs amortise the slot lookup, but the chip wants >= 16 Ki waves
        uint32_t run = (uint32_t)(max_slots / 16384);
        run = run < 1 ? 1 : run > (uint32_t)ENC_MAX_RUN ? (uint32_t)ENC_MAX_RUN : run;
        const uint64_t waves = (max_slots + run - 1) / run;
        const uint32_t sblocks = (uint32_t)((waves + ENC_THREADS / 64 - 1) / (ENC_THREADS / 64));
#ifdef CVEC_ENC_ONE_PASS
        // one pass over the values (k_enc_tile; measured slower than the two passes, see its comment): the state words
        // and the tile counter start at zero
        constexpr int TS = EncTile<EncT<ETYPE>::W>::SECTIONS;
        const uint32_t n_tiles = (uint32_t)((max_slots + TS - 1) / TS);
        (void)hipMemsetAsync(w.tile_state, 0, w.tile_bytes, stream);
        uint32_t *ticket = (uint32_t *)((uint8_t *)w.tile_state + w.tile_bytes - 8);
        const uint32_t tgrid = n_tiles < 512u ? n_tiles : 512u; // two workgroups per CU (their LDS)
        k_enc_tile<ETYPE><<<tgrid, ENC_TILE_THREADS, 0, stream>>>((const U *)d_values, d_elem_off, d_sec_base, n, w.sec_vec,
                                                             max_slots, n_tiles, w.tile_state, ticket, w.sec_scan, d_bytes);
        k_enc_headers<<<vblocks, 256, 0, stream>>>(ETYPE, d_elem_off, d_sec_base, w.sec_scan, n, d_bytes, d_byte_off,
                                                   d_result);
        (void)run; (void)sblocks;
#else
        k_enc_sizes<ETYPE><<<sblocks, ENC_THREADS, 0, stream>>>((const U *)d_values, d_elem_off, d_sec_base, n, w.sec_vec,
                                                                max_slots, run, w.sec_size);
        (void)launch_exclusive_scan(w.sec_size, max_slots, w.sec_scan, w.scan_scratch, stream);
        k_enc_headers<<<vblocks, 256, 0, stream>>>(ETYPE, d_elem_off, d_sec_base, w.sec_scan, n, d_bytes, d_byte_off,
                                                   d_result);
        k_enc_emit<ETYPE><<<sblocks, ENC_THREADS, 0, stream>>>((const U *)d_values, d_elem_off, d_sec_base, n, w.sec_vec,
                                                               max_slots, run, w.sec_scan, d_bytes);
#endif
    } else {
        (void)hipMemsetAsync(w.sec_scan, 0, 16, stream);
        k_enc_headers<<<vblocks, 256, 0, stream>>>(ETYPE, d_elem_off, d_sec_base, w.sec_scan, n, d_bytes, d_byte_off,
                                                   d_result);
    }
}

hipError_t launch_encode(int32_t etype, const void *d_values, const uint64_t *d_elem_off, uint64_t n_vectors,
                         uint64_t n_elements_total, uint8_t *d_bytes, size_t bytes_cap, uint64_t *d_byte_off,
                         uint64_t *d_sec_base, uint64_t *d_result, void *d_workspace, size_t workspace_bytes,
                         hipStream_t stream) {
    (void)bytes_cap; (void)workspace_bytes;
    const EncWs w = carve_enc_ws(d_workspace, n_vectors, n_elements_total);
    const uint64_t max_slots = n_elements_total / FIXED_LEN + n_vectors;
    switch (etype) {
    case CVEC_U32: launch_encode_t<CVEC_U32>(d_values, d_elem_off, n_vectors, max_slots, d_bytes, d_byte_off, d_sec_base, d_result, w, stream); break;
    case CVEC_U64: launch_encode_t<CVEC_U64>(d_values, d_elem_off, n_vectors, max_slots, d_bytes, d_byte_off, d_sec_base, d_result, w, stream); break;
    case CVEC_F32: launch_encode_t<CVEC_F32>(d_values, d_elem_off, n_vectors, max_slots, d_bytes, d_byte_off, d_sec_base, d_result, w, stream); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

} // namespace cvec
