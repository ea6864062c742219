// cvec_stream.hip — the legacy NibblePack stream codecs for batches of streams:
//   pack_u64_delta + DeltaSink        (reference src/nibblepacking.rs:15-23, :239-279)
//   pack_f64_xor + DoubleXorSink      (src/nibblepacking.rs:29-45, :284-325, :392-400)
//   pack_u64 / unpack                 (src/nibblepacking.rs:68-93, :364-377)
// A stream has no section structure: its octets form one serial chain, and the decoded values go
// through a true element-to-element prefix sum / prefix XOR.  Unpack = index (one wave per stream
// walks its octet chain once, in LDS, and cuts it into blocks of 32 octets) -> the section kernel decodes
// the blocks (K_RAW descriptors) -> per-block totals -> per-stream carries -> per-block wavefront
// scan that writes the final values.  Pack = per-block sizes -> scans -> emit.
#include "cvec_device.hpp"

namespace cvec {

hipError_t launch_exclusive_scan(const uint64_t *d_in, uint64_t n, uint64_t *d_out, uint64_t *d_scratch,
                                 hipStream_t stream);
uint64_t scan_scratch_words(uint64_t n);
hipError_t launch_raw_blocks_u64(const uint8_t *d_bytes, const uint64_t *desc, uint64_t n_blocks, uint32_t *sec_res,
                                 uint64_t *d_out, hipStream_t stream);

// values that go through NibblePack: all n for the delta codec, n - 1 for XOR (the first is stored raw)
__device__ __host__ inline uint64_t packed_values(int kind, uint64_t n) {
    return kind == CVEC_STREAM_F64_XOR ? (n ? n - 1 : 0) : n;
}

__global__ void k_stream_nblk(int kind, const uint64_t *__restrict__ elem_off, uint64_t n, uint64_t *__restrict__ nblk) {
    const uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    nblk[s] = (packed_values(kind, elem_off[s + 1] - elem_off[s]) + FIXED_LEN - 1) / FIXED_LEN;
}

// ---- unpack ---------------------------------------------------------------------------------------
// One wave per stream: the serial octet chain of `unpack` (src/nibblepacking.rs:364-377), with the
// checks of nibble_unpack8 (:418, :425) and "an octet must end inside the stream".  The chain is a
// pointer chase (an octet's length is in its two header bytes).  The wave stages the stream through
// LDS in tiles; for every byte position of the tile the lanes compute in parallel where an octet
// starting there would end (a jump table), square the table three times (pointer doubling: 2, 4, 8
// octets ahead), and then every lane follows the real chain eight octets per LDS read (uniform
// addresses: LDS broadcasts).  Positions whose octet would leave the stream, or that the tile does
// not cover, have no table entry: there the chain advances one octet at a time with the exact
// checks.  The next tile is prefetched into registers.  Writes one K_RAW descriptor per block of 32
// octets.
constexpr int IDX_TILE = 1024;           // octets that start here are covered by the jump table
constexpr int IDX_EXT = IDX_TILE + 128;  // staged bytes / table entries: an octet (<= 66 bytes) may end past the tile
constexpr int IDX_WAVES = 4;
static_assert(IDX_EXT % 64 == 0, "whole waves of table entries");
constexpr int IDX_PRE = (IDX_TILE + 256) / 16; // 16-byte pieces prefetched behind the tile
constexpr int IDX_PRE_REGS = (IDX_PRE + 63) / 64;
constexpr uint32_t IDX_NONE = 0xffff;

__device__ inline uint32_t octet_size(uint32_t mask, uint32_t word) {
    return mask ? 2 + ((((word >> 4) + 1) * __popc(mask) + 1) >> 1) : 1;
}

__global__ __launch_bounds__(64 * IDX_WAVES) void k_stream_index(int kind, const uint8_t *__restrict__ bytes,
                                                                 const uint64_t *__restrict__ byte_off,
                                                                 const uint64_t *__restrict__ elem_off,
                                                                 const uint64_t *__restrict__ blk_base, uint64_t n,
                                                                 uint64_t padded_bytes, uint64_t *__restrict__ desc,
                                                                 uint32_t *__restrict__ blk_stream,
                                                                 int32_t *__restrict__ status) {
    __shared__ __attribute__((aligned(16))) uint8_t s_tile[IDX_WAVES][IDX_EXT];
    __shared__ uint16_t s_jump[IDX_WAVES][2][IDX_EXT];
    const int lane = threadIdx.x & 63;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint64_t s = (uint64_t)blockIdx.x * IDX_WAVES + wave;
    if (s >= n) return;
    uint8_t *tile = s_tile[wave];
    uint16_t *ja = s_jump[wave][0], *jb = s_jump[wave][1];
    const uint64_t nval = elem_off[s + 1] - elem_off[s];
    const uint64_t start = byte_off[s], end = byte_off[s + 1];
    const uint64_t b0 = blk_base[s], nb = blk_base[s + 1] - b0;
    int st = CVEC_OK;
    uint64_t p = start;
    if (kind == CVEC_STREAM_F64_XOR) {
        if (nval < 1) st = CVEC_INPUT_TOO_SHORT;            // unpack_f64_xor asserts num_values >= 1 (:395)
        else if (end - start < 8) st = CVEC_NOT_ENOUGH_SPACE; // the raw first value (:396)
        p += 8;
    }
    const uint64_t noct = (packed_values(kind, nval) + 7) / 8;

    auto piece = [&](uint64_t a) { // 16 bytes at the 16-aligned offset a, zeros past the padded blob
        return a + 16 <= padded_bytes ? *(const uint4 *)(bytes + a) : make_uint4(0, 0, 0, 0);
    };
    uint64_t tile_base = 0, pre_base = 0;
    uint4 pre[IDX_PRE_REGS];
    bool have_pre = false;
    // squares a jump table: dst[r] = src[src[r]]
    // (all first reads, then all second reads, then the writes: written per entry, the two dependent reads of
    // every entry wait for each other in turn, 36 LDS round trips per table instead of two)
    auto hop = [&](const uint16_t *src, uint16_t *dst) {
        constexpr int K = IDX_TILE / 64; // entries from IDX_TILE on are IDX_NONE in every table (set once, below)
        uint32_t m[K], v[K];
#pragma unroll
        for (int k = 0; k < K; k++) m[k] = src[64 * k + lane];
#pragma unroll
        for (int k = 0; k < K; k++) v[k] = src[m[k] == IDX_NONE ? 0u : m[k]];
#pragma unroll
        for (int k = 0; k < K; k++) dst[64 * k + lane] = m[k] == IDX_NONE ? (uint16_t)IDX_NONE : (uint16_t)v[k];
    };
    auto stage = [&]() {
        tile_base = p & ~15ull;
        if (!have_pre) {
            for (uint32_t i = lane; i < IDX_EXT / 16; i += 64) ((uint4 *)tile)[i] = piece(tile_base + 16ull * i);
        } else { // the chain left the old tile by less than an octet: the prefetched window covers the new one
            const uint32_t shift = (uint32_t)((tile_base - pre_base) >> 4);
#pragma unroll
            for (int k = 0; k < IDX_PRE_REGS; k++) {
                const uint32_t i = 64 * k + lane;
                if (i < IDX_PRE && i >= shift && i - shift < IDX_EXT / 16) ((uint4 *)tile)[i - shift] = pre[k];
            }
        }
        pre_base = tile_base + IDX_TILE;
#pragma unroll
        for (int k = 0; k < IDX_PRE_REGS; k++) pre[k] = piece(pre_base + 16ull * (64 * k + lane));
        have_pre = true;
        const uint64_t left = end - tile_base; // p < end here
        const uint32_t end_rel = left < 0xffffu ? (uint32_t)left : 0xffffu;
        {   // one octet ahead; both bytes of every position are read before any of them is used
            constexpr int K = IDX_TILE / 64;
            uint32_t mk[K], wd[K];
#pragma unroll
            for (int k = 0; k < K; k++) {
                const uint32_t r = 64 * k + lane;
                mk[k] = tile[r];
                wd[k] = tile[r + 1];
            }
#pragma unroll
            for (int k = 0; k < K; k++) {
                const uint32_t r = 64 * k + lane;
                const uint32_t e = r + octet_size(mk[k], wd[k]);
                ja[r] = (uint16_t)(e <= end_rel ? e : IDX_NONE); // the octet lies inside the stream
            }
        }
        hop(ja, jb); // 2
        hop(jb, ja); // 4
        hop(ja, jb); // 8 octets ahead
    };
    // positions behind the tile start no octet that the tables cover: constant in both tables
    for (uint32_t r = IDX_TILE + lane; r < IDX_EXT; r += 64) { ja[r] = (uint16_t)IDX_NONE; jb[r] = (uint16_t)IDX_NONE; }
    uint64_t blk = 0, blk_start = p, o = 0;
    bool staged = false;
    while (o < noct && st == CVEC_OK) {
        if ((o & 31) == 0) blk_start = p;
        if (p >= end) { st = CVEC_NOT_ENOUGH_SPACE; break; }
        if (!staged || p - tile_base >= IDX_TILE) { stage(); staged = true; }
        // every lane holds the same chain position: keep it in scalar registers
        const uint32_t r = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(p - tile_base));
        const uint64_t want = noct - o < 32 - (o & 31) ? noct - o : 32 - (o & 31); // octets to the end of the block
        const uint32_t far = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)jb[r]);
        if (want >= 8 && far != IDX_NONE) {
            p = tile_base + far;
            o += 8;
        } else {
            const uint32_t mask = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)tile[r]);
            const uint32_t word = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)tile[r + 1]);
            if (mask && end - p < 2) { st = CVEC_NOT_ENOUGH_SPACE; break; }
            const uint64_t size = octet_size(mask, word);
            if (size > end - p) { st = CVEC_NOT_ENOUGH_SPACE; break; }
            p += size;
            o += 1;
        }
        if ((o & 31) == 0 || o == noct) {
            if (lane == 0) desc[b0 + blk] = desc_pack(blk_start, (uint32_t)(p - blk_start), K_RAW);
            blk++;
        }
    }
    for (uint64_t i = blk + lane; i < nb; i += 64) desc[b0 + i] = desc_pack(p < end ? p : end, 0, K_SKIP);
    for (uint64_t i = lane; i < nb; i += 64) blk_stream[b0 + i] = (uint32_t)s;
    if (lane == 0) status[s] = st;
}

// Slots past the real block count (known only on the device) become empty K_SKIP descriptors at
// the end of the blob, which keeps the descriptor offsets non-decreasing.
__global__ void k_fill_skip(const uint64_t *__restrict__ n_real, uint64_t fill_off, uint64_t *__restrict__ desc,
                            uint64_t max_blocks) {
    const uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b < max_blocks && b >= *n_real) desc[b] = desc_pack(fill_off, 0, K_SKIP);
}

// Combine of the stream's running state: wrapping sum (DeltaSink), XOR (DoubleXorSink) or none.
template <int KIND> __device__ inline uint64_t comb(uint64_t a, uint64_t b) {
    // plain streams have no running state: "keep the right operand" is associative, so they run
    // through the same scan and come out unchanged
    return KIND == CVEC_STREAM_F64_XOR ? a ^ b : KIND == CVEC_STREAM_U64_PLAIN ? b : a + b;
}

// Inclusive wave scan of one value per lane.
template <int KIND> __device__ inline uint64_t wave_scan_incl(uint64_t x, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint64_t y = __shfl_up(x, d, 64);
        if (lane >= d) x = comb<KIND>(x, y);
    }
    return x;
}

// One wave per block: lane L owns elements 4L..4L+3 of the block (element order, not octet lanes).
template <int KIND>
__global__ __launch_bounds__(256) void k_stream_totals(const uint64_t *__restrict__ raw, uint64_t n_blocks,
                                                       uint64_t *__restrict__ totals) {
    const int lane = threadIdx.x & 63;
    const uint64_t blk = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (blk >= n_blocks) return;
    const uint4 *p = (const uint4 *)(raw + blk * FIXED_LEN + 4 * lane);
    const uint4 a = p[0], b = p[1];
    uint64_t t = comb<KIND>(comb<KIND>(((uint64_t)a.y << 32) | a.x, ((uint64_t)a.w << 32) | a.z),
                            comb<KIND>(((uint64_t)b.y << 32) | b.x, ((uint64_t)b.w << 32) | b.z));
    t = wave_scan_incl<KIND>(t, lane);
    if (lane == 63) totals[blk] = t;
}

// One lane per stream: exclusive running state at the start of each of its blocks.  The octets of
// the last block past the stream's values are zero padding of the encoder (they do not change the
// state), and a failed stream's blocks are never read back.
template <int KIND>
__global__ void k_stream_carries(const uint8_t *__restrict__ bytes, const uint64_t *__restrict__ byte_off,
                                 const uint64_t *__restrict__ blk_base, uint64_t n,
                                 const int32_t *__restrict__ status, const uint64_t *__restrict__ totals,
                                 uint64_t *__restrict__ carries) {
    const uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    uint64_t acc = 0;
    if (KIND == CVEC_STREAM_F64_XOR && status[s] == CVEC_OK) acc = *(const u64u *)(bytes + byte_off[s]);
    for (uint64_t b = blk_base[s]; b < blk_base[s + 1]; b++) {
        carries[b] = acc;
        acc = comb<KIND>(acc, totals[b]);
    }
}

// One wave per block: in-block inclusive scan in element order + the block's carry -> final values,
// written compactly (stream s occupies out[elem_off[s] .. elem_off[s+1])).
template <int KIND>
__global__ __launch_bounds__(256) void k_stream_scan_write(const uint64_t *__restrict__ raw,
                                                           const uint64_t *__restrict__ blk_base,
                                                           const uint32_t *__restrict__ blk_stream,
                                                           const uint64_t *__restrict__ elem_off, uint64_t n,
                                                           uint64_t n_blocks, const uint64_t *__restrict__ carries,
                                                           const int32_t *__restrict__ status,
                                                           uint64_t *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const uint64_t blk = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (blk >= n_blocks) return;
    if (blk >= blk_base[n]) return; // past the real block count
    const uint64_t s = blk_stream[blk];
    if (status[s] != CVEC_OK) return;
    const uint64_t nval = elem_off[s + 1] - elem_off[s];
    const uint64_t shift = KIND == CVEC_STREAM_F64_XOR ? 1 : 0;
    const uint64_t first = (blk - blk_base[s]) * FIXED_LEN; // index among the packed values
    const uint4 *p = (const uint4 *)(raw + blk * FIXED_LEN + 4 * lane);
    const uint4 a = p[0], b = p[1];
    uint64_t x[4] = {((uint64_t)a.y << 32) | a.x, ((uint64_t)a.w << 32) | a.z, ((uint64_t)b.y << 32) | b.x,
                     ((uint64_t)b.w << 32) | b.z};
    x[1] = comb<KIND>(x[0], x[1]); x[2] = comb<KIND>(x[1], x[2]); x[3] = comb<KIND>(x[2], x[3]);
    const uint64_t incl = wave_scan_incl<KIND>(x[3], lane);
    uint64_t excl = __shfl_up(incl, 1, 64);
    if (lane == 0) excl = 0;
    const uint64_t base = comb<KIND>(carries[blk], excl);
    uint64_t *o = out + elem_off[s] + shift + first + 4 * lane;
#pragma unroll
    for (int i = 0; i < 4; i++)
        if (shift + first + 4 * lane + i < nval) o[i] = comb<KIND>(base, x[i]);
    if (KIND == CVEC_STREAM_F64_XOR && first == 0 && lane == 0) out[elem_off[s]] = carries[blk]; // the raw first value
}

// The same with one wave per stream, which walks the stream's blocks in order and carries the running state in a
// register: no block totals, no carries, the raw values are read once.  For batches of many streams (the
// launcher decides); a batch of a few long streams has too few waves for this and takes the three kernels above.
template <int KIND>
__global__ __launch_bounds__(256) void k_stream_scan_serial(const uint8_t *__restrict__ bytes, const uint64_t *__restrict__ byte_off,
                                                            const uint64_t *__restrict__ raw, const uint64_t *__restrict__ blk_base,
                                                            const uint64_t *__restrict__ elem_off, uint64_t n,
                                                            const int32_t *__restrict__ status, uint64_t *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const uint64_t s = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (s >= n) return;
    if (status[s] != CVEC_OK) return;
    const uint64_t b0 = blk_base[s], b1 = blk_base[s + 1];
    if (b0 == b1) return; // k_stream_finish wrote what such a stream holds
    const uint64_t nval = elem_off[s + 1] - elem_off[s];
    const uint64_t shift = KIND == CVEC_STREAM_F64_XOR ? 1 : 0;
    uint64_t acc = 0;
    if (KIND == CVEC_STREAM_F64_XOR) {
        acc = *(const u64u *)(bytes + byte_off[s]); // the raw first value
        if (lane == 0) out[elem_off[s]] = acc;
    }
    const uint4 *p = (const uint4 *)(raw + b0 * FIXED_LEN + 4 * lane);
    uint4 a = p[0], b = p[1];
    for (uint64_t blk = b0; blk < b1; blk++) {
        uint4 na = a, nb = b;
        if (blk + 1 < b1) { // the next block's values are on their way while this one is scanned and written
            const uint4 *q = (const uint4 *)(raw + (blk + 1) * FIXED_LEN + 4 * lane);
            na = q[0]; nb = q[1];
        }
        uint64_t x[4] = {((uint64_t)a.y << 32) | a.x, ((uint64_t)a.w << 32) | a.z, ((uint64_t)b.y << 32) | b.x,
                         ((uint64_t)b.w << 32) | b.z};
        x[1] = comb<KIND>(x[0], x[1]); x[2] = comb<KIND>(x[1], x[2]); x[3] = comb<KIND>(x[2], x[3]);
        const uint64_t incl = wave_scan_incl<KIND>(x[3], lane);
        uint64_t excl = __shfl_up(incl, 1, 64);
        if (lane == 0) excl = 0;
        const uint64_t base = comb<KIND>(acc, excl);
        const uint64_t first = (blk - b0) * FIXED_LEN; // index among the packed values
        uint64_t *o = out + elem_off[s] + shift + first + 4 * lane;
#pragma unroll
        for (int i = 0; i < 4; i++)
            if (shift + first + 4 * lane + i < nval) o[i] = comb<KIND>(base, x[i]);
        acc = comb<KIND>(acc, __shfl(incl, 63, 64));
        a = na; b = nb;
    }
}

// A stream without any packed value (XOR stream of one value) has no block: its first value is
// written here; also folds the statuses into the result block.
template <int KIND>
__global__ void k_stream_finish(const uint8_t *__restrict__ bytes, const uint64_t *__restrict__ byte_off,
                                const uint64_t *__restrict__ blk_base, const uint64_t *__restrict__ elem_off, uint64_t n,
                                const uint32_t *__restrict__ sec_res, int32_t *__restrict__ status,
                                uint64_t *__restrict__ out, uint64_t *__restrict__ result) {
    const uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    int st = status[s];
    for (uint64_t b = blk_base[s]; b < blk_base[s + 1] && st == CVEC_OK; b++)
        if (sec_res[b] >> 16) st = (int)(sec_res[b] >> 16);
    status[s] = st;
    if (st == CVEC_OK) {
        if (KIND == CVEC_STREAM_F64_XOR && blk_base[s + 1] == blk_base[s]) out[elem_off[s]] = *(const u64u *)(bytes + byte_off[s]);
        atomicAdd((unsigned long long *)&result[0], (unsigned long long)(elem_off[s + 1] - elem_off[s]));
    } else {
        atomicAdd((unsigned long long *)&result[1], 1ull);
        atomicMax((unsigned long long *)&result[2], (unsigned long long)st);
    }
}

struct StreamUnpackWs {
    uint64_t *nblk, *blk_base, *scan_scratch, *desc, *raw, *totals, *carries;
    uint32_t *sec_res, *blk_stream;
    size_t total;
};
static inline size_t al256(size_t x) { return (x + 255) / 256 * 256; }
static StreamUnpackWs carve_unpack(void *base, uint64_t n, uint64_t max_blocks) {
    StreamUnpackWs w;
    uint8_t *b = (uint8_t *)base;
    size_t off = 0;
    w.nblk = (uint64_t *)(b + off); off += al256((n + 1) * 8);
    w.blk_base = (uint64_t *)(b + off); off += al256((n + 1) * 8);
    w.scan_scratch = (uint64_t *)(b + off); off += al256(scan_scratch_words(n) * 8);
    w.desc = (uint64_t *)(b + off); off += al256((max_blocks + 1) * 8);
    w.sec_res = (uint32_t *)(b + off); off += al256((max_blocks + 1) * 4);
    w.blk_stream = (uint32_t *)(b + off); off += al256((max_blocks + 1) * 4);
    w.totals = (uint64_t *)(b + off); off += al256((max_blocks + 1) * 8);
    w.carries = (uint64_t *)(b + off); off += al256((max_blocks + 1) * 8);
    w.raw = (uint64_t *)(b + off); off += al256(max_blocks * FIXED_LEN * 8 + 64);
    w.total = off;
    return w;
}
// blocks of a batch: at most one partial block per stream
static inline uint64_t max_blocks_of(uint64_t n_streams, uint64_t n_values_total) {
    return n_values_total / FIXED_LEN + n_streams;
}
size_t stream_unpack_workspace_bytes(uint64_t n_streams, uint64_t n_values_total) {
    return carve_unpack(nullptr, n_streams, max_blocks_of(n_streams, n_values_total)).total;
}

#ifndef CVEC_STREAM_SERIAL_MIN
#define CVEC_STREAM_SERIAL_MIN 2048
#endif
constexpr uint64_t STREAM_SERIAL_MIN = CVEC_STREAM_SERIAL_MIN;

template <int KIND>
static hipError_t unpack_t(const uint8_t *d_bytes, const uint64_t *d_byte_off, uint64_t n_bytes,
                           const uint64_t *d_elem_off, uint64_t n, uint64_t n_values_total, uint64_t *d_out,
                           int32_t *d_status, uint64_t *d_result, void *d_ws, hipStream_t stream) {
    const uint64_t maxb = max_blocks_of(n, n_values_total);
    StreamUnpackWs w = carve_unpack(d_ws, n, maxb);
    hipError_t e = hipMemsetAsync(d_result, 0, CVEC_RESULT_WORDS * sizeof(uint64_t), stream);
    if (e != hipSuccess || n == 0) return e;
    const uint32_t g = (uint32_t)((n + 255) / 256);
    k_stream_nblk<<<g, 256, 0, stream>>>(KIND, d_elem_off, n, w.nblk);
    e = launch_exclusive_scan(w.nblk, n, w.blk_base, w.scan_scratch, stream);
    if (e != hipSuccess) return e;
    k_stream_index<<<(uint32_t)((n + IDX_WAVES - 1) / IDX_WAVES), 64 * IDX_WAVES, 0, stream>>>(KIND, d_bytes, d_byte_off, d_elem_off, w.blk_base, n,
                                                                 (n_bytes + 15) & ~15ull, w.desc, w.blk_stream, d_status);
    // The block count is only known on the device; the section kernel runs over the upper bound and
    // skips the slots past the real count (their descriptors are K_SKIP).
    k_fill_skip<<<(uint32_t)((maxb + 255) / 256), 256, 0, stream>>>(w.blk_base + n, n_bytes, w.desc, maxb);
    e = launch_raw_blocks_u64(d_bytes, w.desc, maxb, w.sec_res, w.raw, stream);
    if (e != hipSuccess) return e;
    const uint32_t gb = (uint32_t)((maxb + 3) / 4);
    // enough streams for a wave each to fill the chip (256 CUs x 8 waves): one pass over the raw values
    const bool serial = n >= STREAM_SERIAL_MIN;
    if (!serial) {
        k_stream_totals<KIND><<<gb, 256, 0, stream>>>(w.raw, maxb, w.totals);
        k_stream_carries<KIND><<<g, 256, 0, stream>>>(d_bytes, d_byte_off, w.blk_base, n, d_status, w.totals, w.carries);
    }
    k_stream_finish<KIND><<<g, 256, 0, stream>>>(d_bytes, d_byte_off, w.blk_base, d_elem_off, n, w.sec_res, d_status, d_out,
                                                 d_result);
    if (serial)
        k_stream_scan_serial<KIND><<<(uint32_t)((n + 3) / 4), 256, 0, stream>>>(d_bytes, d_byte_off, w.raw, w.blk_base, d_elem_off, n,
                                                                                d_status, d_out);
    else
        k_stream_scan_write<KIND><<<gb, 256, 0, stream>>>(w.raw, w.blk_base, w.blk_stream, d_elem_off, n, maxb, w.carries, d_status, d_out);
    return hipGetLastError();
}

hipError_t launch_stream_unpack(int kind, const uint8_t *d_bytes, const uint64_t *d_byte_off, uint64_t n_bytes,
                                const uint64_t *d_elem_off, uint64_t n, uint64_t n_values_total, uint64_t *d_out,
                                int32_t *d_status, uint64_t *d_result, void *d_ws, hipStream_t stream) {
    if (kind == CVEC_STREAM_F64_XOR)
        return unpack_t<CVEC_STREAM_F64_XOR>(d_bytes, d_byte_off, n_bytes, d_elem_off, n, n_values_total, d_out, d_status,
                                             d_result, d_ws, stream);
    if (kind == CVEC_STREAM_U64_PLAIN)
        return unpack_t<CVEC_STREAM_U64_PLAIN>(d_bytes, d_byte_off, n_bytes, d_elem_off, n, n_values_total, d_out, d_status,
                                               d_result, d_ws, stream);
    return unpack_t<CVEC_STREAM_U64_DELTA>(d_bytes, d_byte_off, n_bytes, d_elem_off, n, n_values_total, d_out, d_status,
                                           d_result, d_ws, stream);
}

// ---- histograms: DeltaDiffPackSink for batches of histogram series ---------------------------------
// (reference src/histogram.rs:92-171).  The histograms are plain NibblePack streams of num_buckets
// values each; they are unpacked like any other stream batch (index -> K_RAW blocks through the
// section kernel), then cvec_encode.hip re-packs every histogram as the difference to its
// predecessor in the series.
size_t hist_pack_workspace_bytes(uint64_t n_hist, uint32_t num_buckets);
hipError_t launch_hist_pack(const uint64_t *raw, const uint32_t *sec_res, const uint64_t *d_series_off, uint64_t n_series,
                            uint64_t n_hist, uint32_t num_buckets, int32_t *d_status, uint32_t *d_first, uint8_t *d_out,
                            uint64_t *d_out_off, uint32_t *d_dropped, uint64_t *d_result, void *d_ws, hipStream_t stream);

__global__ void k_hist_elem_off(uint64_t n_hist, uint32_t num_buckets, uint64_t *__restrict__ elem_off) {
    const uint64_t h = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (h <= n_hist) elem_off[h] = h * num_buckets;
}

struct HistWs {
    StreamUnpackWs u;
    uint64_t *elem_off;
    uint32_t *first;
    void *pack;
    size_t total;
};
static HistWs carve_hist(void *base, uint64_t n_hist, uint32_t num_buckets) {
    HistWs w;
    const uint64_t total_values = n_hist * num_buckets;
    w.u = carve_unpack(base, n_hist, max_blocks_of(n_hist, total_values));
    uint8_t *b = (uint8_t *)base;
    size_t off = w.u.total;
    w.elem_off = (uint64_t *)(b + off); off += al256((n_hist + 1) * 8);
    w.first = (uint32_t *)(b + off); off += al256((n_hist + 1) * 4);
    w.pack = b + off; off += hist_pack_workspace_bytes(n_hist, num_buckets);
    w.total = off;
    return w;
}
size_t hist_workspace_bytes(uint64_t n_hist, uint32_t num_buckets) { return carve_hist(nullptr, n_hist, num_buckets).total; }

hipError_t launch_hist_delta_diff(const uint8_t *d_bytes, const uint64_t *d_byte_off, uint64_t n_bytes,
                                  const uint64_t *d_series_off, uint64_t n_series, uint64_t n_hist, uint32_t num_buckets,
                                  uint8_t *d_out, uint64_t *d_out_off, uint32_t *d_dropped, int32_t *d_status,
                                  uint64_t *d_result, void *d_ws, hipStream_t stream) {
    constexpr int KIND = CVEC_STREAM_U64_PLAIN;
    HistWs w = carve_hist(d_ws, n_hist, num_buckets);
    const uint64_t maxb = max_blocks_of(n_hist, n_hist * num_buckets);
    hipError_t e = hipMemsetAsync(d_result, 0, CVEC_RESULT_WORDS * sizeof(uint64_t), stream);
    if (e != hipSuccess) return e;
    const uint32_t g = (uint32_t)((n_hist + 256) / 256);
    k_hist_elem_off<<<g, 256, 0, stream>>>(n_hist, num_buckets, w.elem_off);
    if (n_hist) {
        k_stream_nblk<<<g, 256, 0, stream>>>(KIND, w.elem_off, n_hist, w.u.nblk);
        e = launch_exclusive_scan(w.u.nblk, n_hist, w.u.blk_base, w.u.scan_scratch, stream);
        if (e != hipSuccess) return e;
        k_stream_index<<<(uint32_t)((n_hist + IDX_WAVES - 1) / IDX_WAVES), 64 * IDX_WAVES, 0, stream>>>(
            KIND, d_bytes, d_byte_off, w.elem_off, w.u.blk_base, n_hist, (n_bytes + 15) & ~15ull, w.u.desc, w.u.blk_stream,
            d_status);
        k_fill_skip<<<(uint32_t)((maxb + 255) / 256), 256, 0, stream>>>(w.u.blk_base + n_hist, n_bytes, w.u.desc, maxb);
        e = launch_raw_blocks_u64(d_bytes, w.u.desc, maxb, w.u.sec_res, w.u.raw, stream);
        if (e != hipSuccess) return e;
    }
    return launch_hist_pack(w.u.raw, w.u.sec_res, d_series_off, n_series, n_hist, num_buckets, d_status, w.first, d_out,
                            d_out_off, d_dropped, d_result, w.pack, stream);
}

} // namespace cvec
