// cvec_device.hpp — shared device-side definitions for the gfx950 kernels.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "cvec.h"

namespace cvec {

constexpr int FIXED_LEN = 256;   // elements per section (reference src/section.rs:184)
constexpr int HEADER_BYTES = 16; // vector header (reference src/vector.rs:110)
constexpr int DELTA_HDR = 12;    // Delta section header (reference src/section.rs:607)

// Section slots given to a vector of `len` >= HEADER_BYTES bytes whose header claims `num_elements`
// (reference src/vector.rs:165): ceil(num_elements / 256), except that a claim beyond what the bytes
// could hold (a section takes at least one byte) gets one slot more than that bound.  The walk can
// then never fill the slots, the vector still reports a section-count mismatch, and a damaged header
// cannot make the library size its buffers by anything but the bytes it was given.
__host__ __device__ inline uint64_t header_slots(uint32_t num_elements, uint64_t len) {
    const uint64_t claimed = ((uint64_t)num_elements + FIXED_LEN - 1) / FIXED_LEN;
    const uint64_t bound = len - HEADER_BYTES;
    return claimed > bound ? bound + 1 : claimed;
}

enum { READ_ALL = 0, READ_INDEX_ONLY = 1, READ_INDEXED = 2 }; // phases of a read operation (launch_read)
enum { PATH_AUTO = 0, PATH_SLOTS = 1, PATH_VSTREAM = 2 }; // which read kernel serves an ungated, unindexed pass

// Range counters of the section kernel in a read workspace: WORK_COUNTERS counters, each on a 128-byte line of its own.
constexpr int WORK_COUNTERS = 256;
constexpr int WORK_STRIDE = 32; // in 32-bit words
// Words 2..3 of a counter's line: a 64-bit partial total of the section kernel.  One more line behind the counters
// holds the number of the last pass in which a vector or a section failed.
constexpr int FAIL_FLAG_WORD = WORK_COUNTERS * WORK_STRIDE;

// Second tag word of a read workspace: set when its per-vector array holds the valid-section counts of a
// vector-streaming pass (k_vstream) instead of the header statuses of an index pass (k_index).
constexpr uint64_t VS_MAGIC = 0x7673747265616d31ull;

// Section kinds as stored in a section descriptor: the reference's type bytes
// (src/section.rs:32-38) plus internal markers.
constexpr uint32_t K_NULL = 0, K_NP = 1, K_DELTA = 3, K_CONST = 5, K_XOR = 6;
constexpr uint32_t K_RAW = 2;        // internal: up to 32 bare octets of a legacy NibblePack stream, no section header
constexpr uint32_t K_GATED = 4;      // internal: well-formed section that a gate switched off: no decode, empty mask
constexpr uint32_t K_ILLEGAL = 0x7e; // well-formed section of a kind the element type cannot hold
constexpr uint32_t K_SKIP = 0x7f;    // slot after a parse error / beyond the sections found

// Section descriptor: byte offset in the blob (40 bits) | num_bytes() (17 bits) | kind (7 bits).
__host__ __device__ inline uint64_t desc_pack(uint64_t off, uint32_t len, uint32_t kind) {
    return off | ((uint64_t)len << 40) | ((uint64_t)kind << 57);
}
__host__ __device__ inline uint64_t desc_off(uint64_t d) { return d & ((1ull << 40) - 1); }
__host__ __device__ inline uint32_t desc_len(uint64_t d) { return (uint32_t)(d >> 40) & 0x1ffffu; }
__host__ __device__ inline uint32_t desc_kind(uint64_t d) { return (uint32_t)(d >> 57); }

// Largest span of a section the octet walk can touch: header + 32 octets of 2 + 8*W/2... bytes.
template <int W> struct SectLimits;
template <> struct SectLimits<4> { static constexpr uint32_t MAX_BYTES = DELTA_HDR + 32 * 34; };
template <> struct SectLimits<8> { static constexpr uint32_t MAX_BYTES = DELTA_HDR + 32 * 66; };

// Unaligned little-endian loads.  gfx950 runs with unaligned global/LDS access enabled, so these
// compile to single global_load / ds_read instructions.
typedef uint16_t __attribute__((aligned(1))) u16u;
typedef uint32_t __attribute__((aligned(1))) u32u;
typedef uint64_t __attribute__((aligned(1))) u64u;

template <int W> struct Elem;
template <> struct Elem<4> { using U = uint32_t; };
template <> struct Elem<8> { using U = uint64_t; };

// Per-section result word: low 16 bits = count (matches / elements stored), high 16 = status.
__device__ inline uint32_t res_pack(uint32_t count, uint32_t err) { return count | (err << 16); }

// Predicate / sink arguments passed to kernels by value.
struct PredArgs {
    uint32_t count;
    uint32_t match_zero;
    uint64_t v[CVEC_MAX_PRED_VALUES];
};
struct SinkArgs {
    uint32_t add_const;
    uint32_t _pad;
    uint64_t base;
};

} // namespace cvec
