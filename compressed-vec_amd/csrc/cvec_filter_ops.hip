// cvec_filter_ops.hip — gfx950 kernels that consume the per-section 256-bit masks of the filter pass:
// MultiVectorFilter (reference src/filter.rs:239-286: AND of the masks of several vectors, section
// by section) and match_positions (src/filter.rs:302-321: element positions of the set bits).
// Both are plain streaming kernels over 32-byte masks; the decode work was done by k_sections.
#include "cvec_device.hpp"

namespace cvec {

hipError_t launch_exclusive_scan(const uint64_t *d_in, uint64_t n, uint64_t *d_out, uint64_t *d_scratch,
                                 hipStream_t stream);
uint64_t scan_scratch_words(uint64_t n);

// Row (vector index) of a global section slot: the last v with sec_base[v] <= slot.
__device__ inline uint64_t row_of_slot(const uint64_t *__restrict__ sec_base, uint64_t n, uint64_t slot) {
    uint64_t lo = 0, hi = n; // invariant: sec_base[lo] <= slot < sec_base[hi]
    while (hi - lo > 1) {
        const uint64_t mid = (lo + hi) >> 1;
        if (sec_base[mid] <= slot) lo = mid; else hi = mid;
    }
    return lo;
}

// Number of leading sections of every vector that yield a mask / decoded data: the reference's
// iteration stops at the first section that fails (src/filter.rs:226) or cannot be parsed.
__global__ __launch_bounds__(256) void k_valid_sections(const uint64_t *__restrict__ sec_base, uint64_t n,
                                                        const uint64_t *__restrict__ desc,
                                                        const uint32_t *__restrict__ sec_res,
                                                        const int32_t *__restrict__ vstatus,
                                                        const uint64_t *__restrict__ tag,
                                                        uint32_t *__restrict__ valid, uint8_t *__restrict__ parse_failed) {
    const uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n) return;
    // after a vector-streaming pass the per-vector array holds these counts themselves (bit 31: the vector
    // ended at a header that did not parse)
    if (tag[1] == VS_MAGIC) {
        const uint32_t x = (uint32_t)vstatus[v];
        valid[v] = x & 0x7fffffffu;
        if (parse_failed) parse_failed[v] = (uint8_t)(x >> 31);
        return;
    }
    // a vector that failed its header checks has only K_SKIP slots
    const uint64_t a = sec_base[v], b = sec_base[v + 1];
    uint64_t s = a;
    bool decode_error = false, stop = false;
    // eight slots per round trip (the loads of a round do not depend on each other)
    while (s < b && !stop) {
        uint64_t d[8];
        uint32_t r[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            d[j] = s + j < b ? desc[s + j] : desc_pack(0, 0, K_SKIP);
            r[j] = s + j < b ? sec_res[s + j] : 0u;
        }
#pragma unroll
        for (int j = 0; j < 8; j++) {
            if (!stop) {
                if (desc_kind(d[j]) == K_SKIP) stop = true;
                else if ((r[j] >> 16) != 0) { decode_error = true; stop = true; }
                else s++;
            }
        }
    }
    valid[v] = (uint32_t)(s - a);
    if (parse_failed) parse_failed[v] = (!decode_error && vstatus[v] != 0) ? 1 : 0;
}

// k_error_detail: for every vector whose status is not OK, the two words the reference's CodingError would carry
// (src/error.rs:2-11) plus where the iteration stopped: detail[2v] = index of the section at which it stopped (all
// ones: the vector header, VectorReader::try_new, src/vector.rs:404-414), detail[2v+1] = the offset of BadOffset, the
// type byte of InvalidSectionType, the subtype byte of WrongVectorType, and for every other code the byte offset of
// that section inside the vector (0 for the header).  One lane per vector; only failed vectors walk their headers
// again, up to the section the pass before stopped at (valid[v], from k_valid_sections).
template <int W>
__global__ __launch_bounds__(256) void k_error_detail(const uint8_t *__restrict__ bytes,
                                                      const uint64_t *__restrict__ byte_off, uint64_t n, int etype,
                                                      const int32_t *__restrict__ status,
                                                      const uint32_t *__restrict__ valid,
                                                      const uint8_t *__restrict__ parse_failed,
                                                      uint64_t *__restrict__ detail) {
    const uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n) return;
    uint64_t d0 = 0, d1 = 0;
    if (status[v] != 0) {
        const uint8_t *p = bytes + byte_off[v];
        const uint64_t L = byte_off[v + 1] - byte_off[v];
        bool header = true;
        if (L == 0 || L < 4) d1 = 0;
        else if (L <= 5) d1 = 5;
        else {
            const uint64_t from_header = (uint64_t)p[0] | ((uint64_t)p[1] << 8) | ((uint64_t)p[2] << 16) | ((uint64_t)p[3] << 24);
            if (L < from_header + 4) d1 = 0;
            else if (p[5] != (uint8_t)etype) d1 = p[5];
            else if (L < (uint64_t)HEADER_BYTES) d1 = 0;
            else header = false;
        }
        if (header) {
            d0 = ~0ull;
        } else {
            uint64_t pos = HEADER_BYTES;
            uint32_t k = 0;
            const uint32_t nv = valid[v];
            while (k < nv && pos < L) { // these headers parsed in the pass before
                const uint32_t b0 = p[pos];
                uint64_t nb = 1;
                if (b0 == K_CONST) nb = 1 + W;
                else if (b0 != K_NULL) {
                    if (L - pos < 3) break;
                    const uint32_t n16 = (uint32_t)p[pos + 1] | ((uint32_t)p[pos + 2] << 8);
                    nb = b0 == K_NP ? (uint64_t)n16 + 3 : (b0 == K_DELTA ? (uint64_t)n16 + DELTA_HDR : n16);
                    if (nb < 3) break;
                }
                pos += nb; k++;
            }
            d0 = k;
            d1 = pos;
            if (parse_failed[v] && pos < L) {
                const uint32_t b0 = p[pos];
                const bool octets = b0 == K_NP || b0 == K_DELTA || b0 == K_XOR;
                if (octets && L - pos <= 1) d1 = 1;                                   // BadOffset(1): the u16 behind the type byte
                else if (!octets && b0 != K_NULL && b0 != K_CONST) d1 = b0;           // InvalidSectionType(byte)
            }
        }
    }
    detail[2 * v] = d0;
    detail[2 * v + 1] = d1;
}

hipError_t launch_error_detail(int etype, const uint8_t *bytes, const uint64_t *byte_off, uint64_t n, const int32_t *status,
                               const uint32_t *valid, const uint8_t *parse_failed, uint64_t *detail, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    const uint32_t blocks = (uint32_t)((n + 255) / 256);
    if (etype == CVEC_U64) k_error_detail<8><<<blocks, 256, 0, stream>>>(bytes, byte_off, n, etype, status, valid, parse_failed, detail);
    else k_error_detail<4><<<blocks, 256, 0, stream>>>(bytes, byte_off, n, etype, status, valid, parse_failed, detail);
    return hipGetLastError();
}

struct MaskColumns {
    const uint8_t *masks[CVEC_MAX_PRED_VALUES];
    const uint64_t *sec_base[CVEC_MAX_PRED_VALUES];
    const uint32_t *valid[CVEC_MAX_PRED_VALUES];
    uint32_t k;
};

// One thread per section slot of column 0: AND the masks of the other columns in order, consulting
// column c only while the running mask is non-zero (the short-circuit of src/filter.rs:269-282).
// `ended` records that a consulted column had no section left, which ends the row's iteration.
__global__ __launch_bounds__(256) void k_multi_and(MaskColumns cols, uint64_t n_rows, uint64_t n_slots0,
                                                   uint8_t *__restrict__ out_masks, uint16_t *__restrict__ popc,
                                                   uint8_t *__restrict__ ended) {
    const uint64_t slot = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= n_slots0) return;
    const uint64_t r = row_of_slot(cols.sec_base[0], n_rows, slot);
    const uint64_t s = slot - cols.sec_base[0][r];
    uint4 m0 = make_uint4(0, 0, 0, 0), m1 = m0;
    bool end = s >= cols.valid[0][r];
    if (!end) {
        const uint4 *p = (const uint4 *)(cols.masks[0] + 32 * slot);
        m0 = p[0]; m1 = p[1];
        for (uint32_t c = 1; c < cols.k; c++) {
            if ((m0.x | m0.y | m0.z | m0.w | m1.x | m1.y | m1.z | m1.w) == 0) break;
            if (s >= cols.valid[c][r]) { end = true; break; }
            const uint4 *q = (const uint4 *)(cols.masks[c] + 32 * (cols.sec_base[c][r] + s));
            const uint4 a = q[0], b = q[1];
            m0.x &= a.x; m0.y &= a.y; m0.z &= a.z; m0.w &= a.w;
            m1.x &= b.x; m1.y &= b.y; m1.z &= b.z; m1.w &= b.w;
        }
    }
    popc[slot] = (uint16_t)(__popc(m0.x) + __popc(m0.y) + __popc(m0.z) + __popc(m0.w) + __popc(m1.x) + __popc(m1.y) +
                            __popc(m1.z) + __popc(m1.w));
    ended[slot] = end ? 1 : 0;
    if (out_masks) {
        uint4 *o = (uint4 *)(out_masks + 32 * slot);
        o[0] = m0; o[1] = m1;
    }
}

// One thread per row: sections up to the first ended one count (src/filter.rs:262-273).
__global__ __launch_bounds__(256) void k_multi_rows(const uint64_t *__restrict__ sec_base0, uint64_t n_rows,
                                                    const uint16_t *__restrict__ popc,
                                                    const uint8_t *__restrict__ ended,
                                                    uint64_t *__restrict__ per_row, uint32_t *__restrict__ row_sections,
                                                    uint64_t *__restrict__ result) {
    __shared__ uint64_t s_cnt[4];
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t cnt = 0;
    if (r < n_rows) {
        const uint64_t a = sec_base0[r], b = sec_base0[r + 1];
        uint64_t s = a;
        for (; s < b && !ended[s]; s++) cnt += popc[s];
        if (per_row) per_row[r] = cnt;
        if (row_sections) row_sections[r] = (uint32_t)(s - a);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) cnt += __shfl_down(cnt, d, 64);
    if ((threadIdx.x & 63) == 0) s_cnt[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd((unsigned long long *)&result[0], s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3]);
}

// ---- match_positions ------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_mask_counts(const uint8_t *__restrict__ masks,
                                                     const uint64_t *__restrict__ sec_base, uint64_t n,
                                                     const uint32_t *__restrict__ valid, uint64_t n_slots,
                                                     uint64_t *__restrict__ counts) {
    const uint64_t slot = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= n_slots) return;
    const uint64_t r = row_of_slot(sec_base, n, slot);
    uint32_t c = 0;
    if (slot - sec_base[r] < valid[r]) {
        const uint4 *p = (const uint4 *)(masks + 32 * slot);
        const uint4 a = p[0], b = p[1];
        c = __popc(a.x) + __popc(a.y) + __popc(a.z) + __popc(a.w) + __popc(b.x) + __popc(b.y) + __popc(b.z) + __popc(b.w);
    }
    counts[slot] = c;
}

// Eight threads per section, one per 32-bit mask word: positions are written in increasing order
// (true positions: the reference's loop forgets to advance over all-zero words, src/filter.rs:309).
__global__ __launch_bounds__(256) void k_mask_positions(const uint8_t *__restrict__ masks,
                                                        const uint64_t *__restrict__ sec_base, uint64_t n,
                                                        const uint32_t *__restrict__ valid, uint64_t n_slots,
                                                        const uint64_t *__restrict__ scan, uint32_t *__restrict__ positions,
                                                        uint64_t cap) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t slot = t >> 3;
    const uint32_t w = (uint32_t)t & 7;
    if (slot >= n_slots) return;
    const uint64_t r = row_of_slot(sec_base, n, slot);
    const uint64_t s = slot - sec_base[r];
    if (s >= valid[r]) return;
    const uint32_t *m = (const uint32_t *)(masks + 32 * slot);
    uint64_t o = scan[slot]; // offset of this section's first position in the batch-wide list
    for (uint32_t i = 0; i < w; i++) o += __popc(m[i]);
    uint32_t bits = m[w];
    const uint32_t base = (uint32_t)(256 * s + 32 * w);
    while (bits) {
        const uint32_t b = __ffs((int)bits) - 1;
        if (o < cap) positions[o] = base + b;
        o++;
        bits &= bits - 1;
    }
}

__global__ void k_pos_off(const uint64_t *__restrict__ sec_base, uint64_t n, const uint64_t *__restrict__ scan,
                          uint64_t *__restrict__ pos_off) {
    const uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v <= n) pos_off[v] = scan[sec_base[v]];
}

// k_merge_filter_pass: adds the outcome of one more filter pass (a further bucket of a large OneOf set; the buckets
// hold distinct values, so an element matches in at most one of them) into the outputs of the passes before it:
// per-vector counts and the total add up, per-section masks are OR-ed (OneOfIF::filter_bitmask ORs the masks of the
// set's members, src/filter.rs:167-174).  Statuses are the same in every pass and stay as the first pass left them.
__global__ __launch_bounds__(256) void k_merge_filter_pass(uint64_t n_vectors, uint64_t n_mask_words,
                                                           const uint64_t *__restrict__ tmp_result,
                                                           const uint64_t *__restrict__ tmp_per,
                                                           const uint64_t *__restrict__ tmp_masks,
                                                           uint64_t *__restrict__ result, uint64_t *__restrict__ per,
                                                           uint64_t *__restrict__ masks) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) result[0] += tmp_result[0];
    if (per != nullptr && i < n_vectors) per[i] += tmp_per[i];
    if (masks != nullptr) {
        const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
        for (uint64_t w = i; w < n_mask_words; w += stride) masks[w] |= tmp_masks[w];
    }
}

// ---- launchers ----------------------------------------------------------------------------------------
hipError_t launch_merge_filter_pass(uint64_t n_vectors, uint64_t n_sections, const uint64_t *tmp_result,
                                    const uint64_t *tmp_per, const uint8_t *tmp_masks, uint64_t *result, uint64_t *per,
                                    uint8_t *masks, hipStream_t stream) {
    const uint64_t words = masks ? n_sections * 4 : 0;
    uint64_t items = n_vectors > words ? n_vectors : words;
    if (items == 0) items = 1;
    uint64_t blocks = (items + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    if (blocks * 256 < n_vectors) blocks = (n_vectors + 255) / 256;
    k_merge_filter_pass<<<(uint32_t)blocks, 256, 0, stream>>>(n_vectors, words, tmp_result, tmp_per, (const uint64_t *)tmp_masks,
                                                              result, per, (uint64_t *)masks);
    return hipGetLastError();
}

hipError_t launch_valid_sections(const uint64_t *sec_base, uint64_t n, const uint64_t *desc, const uint32_t *sec_res,
                                 const int32_t *vstatus, const uint64_t *tag, uint32_t *valid, uint8_t *parse_failed,
                                 hipStream_t stream) {
    if (n)
        k_valid_sections<<<(uint32_t)((n + 255) / 256), 256, 0, stream>>>(sec_base, n, desc, sec_res, vstatus, tag, valid,
                                                                          parse_failed);
    return hipGetLastError();
}

hipError_t launch_multi_and(const MaskColumns &cols, uint64_t n_rows, uint64_t n_slots0, uint8_t *out_masks,
                            uint16_t *popc, uint8_t *ended, uint64_t *per_row, uint32_t *row_sections,
                            uint64_t *result, hipStream_t stream) {
    hipError_t e = hipMemsetAsync(result, 0, CVEC_RESULT_WORDS * sizeof(uint64_t), stream);
    if (e != hipSuccess) return e;
    if (n_slots0)
        k_multi_and<<<(uint32_t)((n_slots0 + 255) / 256), 256, 0, stream>>>(cols, n_rows, n_slots0, out_masks, popc, ended);
    if (n_rows)
        k_multi_rows<<<(uint32_t)((n_rows + 255) / 256), 256, 0, stream>>>(cols.sec_base[0], n_rows, popc, ended, per_row,
                                                                          row_sections, result);
    return hipGetLastError();
}

hipError_t launch_match_positions(const uint8_t *masks, const uint64_t *sec_base, uint64_t n, const uint32_t *valid,
                                  uint64_t n_slots, uint64_t *counts, uint64_t *scan, uint64_t *scan_scratch,
                                  uint64_t *pos_off, uint32_t *positions, uint64_t cap, hipStream_t stream) {
    if (n_slots)
        k_mask_counts<<<(uint32_t)((n_slots + 255) / 256), 256, 0, stream>>>(masks, sec_base, n, valid, n_slots, counts);
    hipError_t e = launch_exclusive_scan(counts, n_slots, scan, scan_scratch, stream);
    if (e != hipSuccess) return e;
    if (n_slots)
        k_mask_positions<<<(uint32_t)((8 * n_slots + 255) / 256), 256, 0, stream>>>(masks, sec_base, n, valid, n_slots, scan,
                                                                                  positions, cap);
    k_pos_off<<<(uint32_t)((n + 256) / 256), 256, 0, stream>>>(sec_base, n, scan, pos_off);
    return hipGetLastError();
}

} // namespace cvec
