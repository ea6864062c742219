// Store-pattern micro-benchmark: how fast can a wave-per-16-sections kernel write 1 GiB of decoded
// values, depending on how many contiguous bytes a lane / a quad of lanes writes per instruction?
//   A: 8 B per lane, 32 B per quad, quads 1 KiB apart (the section kernel's decode epilogue)
//   B: 16 B per lane, 64 B per quad, quads 1 KiB apart, two instructions per 128 B line
//   C: 16 B per lane, 64 lanes contiguous (1 KiB per instruction)
//   D: like A but nontemporal stores
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

template <int P> __global__ __launch_bounds__(256, 2) void k(uint8_t *out, uint64_t chunks_per_wave, uint32_t seed) {
    const int lane = threadIdx.x & 63;
    const uint64_t wave = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const uint32_t q = lane & 3, sec = lane >> 2;
    uint32_t x = seed + lane;
    for (uint64_t c = 0; c < chunks_per_wave; c++) {
        uint8_t *base = out + (wave * chunks_per_wave + c) * 16384; // 16 sections x 1 KiB
        if (P == 0 || P == 3) {
#pragma unroll 4
            for (int k = 0; k < 32; k++) {
                x = x * 1664525u + 1013904223u;
                uint2 v = make_uint2(x, x ^ k);
                uint2 *p = (uint2 *)(base + sec * 1024 + k * 32 + q * 8);
                typedef uint32_t __attribute__((__vector_size__(8))) u32x2;
                if (P == 3) __builtin_nontemporal_store(u32x2{v.x, v.y}, (u32x2 *)p); else *p = v;
            }
        } else if (P == 1) {
#pragma unroll 4
            for (int k = 0; k < 16; k++) {
                x = x * 1664525u + 1013904223u;
                uint4 v = make_uint4(x, x ^ k, x + 1, x + 2);
                *(uint4 *)(base + sec * 1024 + k * 64 + q * 16) = v;
            }
        } else {
#pragma unroll 4
            for (int k = 0; k < 16; k++) {
                x = x * 1664525u + 1013904223u;
                uint4 v = make_uint4(x, x ^ k, x + 1, x + 2);
                *(uint4 *)(base + k * 1024 + lane * 16) = v;
            }
        }
    }
}

int main() {
    const uint64_t bytes = 1ull << 30;
    uint8_t *d;
    hipMalloc(&d, bytes);
    const uint32_t nwg = 512; // 2 per CU, 4 waves each
    const uint64_t chunks = bytes / 16384, per = chunks / (nwg * 4);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    const char *names[4] = {"A 8B/lane 32B/quad", "B 16B/lane 64B/quad", "C 1KiB/instr", "D like A, nontemporal"};
    for (int p = 0; p < 4; p++) {
        float best = 1e9f;
        for (int it = 0; it < 5; it++) {
            hipEventRecord(a);
            if (p == 0) k<0><<<nwg, 256>>>(d, per, it);
            if (p == 1) k<1><<<nwg, 256>>>(d, per, it);
            if (p == 2) k<2><<<nwg, 256>>>(d, per, it);
            if (p == 3) k<3><<<nwg, 256>>>(d, per, it);
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            if (ms < best) best = ms;
        }
        printf("%-24s %.3f ms  %.0f GB/s\n", names[p], best, bytes / (best * 1e-3) / 1e9);
    }
    return 0;
}
