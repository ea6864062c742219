// How fast do 2048 persistent waves stream 1 GiB when every wave reads (A) one contiguous 18 KiB span per
// pass, (B) 16 windows of 1152 bytes per pass from 16 streams `stride` bytes apart (each load instruction
// 1 KiB contiguous), (C) the same windows with 64 contiguous bytes per quad and instruction.  Loads are
// buffer_load_dwordx4 nt, 18 in flight per wave, nothing else happens.  Prints GB/s.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef uint32_t __attribute__((__vector_size__(16))) u32x4;

template <int MODE>
__global__ __launch_bounds__(256, 2) void k(const uint8_t *bytes, uint32_t nbytes, uint32_t per_wave, uint32_t stride,
                                            uint32_t misalign, uint32_t *sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t gw = blockIdx.x * 4 + wave;
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)bytes, 0, (int)nbytes, 0x00020000);
    const uint32_t base = gw * per_wave + misalign;
    u32x4 acc = {0, 0, 0, 0};
    u32x4 R[18];
    // the wave's range = 16 streams of per_wave/16 bytes (B, C) or one run (A)
    const uint32_t passes = per_wave / (16 * 1152);
    for (uint32_t t = 0; t < passes; t++) {
#pragma unroll
        for (int i = 0; i < 18; i++) {
            uint32_t off;
            if (MODE == 0) off = base + t * 18432u + 1024u * i + 16u * lane;
            else if (MODE == 1) {
                const uint32_t id = 64u * i + lane, w = id / 72u, pc = id % 72u;
                off = base + w * stride + t * 1152u + 16u * pc;
            } else {
                const uint32_t w = lane >> 2, pc = 4u * i + (lane & 3);
                off = base + w * stride + t * 1152u + 16u * pc;
            }
            R[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 2);
        }
#pragma unroll
        for (int i = 0; i < 18; i++) acc ^= R[i];
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) sink[gw] = 1;
}

int main(int argc, char **argv) {
    const uint32_t nbytes = 1u << 30;
    uint8_t *d; uint32_t *s;
    hipMalloc(&d, nbytes + 4096); hipMalloc(&s, 1 << 16);
    hipMemset(d, 1, nbytes + 4096);
    const uint32_t nwaves = 2048, per_wave = nbytes / nwaves; // 512 KiB per wave
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const char *names[3] = {"A contiguous span per wave", "B 16 windows, 1 KiB per instruction", "C 16 windows, 64 B per quad"};
    for (int mis : {0, 16, 32, 64, 48})
    for (int mode = 0; mode < 2; mode++) {
        for (uint32_t stride : {per_wave / 16, 8192u, 8192u + 64u, 8192u + 32u, 8192u + 16u}) {
            if (mode == 0 && stride != per_wave / 16) continue;
            float best = 1e9f;
            for (int it = 0; it < 6; it++) {
                hipEventRecord(a);
                if (mode == 0) k<0><<<nwaves / 4, 256>>>(d, nbytes + 4096, per_wave, stride, mis, s);
                if (mode == 1) k<1><<<nwaves / 4, 256>>>(d, nbytes + 4096, per_wave, stride, mis, s);
                if (mode == 2) k<2><<<nwaves / 4, 256>>>(d, nbytes + 4096, per_wave, stride, mis, s);
                hipEventRecord(b); hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b);
                if (ms < best) best = ms;
            }
            const double bytes_read = (double)nwaves * (per_wave / (16 * 1152)) * 18432.0;
            printf("%-40s misalign %2d stream stride %7u: %.3f ms  %.0f GB/s\n", names[mode], mis, stride, best, bytes_read / best / 1e6);
        }
    }
    return 0;
}
