// Microbenchmark: cycles per step of a dependent LDS byte-chase + short VALU chain (the octet walk),
// one wave alone on a CU, in several variants.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int NSTEP = 32, REP = 64;

template <int V>
__global__ __launch_bounds__(512) void k(uint64_t *out, uint32_t *sink, int active_waves) {
    __shared__ __attribute__((aligned(16))) uint8_t data[40000];
    __shared__ uint32_t tab[16 * 36];
    for (int i = threadIdx.x; i < 40000; i += blockDim.x) data[i] = (i % 34 == 0) ? 0xff : ((i % 34 == 1) ? 0x70 : (uint8_t)i);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (wave >= active_waves) return;
    uint32_t acc = 0;
    if (wave == 0) {
        uint64_t t0 = __builtin_readcyclecounter();
        for (int r = 0; r < REP; r++) {
            uint32_t p = (lane & 15) * 34 * 32;
            uint32_t pa = (uint32_t)(uintptr_t)data + p;
            uint32_t *row = tab + (lane & 15) * 36;
            uint32_t mask, nw;
            if (V == 3) {
                // dependent ds_read_b32 chain only (aligned): pure LDS latency + 1 add
                uint32_t a = (uint32_t)(uintptr_t)data + (lane & 15) * 4;
#pragma unroll 8
                for (int k2 = 0; k2 < NSTEP; k2++) {
                    uint32_t v;
                    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(a) : "memory");
                    a += (v & 4);
                }
                acc += a;
                continue;
            }
            if (V == 4) {
                uint32_t a = (uint32_t)(uintptr_t)data + (lane & 15) * 4;
#pragma unroll 8
                for (int k2 = 0; k2 < NSTEP; k2++) {
                    uint32_t v;
                    asm volatile("ds_read_u8 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(a) : "memory");
                    a += (v & 4);
                }
                acc += a;
                continue;
            }
            asm volatile("ds_read_u8 %0, %2\n\tds_read_u8 %1, %2 offset:1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(mask), "=&v"(nw) : "v"(pa) : "memory");
#pragma unroll 8
            for (int k2 = 0; k2 < NSTEP; k2++) {
                const uint32_t hn = nw >> 4;
                const uint32_t cnt = __popc(mask);
                const uint32_t a = (uint32_t)__umul24(min(mask, 1u), 3u) + (cnt + 2u);
                const uint32_t size = ((uint32_t)__umul24(hn, cnt) + a) >> 1;
                pa += size;
                uint32_t mask_n, nw_n;
                asm volatile("ds_read_u8 %0, %4\n\tds_read_u8 %1, %4 offset:1" : "=&v"(mask_n), "=&v"(nw_n), "+v"(mask), "+v"(nw) : "v"(pa) : "memory");
                if (V != 1) row[k2] = p | (mask << 16) | (nw << 24);
                p += size;
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(mask_n), "+v"(nw_n) : : "memory");
                mask = mask_n; nw = nw_n;
            }
            acc += p;
        }
        uint64_t t1 = __builtin_readcyclecounter();
        if (lane == 0) out[0] = t1 - t0;
    } else {
        // other waves: VALU-busy loop to model decoders sharing the SIMDs
        float f = lane;
        for (int i = 0; i < 40000; i++) f = f * 1.0001f + 0.5f;
        acc = (uint32_t)f;
    }
    if (acc == 0x12345) sink[0] = acc;
}

template <int V> void run(const char *name, int waves, uint64_t *d_out, uint32_t *d_sink) {
    k<V><<<1, 512>>>(d_out, d_sink, waves);
    k<V><<<1, 512>>>(d_out, d_sink, waves);
    CK(hipDeviceSynchronize());
    uint64_t c; CK(hipMemcpy(&c, d_out, 8, hipMemcpyDeviceToHost));
    printf("%-44s waves=%d: %.1f cycles/step (s_memtime ticks)\n", name, waves, (double)c / (REP * NSTEP));
}

int main() {
    uint64_t *d_out; uint32_t *d_sink;
    CK(hipMalloc(&d_out, 8)); CK(hipMalloc(&d_sink, 4));
    for (int w : {1, 8}) {
        run<0>("walk step (2x u8 read, chain, table write)", w, d_out, d_sink);
        run<1>("walk step without the table write", w, d_out, d_sink);
        run<3>("ds_read_b32 chase + and + add", w, d_out, d_sink);
        run<4>("ds_read_u8 chase + and + add", w, d_out, d_sink);
    }
    return 0;
}
