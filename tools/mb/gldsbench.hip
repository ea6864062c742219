// Microbenchmark / semantics check of LDS-DMA loads (global_load_lds_dwordx4) on gfx950:
// where the 64 x 16 bytes of one wave instruction land for LDS destinations above 64 KiB, and the
// rate at which one loader wave per workgroup streams a buffer through an LDS ring.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int LDS_BYTES = 80 * 1024;

__device__ inline void glds16(const void *gsrc, uint32_t lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

__global__ __launch_bounds__(512) void k_sem(const uint8_t *src, uint8_t *out, uint32_t lds_off) {
    __shared__ __attribute__((aligned(16))) uint8_t buf[LDS_BYTES];
    for (int i = threadIdx.x; i < LDS_BYTES / 4; i += 512) ((uint32_t *)buf)[i] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (wave == 7) {
        const uint32_t dst = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)buf + lds_off);
        glds16(src + 16 * (63 - lane), dst); // reversed source order: shows the per-lane placement
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    for (int i = threadIdx.x; i < LDS_BYTES / 4; i += 512) ((uint32_t *)out)[i] = ((uint32_t *)buf)[i];
}

// streaming: each workgroup's wave 7 streams its slice through a ring of `slots` slots of 18 KiB,
// `ahead` slots in flight; the other waves only meet it at the barrier (and checksum one dword per slot).
constexpr int SLOT = 18 * 1024;
__global__ __launch_bounds__(512, 4) void k_stream(const uint8_t *src, uint64_t bytes_per_wg, uint32_t *sink, int nslots) {
    __shared__ __attribute__((aligned(16))) uint8_t buf[4 * SLOT];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint8_t *base = src + (uint64_t)blockIdx.x * bytes_per_wg;
    const int nsteps = (int)(bytes_per_wg / SLOT);
    uint32_t acc = 0;
    for (int t = 0; t < nsteps + 2; t++) {
        if (wave == 7) {
            if (t < nsteps) {
                const uint8_t *g = base + (uint64_t)t * SLOT + lane * 16;
                const uint32_t dst = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)buf + (t % nslots) * SLOT);
#pragma unroll
                for (int i = 0; i < 18; i++) glds16(g + i * 1024, dst + i * 1024);
            } else {
#pragma unroll
                for (int i = 0; i < 18; i++) glds16(base + lane * 16, (uint32_t)(uintptr_t)buf + (t % nslots) * SLOT + i * 1024);
            }
            asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
        }
        __syncthreads();
        if (t >= 1 && t - 1 < nsteps) acc += ((uint32_t *)(buf + ((t - 1) % nslots) * SLOT))[threadIdx.x];
    }
    if (wave == 7) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (acc == 0x12345678) sink[0] = acc;
}

int main() {
    std::vector<uint8_t> h(1 << 20);
    for (size_t i = 0; i < h.size(); i++) h[i] = (uint8_t)(i * 7 + (i >> 8));
    uint8_t *d_src, *d_out;
    CK(hipMalloc(&d_src, h.size()));
    CK(hipMalloc(&d_out, LDS_BYTES));
    CK(hipMemcpy(d_src, h.data(), h.size(), hipMemcpyHostToDevice));
    std::vector<uint8_t> o(LDS_BYTES);
    const uint32_t offs[] = {0, 1024, 16, 65536, 70000, 80896};
    for (uint32_t off : offs) {
        k_sem<<<1, 512>>>(d_src, d_out, off);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(o.data(), d_out, LDS_BYTES, hipMemcpyDeviceToHost));
        int bad = 0, stray = 0;
        for (int l = 0; l < 64; l++)
            for (int b = 0; b < 16; b++)
                if (o[off + l * 16 + b] != h[16 * (63 - l) + b]) bad++;
        for (int i = 0; i < LDS_BYTES; i++)
            if ((i < (int)off || i >= (int)off + 1024) && o[i]) stray++;
        printf("glds16 lds_off=%u: %d wrong bytes in the expected 1 KiB, %d stray bytes elsewhere\n", off, bad, stray);
    }
    // streaming rate
    const uint64_t per_wg = 112ull * SLOT; // 2 MiB-ish per workgroup
    const int nwg = 512;
    uint8_t *d_big; uint32_t *d_sink;
    CK(hipMalloc(&d_big, per_wg * nwg + (1 << 20)));
    CK(hipMemset(d_big, 1, per_wg * nwg + (1 << 20)));
    CK(hipMalloc(&d_sink, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int nslots = 2; nslots <= 4; nslots++) {
        for (int rep = 0; rep < 2; rep++) {
            CK(hipEventRecord(e0));
            for (int i = 0; i < 10; i++) k_stream<<<nwg, 512>>>(d_big, per_wg, d_sink, nslots);
            CK(hipEventRecord(e1));
            CK(hipDeviceSynchronize());
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) printf("stream ring slots=%d: %.4f ms per launch, %.1f GB/s\n", nslots, ms / 10, per_wg * nwg / (ms / 10 * 1e-3) / 1e9);
        }
    }
    return 0;
}
