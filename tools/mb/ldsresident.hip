// ldsresident.hip - do workgroups that share a CU keep their LDS to themselves?  Every workgroup fills its static LDS
// (24 KiB, like the 6 KiB-per-wave region shape that was removed) with a pattern of its own, keeps the CU busy for a
// while with register-only work, and checks the pattern again, many rounds.  The kernel is built for a chosen VGPR
// count (__launch_bounds__(256, 5) holds it to 96, so five workgroups share a CU; -DNR=... is the number of live
// registers of the recurrence) and launched with 1024 or more workgroups.  It prints what the occupancy query says.
//   hipcc -O3 --offload-arch=gfx950 -DNR=48 -o /tmp/ldsresident tools/mb/ldsresident.hip && /tmp/ldsresident 1280
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#ifndef NV
#define NV 96
#endif
#ifndef LDS_WORDS
#define LDS_WORDS 6144
#endif

#ifndef NR
#define NR (NV - 24)
#endif
__device__ __forceinline__ uint32_t churn(uint32_t me, int rounds, int spin, uint32_t *lds, uint32_t blk, uint32_t &bad, bool touch_lds) {
    uint32_t r[NR];
#pragma unroll
    for (int i = 0; i < NR; i++) r[i] = me * 2654435761u + i;
    for (int round = 0; round < rounds; round++) {
        const uint32_t tagv = (blk << 8) ^ (uint32_t)round * 0x9e3779b9u;
        if (touch_lds) {
            for (int i = threadIdx.x; i < LDS_WORDS; i += 256) lds[i] = tagv + i;
            __syncthreads();
        }
        for (int s = 0; s < spin; s++) {
#pragma unroll
            for (int i = 0; i < NR; i++) r[i] = r[i] * 1664525u + 1013904223u;
        }
        if (touch_lds) {
            __syncthreads();
            for (int i = threadIdx.x; i < LDS_WORDS; i += 256) bad += lds[i] != tagv + i;
            __syncthreads();
        }
    }
    uint32_t h = 0;
#pragma unroll
    for (int i = 0; i < NR; i++) h = (h ^ r[i]) * 16777619u;
    return h;
}

__global__ __launch_bounds__(256, 5) void k(uint32_t *errors, uint32_t *regerrs, int rounds, int spin) {
    __shared__ uint32_t lds[LDS_WORDS];
    const uint32_t me = blockIdx.x * 256u + threadIdx.x;
    uint32_t bad = 0, bad2 = 0;
    // the same register-only recurrence twice, the first time with the LDS pattern written and checked around it
    const uint32_t h1 = churn(me, rounds, spin, lds, blockIdx.x, bad, true);
    const uint32_t h2 = churn(me, rounds, spin, lds, blockIdx.x, bad2, false);
    if (bad) atomicAdd(errors, bad);
    if (h1 != h2) atomicAdd(regerrs, 1u);
}

int main(int argc, char **argv) {
    const int nwg = argc > 1 ? atoi(argv[1]) : 1024;
    uint32_t *d, h[2] = {0, 0};
    hipMalloc(&d, 8);
    hipMemset(d, 0, 8);
    hipFuncAttributes fa;
    hipFuncGetAttributes(&fa, (const void *)k);
    for (int rep = 0; rep < 5; rep++) k<<<nwg, 256>>>(d, d + 1, 20, 200);
    hipError_t e = hipDeviceSynchronize();
    hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
    int occ = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void *)k, 256, 0);
    printf("NV=%d: kernel uses %d VGPRs, %zu B LDS; %d workgroups per CU fit; %d workgroups launched; LDS words changed under a workgroup: %u; register values changed: %u (%s)\n",
           NV, fa.numRegs, fa.sharedSizeBytes, occ, nwg, h[0], h[1], hipGetErrorString(e));
    return 0;
}
