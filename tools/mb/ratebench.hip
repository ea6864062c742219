// Issue rate of a few VALU instructions on one wave per SIMD: cycles per wave-instruction.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define REP8(X) X X X X X X X X
template <int OP> __global__ void k(uint32_t *out, uint32_t seed, long long *cycles) {
    uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, s = (seed & 7) + 1;
    uint64_t b0 = a0, b1 = a1, b2 = a2, b3 = a3;
    const long long t0 = clock64();
    for (int i = 0; i < 256; i++) {
        if (OP == 0) { REP8(asm volatile("v_alignbit_b32 %0, %0, %1, %2\n\tv_alignbit_b32 %1, %1, %0, %2" : "+v"(a0), "+v"(a1) : "v"(s)); asm volatile("v_alignbit_b32 %0, %0, %1, %2\n\tv_alignbit_b32 %1, %1, %0, %2" : "+v"(a2), "+v"(a3) : "v"(s));) }
        if (OP == 1) { REP8(asm volatile("v_lshrrev_b64 %0, %2, %0\n\tv_lshrrev_b64 %1, %2, %1" : "+v"(b0), "+v"(b1) : "v"(s)); asm volatile("v_lshrrev_b64 %0, %2, %0\n\tv_lshrrev_b64 %1, %2, %1" : "+v"(b2), "+v"(b3) : "v"(s));) }
        if (OP == 2) { REP8(asm volatile("v_mul_u32_u24 %0, %0, %2\n\tv_mul_u32_u24 %1, %1, %2" : "+v"(a0), "+v"(a1) : "v"(s)); asm volatile("v_mul_u32_u24 %0, %0, %2\n\tv_mul_u32_u24 %1, %1, %2" : "+v"(a2), "+v"(a3) : "v"(s));) }
        if (OP == 3) { REP8(asm volatile("v_bcnt_u32_b32 %0, %0, %2\n\tv_bcnt_u32_b32 %1, %1, %2" : "+v"(a0), "+v"(a1) : "v"(s)); asm volatile("v_bcnt_u32_b32 %0, %0, %2\n\tv_bcnt_u32_b32 %1, %1, %2" : "+v"(a2), "+v"(a3) : "v"(s));) }
        if (OP == 4) { REP8(asm volatile("v_bfe_i32 %0, %0, %2, 1\n\tv_bfe_i32 %1, %1, %2, 1" : "+v"(a0), "+v"(a1) : "v"(s)); asm volatile("v_bfe_i32 %0, %0, %2, 1\n\tv_bfe_i32 %1, %1, %2, 1" : "+v"(a2), "+v"(a3) : "v"(s));) }
        if (OP == 5) { REP8(asm volatile("v_add3_u32 %0, %0, %1, %2\n\tv_add3_u32 %1, %1, %0, %2" : "+v"(a0), "+v"(a1) : "v"(s)); asm volatile("v_add3_u32 %0, %0, %1, %2\n\tv_add3_u32 %1, %1, %0, %2" : "+v"(a2), "+v"(a3) : "v"(s));) }
        if (OP == 6) { REP8(asm volatile("v_mad_u32_u24 %0, %0, %2, %1\n\tv_mad_u32_u24 %1, %1, %2, %0" : "+v"(a0), "+v"(a1) : "v"(s)); asm volatile("v_mad_u32_u24 %0, %0, %2, %1\n\tv_mad_u32_u24 %1, %1, %2, %0" : "+v"(a2), "+v"(a3) : "v"(s));) }
        if (OP == 7) { REP8(asm volatile("v_xor_b32 %0, %0, %1\n\tv_xor_b32 %1, %1, %0" : "+v"(a0), "+v"(a1)); asm volatile("v_xor_b32 %0, %0, %1\n\tv_xor_b32 %1, %1, %0" : "+v"(a2), "+v"(a3));) }
    }
    const long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + (uint32_t)(b0 + b1 + b2 + b3);
    if (threadIdx.x == 0 && blockIdx.x == 0) *cycles = t1 - t0;
}

int main() {
    uint32_t *d; long long *c, h;
    hipMalloc(&d, 1 << 22); hipMalloc(&c, 8);
    const char *names[8] = {"v_alignbit_b32", "v_lshrrev_b64", "v_mul_u32_u24", "v_bcnt_u32_b32", "v_bfe_i32", "v_add3_u32", "v_mad_u32_u24", "v_xor_b32"};
    for (int op = 0; op < 8; op++) {
        for (int it = 0; it < 2; it++) {
            // 1024 workgroups of one wave: at most one wave per SIMD at a time on 256 CUs
            switch (op) {
            case 0: k<0><<<1024, 64>>>(d, it, c); break; case 1: k<1><<<1024, 64>>>(d, it, c); break;
            case 2: k<2><<<1024, 64>>>(d, it, c); break; case 3: k<3><<<1024, 64>>>(d, it, c); break;
            case 4: k<4><<<1024, 64>>>(d, it, c); break; case 5: k<5><<<1024, 64>>>(d, it, c); break;
            case 6: k<6><<<1024, 64>>>(d, it, c); break; default: k<7><<<1024, 64>>>(d, it, c); break;
            }
            hipDeviceSynchronize();
        }
        hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
        // clock64 counts at a fixed 100 MHz on this family: report relative to v_xor_b32 at the end
        printf("%-16s %lld ticks per 8192 instructions\n", names[op], h);
    }
    return 0;
}
