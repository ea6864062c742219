// topvgpr.hip - a gfx950 hazard: a 64-bit shift that takes its shift count from the LAST VGPR of the wave's
// allocation can read VGPR0 instead (what the ISA prescribes for an out-of-range source register: the 32-bit count is
// range-checked like a 64-bit operand, and the pair would end one register beyond the allocation).
//
// Every lane computes op(count, x) with the count in a named VGPR and checks the result; v0 holds a known value that
// differs from the count, so a wrong result that equals op(v0, x) is counted separately.  The named register is the
// last one of the kernel's allocation (v63 of 64, v79 of 80, v95 of 96, v103 of 104, v127 of 128) or, as the control,
// the one below it with the same allocation.  The check is a divergent branch (EXEC is saved, cleared and restored
// in every iteration): with a branch-free check the fault did not show.  Dynamic LDS limits the workgroups per CU.
//   hipcc -w -O3 --offload-arch=gfx950 -o topvgpr topvgpr.hip && ./topvgpr
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define STR_(x) #x
#define STR(x) STR_(x)

enum { OP_LSHR64, OP_LSHL64, OP_ASHR64, OP_LSHR32, OP_ADD32, OP_MUL32, OP_MAD64, N_OPS };
static const char *OPS[N_OPS] = {"v_lshrrev_b64", "v_lshlrev_b64", "v_ashrrev_i64", "v_lshrrev_b32", "v_add_u32", "v_mul_lo_u32", "v_mad_u64_u32"};

__device__ __forceinline__ uint64_t expect(int op, uint32_t s, uint64_t x) {
    switch (op) {
    case OP_LSHR64: return x >> (s & 63u);
    case OP_LSHL64: return x << (s & 63u);
    case OP_ASHR64: return (uint64_t)((int64_t)x >> (s & 63u));
    case OP_LSHR32: return (uint32_t)x >> (s & 31u);
    case OP_ADD32: return (uint32_t)((uint32_t)x + s);
    case OP_MUL32: return (uint32_t)((uint32_t)x * s);
    default: return (uint64_t)s * (uint32_t)x + x;
    }
}

#define ASM64(INSN, REG, TOPREG) asm volatile("v_mov_b32 v0, %3\n\tv_mov_b32 " STR(REG) ", %1\n\ts_nop 1\n\t" INSN " %0, " STR(REG) ", %2" : "=v"(r) : "v"(s), "v"(x), "v"(z) : "v0", STR(REG), STR(TOPREG))
#define ASM32(INSN, REG, TOPREG) asm volatile("v_mov_b32 v0, %3\n\tv_mov_b32 " STR(REG) ", %1\n\ts_nop 1\n\t" INSN " %0, " STR(REG) ", %2" : "=v"(r32) : "v"(s), "v"((uint32_t)x), "v"(z) : "v0", STR(REG), STR(TOPREG))

#define DEFINE_KERNEL(NAME, REG, TOPREG)                                                                              \
    template <int OP>                                                                                                 \
    __global__ __launch_bounds__(256) void NAME(uint32_t *bad, int iters) {                                           \
        extern __shared__ uint32_t dyn_lds[];                                                                         \
        const uint32_t lane = threadIdx.x;                                                                            \
        if (iters < 0) dyn_lds[lane] = lane;                                                                          \
        const uint64_t x = 0x8123456789abcdefull ^ ((uint64_t)lane << 17) ^ blockIdx.x;                               \
        uint32_t wrong = 0, wrong_v0 = 0;                                                                             \
        for (int i = 0; i < iters; i++) {                                                                             \
            const uint32_t s = (uint32_t)(i * 7 + lane) & 31u, z = s + 32u; /* v0 = z never equals the count */       \
            uint64_t r = 0;                                                                                           \
            uint32_t r32 = 0;                                                                                         \
            if (OP == OP_LSHR64) ASM64("v_lshrrev_b64", REG, TOPREG);                                                 \
            if (OP == OP_LSHL64) ASM64("v_lshlrev_b64", REG, TOPREG);                                                 \
            if (OP == OP_ASHR64) ASM64("v_ashrrev_i64", REG, TOPREG);                                                 \
            if (OP == OP_LSHR32) { ASM32("v_lshrrev_b32", REG, TOPREG); r = r32; }                                    \
            if (OP == OP_ADD32) { ASM32("v_add_u32", REG, TOPREG); r = r32; }                                         \
            if (OP == OP_MUL32) { ASM32("v_mul_lo_u32", REG, TOPREG); r = r32; }                                      \
            if (OP == OP_MAD64) asm volatile("v_mov_b32 v0, %4\n\tv_mov_b32 " STR(REG) ", %1\n\ts_nop 1\n\tv_mad_u64_u32 %0, vcc, " STR(REG) ", %2, %3" : "=v"(r) : "v"(s), "v"((uint32_t)x), "v"(x), "v"(z) : "v0", STR(REG), STR(TOPREG), "vcc"); \
            if (r != expect(OP, s, x)) { /* divergent on purpose */                                                   \
                wrong++;                                                                                              \
                if (r == expect(OP, OP == OP_LSHR32 ? z & 31u : z, x)) wrong_v0++;                                    \
            }                                                                                                         \
        }                                                                                                             \
        if (wrong) { atomicAdd(&bad[0], wrong); atomicAdd(&bad[1], wrong_v0); if (blockIdx.x < 256) atomicAdd(&bad[2], wrong); } \
    }

DEFINE_KERNEL(k_top64, v63, v63)
DEFINE_KERNEL(k_ctl64, v62, v63)
DEFINE_KERNEL(k_top80, v79, v79)
DEFINE_KERNEL(k_ctl80, v78, v79)
DEFINE_KERNEL(k_top96, v95, v95)
DEFINE_KERNEL(k_ctl96, v94, v95)
DEFINE_KERNEL(k_top92, v91, v91)   // 92 registers used, 96 allocated (granule of 8): v91 is the last one used, not the last one allocated
DEFINE_KERNEL(k_top90, v89, v89)   // 90 used, 96 allocated
DEFINE_KERNEL(k_top104, v103, v103)
DEFINE_KERNEL(k_ctl104, v102, v103)
DEFINE_KERNEL(k_top128, v127, v127)
DEFINE_KERNEL(k_ctl128, v126, v127)

typedef void (*kern_t)(uint32_t *, int);

static void run(const char *reg, int op, kern_t k, int lds_bytes, uint32_t *d, int blocks_per_cu = 0) {
    (void)hipMemset(d, 0, 64);
    int nb = 0;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k, 256, lds_bytes);
    hipFuncAttributes fa;
    (void)hipFuncGetAttributes(&fa, (const void *)k);
    const int iters = 4000;
    const int blocks = 256 * (blocks_per_cu ? blocks_per_cu : nb);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), lds_bytes, 0, d, iters);
    (void)hipDeviceSynchronize();
    uint32_t h[4] = {0, 0, 0, 0};
    (void)hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("%-14s operand in %-5s of %3d VGPRs, %d workgroups/CU fit, grid %4d: wrong %10u of %llu (%6.3f %%), of them = op(v0, x): %u, in the first 256 workgroups: %u\n",
           OPS[op], reg, fa.numRegs, nb, blocks, h[0], (unsigned long long)blocks * 256 * iters, 100.0 * h[0] / ((double)blocks * 256 * iters), h[1], h[2]);
}

#define RUN_ALL_OPS(NAME, REGNAME)                                                                                   \
    run(REGNAME, 0, NAME<0>, 0, d); run(REGNAME, 1, NAME<1>, 0, d); run(REGNAME, 2, NAME<2>, 0, d); run(REGNAME, 3, NAME<3>, 0, d); \
    run(REGNAME, 4, NAME<4>, 0, d); run(REGNAME, 5, NAME<5>, 0, d); run(REGNAME, 6, NAME<6>, 0, d);

int main() {
    uint32_t *d;
    (void)hipMalloc(&d, 64);
    printf("== every operation: operand in the last register of a 96-VGPR allocation, then in the one below it\n");
    RUN_ALL_OPS(k_top96, "v95")
    RUN_ALL_OPS(k_ctl96, "v94")
    printf("== v_lshrrev_b64 by allocation size (last register, then the one below)\n");
    run("v63", 0, k_top64<0>, 0, d); run("v62", 0, k_ctl64<0>, 0, d);
    run("v79", 0, k_top80<0>, 0, d); run("v78", 0, k_ctl80<0>, 0, d);
    run("v91", 0, k_top92<0>, 0, d); run("v89", 0, k_top90<0>, 0, d);
    run("v103", 0, k_top104<0>, 0, d); run("v102", 0, k_ctl104<0>, 0, d);
    run("v127", 0, k_top128<0>, 0, d); run("v126", 0, k_ctl128<0>, 0, d);
    printf("== v_lshrrev_b64, count in v95 of 96: workgroups per CU limited by LDS (5 fit by registers)\n");
    for (int lds : {0, 40 * 1024, 52 * 1024, 64 * 1024}) run("v95", 0, k_top96<0>, lds, d);
    printf("== v_lshrrev_b64, count in v95 of 96: fewer workgroups launched than fit\n");
    for (int per : {5, 4, 3, 2, 1, 8}) run("v95", 0, k_top96<0>, 0, d, per);
    printf("== v_lshrrev_b64, count in v63 of 64: workgroups per CU limited by LDS (8 fit by registers)\n");
    for (int lds : {0, 26 * 1024, 32 * 1024, 40 * 1024, 64 * 1024}) run("v63", 0, k_top64<0>, lds, d);
    return 0;
}
