// LDS read microbenchmark: cycles per wave-instruction for different widths / alignments / patterns.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
typedef uint16_t __attribute__((aligned(1))) u16u;
typedef uint32_t __attribute__((aligned(1))) u32u;
typedef uint64_t __attribute__((aligned(1))) u64u;
struct __attribute__((packed, aligned(1))) u128u { uint64_t a, b; };
constexpr int LDSB = 32768;
// MODE: 0 = b32 aligned, 1 = b64 aligned(8), 2 = b64 unaligned (u64u), 3 = read2_b32 (aligned 4, two dwords), 4 = u128 unaligned, 5 = b128 aligned, 6 = u16 unaligned, 7 = u8
template<int MODE>
__global__ __launch_bounds__(256) void kern(const uint32_t* addrs, int iters, uint64_t* out, uint64_t* cyc) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[LDSB + 64];
  for (int i = threadIdx.x; i < (LDSB+64)/4; i += 256) ((uint32_t*)lds)[i] = i * 2654435761u;
  __syncthreads();
  uint32_t a[8];
  for (int k = 0; k < 8; k++) a[k] = addrs[(blockIdx.x % 4) * 256 * 8 + k * 256 + threadIdx.x];
  uint64_t acc = 0;
  uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int k = 0; k < 8; k++) {
      uint32_t p = (a[k] + it * 64) & (LDSB - 1);
      if (MODE == 0) acc += *(const uint32_t*)(lds + (p & ~3u));
      if (MODE == 1) acc += *(const uint64_t*)(lds + (p & ~7u));
      if (MODE == 2) acc += *(const u64u*)(lds + p);
      if (MODE == 3) { const uint32_t* q = (const uint32_t*)(lds + (p & ~3u)); acc += q[0] + ((uint64_t)q[1] << 32); }
      if (MODE == 4) { u128u v = *(const u128u*)(lds + p); acc += v.a ^ v.b; }
      if (MODE == 5) { uint4 v = *(const uint4*)(lds + (p & ~15u)); acc += v.x ^ v.y ^ v.z ^ v.w; }
      if (MODE == 6) acc += *(const u16u*)(lds + p);
      if (MODE == 7) acc += lds[p];
    }
  }
  uint64_t t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * 256 + threadIdx.x] = acc;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template<int MODE> double run(const uint32_t* d_addr, int iters, uint64_t* d_out, uint64_t* d_cyc, int blocks) {
  kern<MODE><<<blocks, 256>>>(d_addr, iters, d_out, d_cyc);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  kern<MODE><<<blocks, 256>>>(d_addr, iters, d_out, d_cyc);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<uint64_t> c(blocks); hipMemcpy(c.data(), d_cyc, blocks * 8, hipMemcpyDeviceToHost);
  double avg = 0; for (auto x : c) avg += x; avg /= blocks;
  // cycles per wave-instruction per CU: 4 blocks/CU resident (1024 blocks over 256 CUs), each block 4 waves
  double insts_per_block = (double)iters * 8 * 4;  // wave-instructions per block
  printf("  mode %d: %.3f ms, block cycles %.0f, => %.2f cycles per wave-instr per block-slot\n", MODE, ms, avg, avg / insts_per_block);
  return ms;
}
int main() {
  const int blocks = 1024, iters = 2000;
  uint32_t* d_addr; uint64_t *d_out, *d_cyc;
  hipMalloc(&d_addr, 4 * 256 * 8 * 4); hipMalloc(&d_out, blocks * 256 * 8); hipMalloc(&d_cyc, blocks * 8);
  std::vector<uint32_t> h(4 * 256 * 8);
  const char* names[] = {"linear stride 8 (lane*8)", "linear stride 8 +1 (misaligned)", "linear stride 4", "random bytes", "decode-like: 8 groups x 8 lanes, 4B apart in group, groups ~130B apart", "decode-like +1 byte"};
  for (int pat = 0; pat < 6; pat++) {
    srand(123);
    for (int b = 0; b < 4; b++) for (int k = 0; k < 8; k++) for (int t = 0; t < 256; t++) {
      uint32_t lane = t & 63, w = t >> 6; uint32_t v;
      if (pat == 0) v = lane * 8 + w * 512 + k * 2048;
      else if (pat == 1) v = lane * 8 + w * 512 + k * 2048 + 1;
      else if (pat == 2) v = lane * 4 + w * 256 + k * 1024;
      else if (pat == 3) v = rand() % LDSB;
      else { uint32_t g = lane >> 3, j = lane & 7; uint32_t base = (w * 8 + (k % 8)) * 1000 + 3 + (b * 7); v = base + (4 * g + (k & 3)) * 33 + 2 + j * 4 + (pat == 5 ? 1 : 0) + ((g * 5 + k) % 3); }
      h[(b * 8 + k) * 256 + t] = v % LDSB;
    }
    hipMemcpy(d_addr, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    printf("pattern %d: %s\n", pat, names[pat]);
    run<0>(d_addr, iters, d_out, d_cyc, blocks); run<1>(d_addr, iters, d_out, d_cyc, blocks); run<2>(d_addr, iters, d_out, d_cyc, blocks);
    run<3>(d_addr, iters, d_out, d_cyc, blocks); run<4>(d_addr, iters, d_out, d_cyc, blocks); run<5>(d_addr, iters, d_out, d_cyc, blocks);
    run<6>(d_addr, iters, d_out, d_cyc, blocks); run<7>(d_addr, iters, d_out, d_cyc, blocks);
  }
  return 0;
}
