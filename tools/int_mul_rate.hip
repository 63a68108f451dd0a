// Issue cost of the integer instructions a 64-bit hash is made of, on one wavefront per SIMD:
// cycles per instruction for chains of dependent and of independent instructions.
//   hipcc -O3 --offload-arch=gfx950 -o tools/int_mul_rate tools/int_mul_rate.hip && tools/int_mul_rate
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

constexpr int REPS = 4096, UNROLL = 16;

template <int KIND, bool DEPENDENT>
__global__ __launch_bounds__(64) void k(uint64_t* out, uint32_t seed) {
  uint32_t a[4] = {threadIdx.x + seed, threadIdx.x * 3 + 1, threadIdx.x * 5 + 2, threadIdx.x * 7 + 3};
  uint64_t w[4] = {a[0], a[1], a[2], a[3]};
  const uint32_t c = seed | 0x9E3779B9u;
  const uint64_t t0 = __builtin_readcyclecounter();
  for (int r = 0; r < REPS; r++) {
#pragma unroll
    for (int u = 0; u < UNROLL; u++) {
      const int x = DEPENDENT ? 0 : (u & 3);
      if (KIND == 0) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[x]) : "s"(c));
      if (KIND == 1) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[x]) : "s"(c));
      if (KIND == 2) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(w[x]) : "v"(a[x]), "s"(c) : "vcc");
      if (KIND == 3) asm volatile("v_lshrrev_b64 %0, 7, %0" : "+v"(w[x]));
      if (KIND == 4) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[x]) : "v"(a[(x + 1) & 3]));
      if (KIND == 5) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[x]) : "s"(c));
      if (KIND == 6) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(a[x]) : "v"(a[(x + 1) & 3]));
    }
  }
  const uint64_t t1 = __builtin_readcyclecounter();
  uint64_t s = t1 - t0;
  out[blockIdx.x * 64 + threadIdx.x] = s + ((a[0] ^ a[1] ^ a[2] ^ a[3] ^ w[0] ^ w[1] ^ w[2] ^ w[3]) == 12345 ? 1 : 0);
}

template <int KIND, bool DEP>
static void run(const char* name, uint64_t* d) {
  k<KIND, DEP><<<1024, 64>>>(d, 7);
  hipDeviceSynchronize();
  k<KIND, DEP><<<1024, 64>>>(d, 7);
  hipDeviceSynchronize();
  uint64_t h = 0;
  hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
  printf("%-16s %-12s %6.2f clock ticks per instruction (s_memtime ticks at 100 MHz: see below)\n", name, DEP ? "dependent" : "independent", (double)h / (REPS * UNROLL));
}

int main() {
  uint64_t* d;
  hipMalloc(&d, 1024 * 64 * 8);
#define BOTH(K, NAME) run<K, true>(NAME, d); run<K, false>(NAME, d);
  BOTH(4, "v_xor_b32")
  BOTH(6, "v_add3_u32")
  BOTH(5, "v_mul_u32_u24")
  BOTH(0, "v_mul_lo_u32")
  BOTH(1, "v_mul_hi_u32")
  BOTH(2, "v_mad_u64_u32")
  BOTH(3, "v_lshrrev_b64")
  printf("(ratios to v_xor_b32, which issues in 4 cycles per wavefront, give cycles)\n");
  return 0;
}
