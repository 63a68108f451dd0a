// Microbenchmark of one "pipeline step": 4 dependent 32x32x2 f32 MFMAs + optional LDS read + NV VALU ops in various placements.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
__device__ __forceinline__ unsigned umin32(unsigned a, unsigned b) { return a < b ? a : b; }
__device__ __forceinline__ unsigned umax32(unsigned a, unsigned b) { return a > b ? a : b; }
__device__ __forceinline__ unsigned med3u(unsigned a, unsigned b, unsigned c) { return umin32(umax32(a, b), umax32(umin32(a, b), c)); }
#define VALU1(i) v[(i) & 7] = med3u(v[(i) & 7], v[((i) + 1) & 7], x)
#define ANCHOR asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]))
// MODE 0: no VALU. 1: 10 VALU after MFMA#1. 2: spread 3,3,2,2. 3: 10 VALU before the 4 MFMAs. 4: 10 VALU after MFMA#4 (same as 3 modulo loop)
// 5: 20 VALU every second step (after MFMA #4).   LDS: 1 = ds_read_b128 + counted wait per step
template <int MODE, int LDS>
__global__ __launch_bounds__(512) void kern(float* out, unsigned long long* cyc, int iters) {
  __shared__ float lds[4096];
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = i * 1e-4f;
  __syncthreads();
  f32x16 acc; for (int i = 0; i < 16; i++) acc[i] = 0.f;
  float a = threadIdx.x * 1e-3f;
  unsigned v[8]; for (int i = 0; i < 8; i++) v[i] = threadIdx.x * 2654435761u + i;
  unsigned x = threadIdx.x;
  unsigned addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)lds + (threadIdx.x & 63) * 16;
  f32x4 b[3]; for (int i = 0; i < 3; i++) for (int e = 0; e < 4; e++) b[i][e] = 1.0f + e;
  if (LDS) { asm volatile("ds_read_b128 %0, %1" : "=v"(b[0]) : "v"(addr)); asm volatile("ds_read_b128 %0, %1 offset:1024" : "=v"(b[1]) : "v"(addr)); }
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int s = 0; s < 12; s++) {
      if (LDS) { asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(b[(s + 2) % 3]) : "v"(addr)); asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); }
      const f32x4 bs = b[s % 3];
      if (MODE == 3) { VALU1(0); VALU1(1); VALU1(2); VALU1(3); VALU1(4); VALU1(5); VALU1(6); VALU1(7); VALU1(8); VALU1(9); ANCHOR; __builtin_amdgcn_sched_barrier(0); }
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bs.x, acc, 0, 0, 0);
      if (MODE == 1) { VALU1(0); VALU1(1); VALU1(2); VALU1(3); VALU1(4); VALU1(5); VALU1(6); VALU1(7); VALU1(8); VALU1(9); ANCHOR; __builtin_amdgcn_sched_barrier(0); }
      if (MODE == 2) { VALU1(0); VALU1(1); VALU1(2); ANCHOR; __builtin_amdgcn_sched_barrier(0); }
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bs.y, acc, 0, 0, 0);
      if (MODE == 2) { VALU1(3); VALU1(4); VALU1(5); ANCHOR; __builtin_amdgcn_sched_barrier(0); }
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bs.z, acc, 0, 0, 0);
      if (MODE == 2) { VALU1(6); VALU1(7); ANCHOR; __builtin_amdgcn_sched_barrier(0); }
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bs.w, acc, 0, 0, 0);
      if (MODE == 2) { VALU1(8); VALU1(9); ANCHOR; }
      if (MODE == 5 && (s & 1)) { VALU1(0); VALU1(1); VALU1(2); VALU1(3); VALU1(4); VALU1(5); VALU1(6); VALU1(7); VALU1(8); VALU1(9);
                                  VALU1(10); VALU1(11); VALU1(12); VALU1(13); VALU1(14); VALU1(15); VALU1(16); VALU1(17); VALU1(18); VALU1(19); ANCHOR; }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float sm = 0; for (int i = 0; i < 16; i++) sm += acc[i];
  unsigned q = 0; for (int i = 0; i < 8; i++) q ^= v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = sm + (float)q;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
}
template <typename Kn>
void run(const char* name, Kn k, int threads) {
  int blocks = 256, iters = 1000;
  float* out; unsigned long long* cyc;
  CK(hipMalloc(&out, (size_t)blocks * threads * 4)); CK(hipMalloc(&cyc, blocks * 64));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, 0));
  hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
  CK(hipEventRecord(e1, 0)); CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  unsigned long long c[8]; CK(hipMemcpy(c, cyc, 64, hipMemcpyDeviceToHost));
  int nw = threads / 64; unsigned long long mx = 0; for (int i = 0; i < nw; i++) mx = c[i] > mx ? c[i] : mx;
  double steps = (double)iters * 12;
  printf("%-44s waves/SIMD %d: slowest wave %.1f cyc/step; SIMD cycles per step-of-one-wave %.1f ; kernel %.3f ms\n", name, nw / 4, mx / steps, mx / steps / (nw / 4), ms);
  CK(hipFree(out)); CK(hipFree(cyc));
}
int main() {
  run("no VALU, no LDS", kern<0, 0>, 256);
  run("no VALU, LDS read+wait", kern<0, 1>, 256);
  run("10 VALU after MFMA1, no LDS", kern<1, 0>, 256);
  run("10 VALU after MFMA1, LDS", kern<1, 1>, 256);
  run("10 VALU spread 3/3/2/2, LDS", kern<2, 1>, 256);
  run("10 VALU before MFMAs, LDS", kern<3, 1>, 256);
  run("20 VALU every 2nd step, LDS", kern<5, 1>, 256);
  run("no VALU, LDS  (2 waves)", kern<0, 1>, 512);
  run("10 VALU after MFMA1, LDS (2 waves)", kern<1, 1>, 512);
  run("10 VALU spread, LDS (2 waves)", kern<2, 1>, 512);
  run("10 VALU before MFMAs, LDS (2 waves)", kern<3, 1>, 512);
  run("20 VALU every 2nd step, LDS (2 waves)", kern<5, 1>, 512);
  return 0;
}
