// Microbenchmark: marginal cost of K independent VALU ops next to each MFMA (fp32 32x32x2 and
// 16x16x4, fp16 32x32x16: SHAPE 32, 16, 3216).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
__device__ __forceinline__ unsigned umin32(unsigned a, unsigned b) { return a < b ? a : b; }
__device__ __forceinline__ unsigned umax32(unsigned a, unsigned b) { return a > b ? a : b; }
__device__ __forceinline__ unsigned med3u(unsigned a, unsigned b, unsigned c) { return umin32(umax32(a, b), umax32(umin32(a, b), c)); }
template <int K, int SHAPE>
__global__ __launch_bounds__(512) void kern(float* out, unsigned long long* cyc, int iters) {
  f32x16 acc; f32x4 acc4;
  for (int i = 0; i < 16; i++) acc[i] = 0.f;
  for (int i = 0; i < 4; i++) acc4[i] = 0.f;
  float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
  unsigned v[8];
  for (int i = 0; i < 8; i++) v[i] = threadIdx.x * 2654435761u + i;
  unsigned x = threadIdx.x;
  f16x8 ha, hb;
  for (int i = 0; i < 8; i++) { ha[i] = (_Float16)(threadIdx.x * 1e-3f + i); hb[i] = (_Float16)(1.0f + i * 0.125f); }
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < 16; u++) {
      if (SHAPE == 3217) { asm volatile("" : "+a"(ha)); acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, acc, 0, 0, 0); }   // A operand in AGPRs
      else if (SHAPE == 3216) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, acc, 0, 0, 0);
      else if (SHAPE == 32) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
      else acc4 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc4, 0, 0, 0);
#pragma unroll
      for (int k = 0; k < K; k++) { v[k & 7] = med3u(v[k & 7], v[(k + 1) & 7], x); }
      asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0; for (int i = 0; i < 16; i++) s += acc[i]; for (int i = 0; i < 4; i++) s += acc4[i];
  unsigned q = 0; for (int i = 0; i < 8; i++) q ^= v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s + (float)q;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <typename Kn>
void run(const char* name, Kn k, int K, int threads) {
  int blocks = 256, iters = 2000;
  float* out; unsigned long long* cyc;
  CK(hipMalloc(&out, (size_t)blocks * threads * 4)); CK(hipMalloc(&cyc, blocks * 8));
  hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
  hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
  CK(hipDeviceSynchronize());
  unsigned long long c; CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
  printf("%s K=%d threads=%d: %.1f cycles per (MFMA + K VALU) per wave\n", name, K, threads, c / (double)(iters * 16));
  CK(hipFree(out)); CK(hipFree(cyc));
}
int main() {
  run("32x32x2", kern<0, 32>, 0, 256); run("32x32x2", kern<1, 32>, 1, 256); run("32x32x2", kern<2, 32>, 2, 256);
  run("32x32x2", kern<4, 32>, 4, 256); run("32x32x2", kern<8, 32>, 8, 256); run("32x32x2", kern<16, 32>, 16, 256);
  run("32x32x2", kern<4, 32>, 4, 512); run("32x32x2", kern<8, 32>, 8, 512);
  run("16x16x4", kern<0, 16>, 0, 256); run("16x16x4", kern<1, 16>, 1, 256); run("16x16x4", kern<2, 16>, 2, 256);
  run("16x16x4", kern<4, 16>, 4, 256); run("16x16x4", kern<8, 16>, 8, 256);
  run("16x16x4", kern<2, 16>, 2, 512); run("16x16x4", kern<4, 16>, 4, 512);
  run("f16 32x32x16", kern<0, 3216>, 0, 256); run("f16 32x32x16", kern<1, 3216>, 1, 256); run("f16 32x32x16", kern<2, 3216>, 2, 256);
  run("f16 32x32x16", kern<4, 3216>, 4, 256); run("f16 32x32x16", kern<6, 3216>, 6, 256); run("f16 32x32x16", kern<8, 3216>, 8, 256);
  run("f16 32x32x16", kern<16, 3216>, 16, 256); run("f16 32x32x16", kern<4, 3216>, 4, 512); run("f16 32x32x16", kern<8, 3216>, 8, 512);
  run("f16 32x32x16, A in AGPRs", kern<0, 3217>, 0, 256); run("f16 32x32x16, A in AGPRs", kern<2, 3217>, 2, 256);
  run("f16 32x32x16, A in AGPRs", kern<4, 3217>, 4, 256); run("f16 32x32x16, A in AGPRs", kern<8, 3217>, 8, 256);
  return 0;
}
