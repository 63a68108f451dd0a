// Microbenchmark: cycles per v_mfma_f32_32x32x2_f32 / 16x16x4 for 1, 2, 4 independent accumulators, 1 or 2 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
template <int NACC>
__global__ __launch_bounds__(512) void k32(float* out, unsigned long long* cyc, int iters) {
  f32x16 acc[NACC];
  for (int n = 0; n < NACC; n++) for (int i = 0; i < 16; i++) acc[n][i] = 0.f;
  float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < 16; u++)
#pragma unroll
      for (int n = 0; n < NACC; n++) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[n], 0, 0, 0);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0; for (int n = 0; n < NACC; n++) for (int i = 0; i < 16; i++) s += acc[n][i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int NACC>
__global__ __launch_bounds__(512) void k16(float* out, unsigned long long* cyc, int iters) {
  f32x4 acc[NACC];
  for (int n = 0; n < NACC; n++) for (int i = 0; i < 4; i++) acc[n][i] = 0.f;
  float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < 16; u++)
#pragma unroll
      for (int n = 0; n < NACC; n++) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[n], 0, 0, 0);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0; for (int n = 0; n < NACC; n++) for (int i = 0; i < 4; i++) s += acc[n][i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <typename K>
void run(const char* name, K kern, int nacc, int threads, int blocks, float flop_per_mfma) {
  float* out; unsigned long long* cyc; int iters = 2000;
  CK(hipMalloc(&out, (size_t)blocks * threads * 4)); CK(hipMalloc(&cyc, blocks * 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, out, cyc, 10);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, 0));
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
  CK(hipEventRecord(e1, 0)); CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  unsigned long long c; CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
  double mfma_per_wave = (double)iters * 16 * nacc;
  double waves = (double)blocks * threads / 64;
  printf("%-28s threads %d blocks %d: %.1f cyc/MFMA/wave, %.2f ms, %.1f TFLOP/s, clock %.2f GHz\n", name, threads, blocks, c / mfma_per_wave, ms,
         waves * mfma_per_wave * flop_per_mfma / (ms * 1e-3) / 1e12, c / (ms * 1e-3) / 1e9);
  CK(hipFree(out)); CK(hipFree(cyc));
}
int main() {
  const float F32 = 32.f * 32 * 2 * 2, F16 = 16.f * 16 * 4 * 2;
  run("32x32x2 1acc 1wave/SIMD", k32<1>, 1, 256, 256, F32);
  run("32x32x2 2acc 1wave/SIMD", k32<2>, 2, 256, 256, F32);
  run("32x32x2 4acc 1wave/SIMD", k32<4>, 4, 256, 256, F32);
  run("32x32x2 1acc 2wave/SIMD", k32<1>, 1, 512, 256, F32);
  run("32x32x2 2acc 2wave/SIMD", k32<2>, 2, 512, 256, F32);
  run("16x16x4 1acc 1wave/SIMD", k16<1>, 1, 256, 256, F16);
  run("16x16x4 2acc 1wave/SIMD", k16<2>, 2, 256, 256, F16);
  run("16x16x4 4acc 1wave/SIMD", k16<4>, 4, 256, 256, F16);
  run("16x16x4 1acc 2wave/SIMD", k16<1>, 1, 512, 256, F16);
  run("16x16x4 4acc 2wave/SIMD", k16<4>, 4, 512, 256, F16);
  return 0;
}
