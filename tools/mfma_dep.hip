// Cycles per v_mfma_f32_32x32x2_f32 when NACC accumulators take turns (NACC = 1: every instruction
// waits for the one before it), one wave per SIMD on every compute unit.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* cyc, int iters) {
  f32x16 acc[NACC];
  for (int n = 0; n < NACC; n++) for (int i = 0; i < 16; i++) acc[n][i] = 0.f;
  float a[16], b[16];
  unsigned s = threadIdx.x * 2654435761u + 12345u;
  for (int u = 0; u < 16; u++) {
    s = s * 1664525u + 1013904223u; a[u] = ((s >> 8) * (1.0f / 16777216.0f) - 0.5f) * 0.25f;
    s = s * 1664525u + 1013904223u; b[u] = ((s >> 8) * (1.0f / 16777216.0f)) * 0.125f;
  }
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < 48; u++)
      acc[u % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u & 15], b[(u * 7) & 15], acc[u % NACC], 0, 0, 0);
  }
  float r = 0; for (int n = 0; n < NACC; n++) for (int i = 0; i < 16; i++) r += acc[n][i];
  asm volatile("" ::"v"(r));
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int NACC>
void run(float* out, unsigned long long* dc, int iters) {
  for (int rep = 0; rep < 3; rep++) hipLaunchKernelGGL(k<NACC>, dim3(256), dim3(256), 0, 0, out, dc, iters);
  CK(hipDeviceSynchronize());
  std::vector<unsigned long long> c(256);
  CK(hipMemcpy(c.data(), dc, 256 * 8, hipMemcpyDeviceToHost));
  std::sort(c.begin(), c.end());
  printf("%d accumulator(s): %.2f cycles per instruction (median workgroup)\n", NACC, (double)c[128] / (iters * 48.0));
}
int main() {
  float* out; unsigned long long* dc;
  CK(hipMalloc(&out, 256 * 256 * 4)); CK(hipMalloc(&dc, 256 * 8));
  const int iters = 200;
  run<1>(out, dc, iters); run<2>(out, dc, iters); run<3>(out, dc, iters); run<4>(out, dc, iters);
  return 0;
}
