// How many fp32 MFMA FLOP/s the chip sustains over seconds (not a burst of milliseconds): a loop of
// v_mfma_f32_32x32x2_f32 (4 accumulators, one wave per SIMD) launched back to back, rate per 200 ms.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
__global__ __launch_bounds__(256) void k(float* out, int iters, int varied) {
  f32x16 acc[4];
  for (int n = 0; n < 4; n++) for (int i = 0; i < 16; i++) acc[n][i] = 0.f;
  // varied != 0: sixteen different operand pairs of unrelated values per lane, as a real contraction
  // feeds the multipliers; varied == 0: one constant pair (the multipliers' inputs never change)
  float a[16], b[16];
  unsigned s = threadIdx.x * 2654435761u + 12345u;
  for (int u = 0; u < 16; u++) {
    s = s * 1664525u + 1013904223u; a[u] = varied ? ((s >> 8) * (1.0f / 16777216.0f) - 0.5f) * 0.25f : threadIdx.x * 1e-3f;
    s = s * 1664525u + 1013904223u; b[u] = varied ? ((s >> 8) * (1.0f / 16777216.0f)) * 0.125f : 1.0f + threadIdx.x * 1e-4f;
  }
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < 16; u++)
#pragma unroll
      for (int n = 0; n < 4; n++) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[(u + n) & 15], acc[n], 0, 0, 0);
  }
  float s = 0; for (int n = 0; n < 4; n++) for (int i = 0; i < 16; i++) s += acc[n][i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main(int argc, char** argv) {
  const double seconds = argc > 1 ? atof(argv[1]) : 3.0;
  const int varied = argc > 2 ? atoi(argv[2]) : 1;
  float* out; CK(hipMalloc(&out, 256 * 256 * 4));
  const int iters = 500;   // 500 * 64 MFMAs * 64 cycles = 2 M cycles, about 1 ms per launch
  const double flop_per_launch = 256.0 * 4 * iters * 64 * 4096.0;
  auto t0 = std::chrono::steady_clock::now();
  auto w0 = t0; int launches = 0, window = 0;
  while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
    for (int i = 0; i < 20; i++) hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, 0, out, iters, varied);
    CK(hipDeviceSynchronize());
    launches += 20;
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - w0).count();
    if (dt >= 0.2) {
      printf("window %2d: %.1f TFLOP/s (%.3f of 157.3)\n", window++, launches * flop_per_launch / dt / 1e12, launches * flop_per_launch / dt / 157.3e12);
      launches = 0; w0 = std::chrono::steady_clock::now();
    }
  }
  return 0;
}
