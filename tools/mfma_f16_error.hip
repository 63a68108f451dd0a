// Measures the rounding behaviour of v_mfma_f32_32x32x16_f16 on gfx950: for random operands with
// a wide exponent spread (and for subnormal fp16 inputs) compares D = C + sum_k a_k b_k against
// the exact value in double, in units of u * (|C| + sum |a_k b_k|), u = 2^-24.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void k(const _Float16* A, const _Float16* B, const float* C, float* D) {
  // A[row][16], B[col][16] (k contiguous), C/D[row][col]
  const int l = threadIdx.x, r = l & 31, h = l >> 5;
  f16x8 a, b;
  for (int j = 0; j < 8; j++) { a[j] = A[r * 16 + 8 * h + j]; b[j] = B[r * 16 + 8 * h + j]; }
  f32x16 c;
  for (int reg = 0; reg < 16; reg++) c[reg] = C[((reg & 3) + 8 * (reg >> 2) + 4 * h) * 32 + r];
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  for (int reg = 0; reg < 16; reg++) D[((reg & 3) + 8 * (reg >> 2) + 4 * h) * 32 + r] = c[reg];
}

int main() {
  std::vector<_Float16> A(32 * 16), B(32 * 16);
  std::vector<float> C(1024), D(1024);
  _Float16 *dA, *dB; float *dC, *dD;
  hipMalloc(&dA, A.size() * 2); hipMalloc(&dB, B.size() * 2); hipMalloc(&dC, 4096); hipMalloc(&dD, 4096);
  srand(1);
  auto rnd = [] { return rand() / (double)RAND_MAX; };
  double worst[3] = {0, 0, 0};
  int sub_kept = 0, sub_total = 0;
  for (int mode = 0; mode < 3; mode++) {
    for (int trial = 0; trial < 200; trial++) {
      for (auto& v : A) v = (_Float16)((rnd() - 0.5) * std::ldexp(1.0, mode == 0 ? 0 : (rand() % 20) - 10));
      for (auto& v : B) v = (_Float16)((rnd() - 0.5) * std::ldexp(1.0, mode == 0 ? 0 : (rand() % 20) - 10));
      for (auto& v : C) v = (float)((rnd() - 0.5) * (mode == 2 ? 1e4 : 1.0));
      hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice);
      hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice);
      hipMemcpy(dC, C.data(), 4096, hipMemcpyHostToDevice);
      hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD);
      hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost);
      for (int i = 0; i < 32; i++)
        for (int j = 0; j < 32; j++) {
          double exact = C[i * 32 + j], mag = std::fabs((double)C[i * 32 + j]);
          for (int kk = 0; kk < 16; kk++) {
            double p = (double)(float)A[i * 16 + kk] * (double)(float)B[j * 16 + kk];
            exact += p; mag += std::fabs(p);
          }
          double err = std::fabs((double)D[i * 32 + j] - exact) / (mag * 5.9604644775390625e-8);
          if (err > worst[mode]) worst[mode] = err;
        }
    }
  }
  // subnormal fp16 inputs: a = 2^-20 (subnormal), b = 1024, C = 0 -> exact 2^-10 per term
  for (auto& v : A) v = (_Float16)std::ldexp(1.0, -20);
  for (auto& v : B) v = (_Float16)1024.0;
  for (auto& v : C) v = 0.0f;
  hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(dC, C.data(), 4096, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD);
  hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost);
  for (int i = 0; i < 1024; i++) { sub_total++; if (D[i] == 16 * std::ldexp(1.0, -10)) sub_kept++; }
  printf("{\"worst_err_in_u_times_magnitude\": {\"same_scale\": %.3f, \"spread_exponents\": %.3f, \"large_C\": %.3f}, "
         "\"subnormal_inputs_kept\": %d, \"subnormal_cases\": %d, \"subnormal_result\": %g}\n",
         worst[0], worst[1], worst[2], sub_kept, sub_total, D[0]);
  return 0;
}
