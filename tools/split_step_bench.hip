// Microbenchmark of the split-fp16 engine's inner loop in isolation: 16 steps per phase, steps
// alternate between two and one v_mfma_f32_32x32x16_f16 on one accumulator chain, each step with an
// LDS fragment read (two steps ahead, counted wait), the A operands in AGPRs, and a variable
// amount of vector work per step:
//   MODE 0  none
//   MODE 1  K simple independent vector instructions after the matrix instructions
//   MODE 2  the real fold of one register of a finished accumulator (row list + column list)
//   MODE 3  as 2, but the vector instructions placed before the matrix instructions of the step
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)
__device__ __forceinline__ unsigned umin32(unsigned a, unsigned b) { return a < b ? a : b; }
__device__ __forceinline__ unsigned umax32(unsigned a, unsigned b) { return a > b ? a : b; }
__device__ __forceinline__ unsigned med3u(unsigned a, unsigned b, unsigned c) { return umin32(umax32(a, b), umax32(umin32(a, b), c)); }
__device__ __forceinline__ void insert3(unsigned& a, unsigned& b, unsigned& c, unsigned x) { c = med3u(b, c, x); b = med3u(a, b, x); a = a < x ? a : x; }

template <int MODE, int K, int S>
__device__ __forceinline__ void step(f32x16& acc, const f32x4 (&a)[16], f32x4 (&b)[3], unsigned addr, const f32x16& done,
                                     unsigned (&r0)[16], unsigned (&r1)[16], unsigned (&r2)[16], unsigned& c0, unsigned& c1,
                                     unsigned& c2, unsigned (&v)[8], unsigned tv) {
  asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(b[(S + 2) % 3]) : "v"(addr));
  asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
  auto fold = [&] {
    const unsigned gb = __float_as_uint(done[S]);
    insert3(r0[S], r1[S], r2[S], (gb & 0x7FFFFE00u) | tv);
    insert3(c0, c1, c2, (gb & 0x7FFFFF00u) | (unsigned)S);
  };
  if (MODE == 3) { fold(); __builtin_amdgcn_sched_barrier(0); }
  const f16x8 frag = __builtin_bit_cast(f16x8, b[S % 3]);
  if ((S & 1) == 0) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[S]), frag, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[S + 1]), frag, acc, 0, 0, 0);
  } else {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[S - 1]), frag, acc, 0, 0, 0);
  }
  if (MODE == 1) {
#pragma unroll
    for (int k = 0; k < K; k++) v[k & 7] = med3u(v[k & 7], v[(k + 1) & 7], tv);
    asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
  }
  if (MODE == 2) fold();
  __builtin_amdgcn_sched_barrier(0);
}

template <int MODE, int K, int... S>
__device__ __forceinline__ void phase(std::integer_sequence<int, S...>, f32x16& acc, const f32x4 (&a)[16], f32x4 (&b)[3],
                                      unsigned addr, const f32x16& done, unsigned (&r0)[16], unsigned (&r1)[16],
                                      unsigned (&r2)[16], unsigned& c0, unsigned& c1, unsigned& c2, unsigned (&v)[8], unsigned tv) {
  (step<MODE, K, S>(acc, a, b, addr, done, r0, r1, r2, c0, c1, c2, v, tv), ...);
}

template <int MODE, int K>
__global__ __launch_bounds__(256, 1) void kern(float* out, unsigned long long* cyc, int iters) {
  __shared__ float lds[8192];
  for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = i * 1e-4f;
  __syncthreads();
  f32x16 acc0, acc1;
  for (int i = 0; i < 16; i++) { acc0[i] = 0.f; acc1[i] = threadIdx.x + i; }
  f32x4 a0[16], a1[16];
  for (int i = 0; i < 16; i++) for (int e = 0; e < 4; e++) { a0[i][e] = (threadIdx.x + i) * 1e-3f; a1[i][e] = (threadIdx.x * 3 + i) * 1e-3f; }
  for (int i = 0; i < 16; i++) asm volatile("" : "+a"(a0[i]), "+a"(a1[i]));
  unsigned r0[16], r1[16], r2[16], q0[16], q1[16], q2[16], v[8];
  for (int i = 0; i < 16; i++) r0[i] = r1[i] = r2[i] = q0[i] = q1[i] = q2[i] = 0xFFFFFFFFu;
  for (int i = 0; i < 8; i++) v[i] = threadIdx.x * 2654435761u + i;
  unsigned c0 = ~0u, c1 = ~0u, c2 = ~0u;
  unsigned addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)lds + (threadIdx.x & 63) * 16;
  f32x4 b[3];
  asm volatile("ds_read_b128 %0, %1" : "=v"(b[0]) : "v"(addr));
  asm volatile("ds_read_b128 %0, %1 offset:1024" : "=v"(b[1]) : "v"(addr));
  constexpr auto STEPS = std::make_integer_sequence<int, 16>{};
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it++) {
    unsigned tv = (unsigned)it;
    asm volatile("" : "+v"(tv));
    phase<MODE, K>(STEPS, acc0, a0, b, addr, acc1, q0, q1, q2, c0, c1, c2, v, tv);
    phase<MODE, K>(STEPS, acc1, a1, b, addr, acc0, r0, r1, r2, c0, c1, c2, v, tv);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float sm = 0; unsigned q = c0 ^ c1 ^ c2;
  for (int i = 0; i < 16; i++) { sm += acc0[i] + acc1[i]; q ^= r0[i] ^ r1[i] ^ r2[i] ^ q0[i] ^ q1[i] ^ q2[i]; }
  for (int i = 0; i < 8; i++) q ^= v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = sm + (float)q;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <typename Kn>
void run(const char* name, Kn k) {
  int blocks = 256, iters = 500;
  float* out; unsigned long long* cyc;
  CK(hipMalloc(&out, (size_t)blocks * 256 * 4)); CK(hipMalloc(&cyc, blocks * 8));
  hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
  hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
  CK(hipDeviceSynchronize());
  unsigned long long c; CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
  printf("%-52s %.1f cycles per phase of 24 matrix instructions (768 if they alone set the pace)\n", name, c / (double)(iters * 2));
  CK(hipFree(out)); CK(hipFree(cyc));
}
int main() {
  run("no vector work", kern<0, 0>);
  run("1 simple vector instruction per step", kern<1, 1>);
  run("4 simple vector instructions per step", kern<1, 4>);
  run("8 simple vector instructions per step", kern<1, 8>);
  run("real fold (8 per step) after the matrix instructions", kern<2, 0>);
  run("real fold (8 per step) before the matrix instructions", kern<3, 0>);
  return 0;
}
