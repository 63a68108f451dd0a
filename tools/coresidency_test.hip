// Scheduling experiment: can a small latency-bound kernel (20 KiB LDS, <= 128 VGPRs, 256 threads)
// run beside the distance kernel's workgroups (143 KiB LDS, one per compute unit) instead of
// between two distance launches?  Times N steps of
//   A  distance kernel only, one stream
//   B  distance kernel + small kernel alternating on one stream
//   C  distance kernels on one stream, small kernels on a second stream (each waits for its
//      distance kernel through an event), equal priorities
//   D  as C, distance stream created with the highest priority, small-kernel stream the lowest
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <chrono>
#include "../sfm_amd/csrc/bf_kernels.hpp"
using namespace sfm;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

// ~latency-bound stand-in: dependent global loads, 20 KiB of LDS, about 10 us per workgroup
// VREG > 0: the kernel claims that many vector registers (it touches the last one)
template <int LDS_INTS, int VREG = 0>
__global__ __launch_bounds__(256) void k_small(const int* __restrict__ chain, int* __restrict__ out, int hops) {
  __shared__ int pad[LDS_INTS];
  if constexpr (VREG == 104) asm volatile("v_mov_b32 v103, 0" ::: "v103");
  if constexpr (VREG == 64) asm volatile("v_mov_b32 v63, 0" ::: "v63");
  if constexpr (VREG == 128) asm volatile("v_mov_b32 v127, 0" ::: "v127");
  if constexpr (VREG == 144) asm volatile("v_mov_b32 v143, 0" ::: "v143");
  int i = (blockIdx.x * 256 + threadIdx.x) & 0xFFFF;
  for (int h = 0; h < hops; h++) i = chain[i];
  pad[threadIdx.x] = i;
  __syncthreads();
  if (i == -1) out[0] = pad[(threadIdx.x + 1) & 255];
}

int main(int argc, char** argv) {
  const int steps = argc > 1 ? atoi(argv[1]) : 200;
  const int hops = argc > 2 ? atoi(argv[2]) : 6;
  const int na = 8192, nb = 8192, chunk = 1024;
  float *dx, *dy, *dxn, *dyn;
  CK(hipMalloc(&dx, (size_t)na * 512)); CK(hipMalloc(&dy, (size_t)nb * 512));
  CK(hipMalloc(&dxn, (na + 2) * 4)); CK(hipMalloc(&dyn, (nb + 2) * 4));
  std::vector<float> A((size_t)na * 128);
  unsigned s = 12345; auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (s >> 8) * (1.0f / 16777216.0f); };
  for (auto& v : A) v = rnd() * 0.125f;
  CK(hipMemcpy(dx, A.data(), (size_t)na * 512, hipMemcpyHostToDevice));
  for (auto& v : A) v = rnd() * 0.125f;
  CK(hipMemcpy(dy, A.data(), (size_t)nb * 512, hipMemcpyHostToDevice));
  CK(hipMemset(dxn, 0, (na + 2) * 4)); CK(hipMemset(dyn, 0, (nb + 2) * 4));
  hipLaunchKernelGGL(k_norms, dim3(na / NORM_ROWS_PER_BLOCK), dim3(256), 0, 0, dx, dxn, na, na);
  hipLaunchKernelGGL(k_norms, dim3(nb / NORM_ROWS_PER_BLOCK), dim3(256), 0, 0, dy, dyn, nb, nb);
  int nchunks = nb / chunk, nrb = na / BM;
  uint32_t *rowpart, *colpart;
  CK(hipMalloc(&rowpart, (size_t)nchunks * ROWPART_FIELDS * na * 4)); CK(hipMalloc(&colpart, (size_t)nrb * 3 * nb * 4));
  PairDev P{}; P.x = dx; P.xn = dxn; P.y = dy; P.yn = dyn; P.rowpart = rowpart; P.colpart = colpart;
  P.na = na; P.nb = nb; P.na_pad = na; P.nb_pad = nb; P.nchunks = nchunks; P.chunk_cols = chunk;
  std::vector<BlockDev> blocks;
  for (int c = 0; c < nchunks; c++) for (int rb = 0; rb < nrb; rb++) {
    BlockDev b{}; b.pair = 0; b.row0 = rb * BM; b.col0 = c * chunk; b.ncols = chunk; b.chunk = c;
    b.xrows = dx + (size_t)rb * BM * DIM; blocks.push_back(b);
  }
  PairDev* dP; BlockDev* dB;
  CK(hipMalloc(&dP, sizeof(P))); CK(hipMalloc(&dB, sizeof(BlockDev) * blocks.size()));
  CK(hipMemcpy(dP, &P, sizeof(P), hipMemcpyHostToDevice));
  CK(hipMemcpy(dB, blocks.data(), sizeof(BlockDev) * blocks.size(), hipMemcpyHostToDevice));
  std::vector<int> chain(65536);
  for (int i = 0; i < 65536; i++) chain[i] = (int)((i * 40503u + 12345u) & 0xFFFF);
  int *dchain, *dout;
  CK(hipMalloc(&dchain, 65536 * 4)); CK(hipMalloc(&dout, 4));
  CK(hipMemcpy(dchain, chain.data(), 65536 * 4, hipMemcpyHostToDevice));

  int lo_p, hi_p;
  CK(hipDeviceGetStreamPriorityRange(&lo_p, &hi_p));   // lo_p = numerically greatest = lowest priority
  hipStream_t s_main, s_eq, s_hi, s_lo;
  CK(hipStreamCreateWithFlags(&s_main, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s_eq, hipStreamNonBlocking));
  CK(hipStreamCreateWithPriority(&s_hi, hipStreamNonBlocking, hi_p));
  CK(hipStreamCreateWithPriority(&s_lo, hipStreamNonBlocking, lo_p));
  std::vector<hipEvent_t> ev(steps);
  for (auto& evt : ev) CK(hipEventCreateWithFlags(&evt, hipEventDisableTiming));
  hipEvent_t t0, t1; CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
  auto k1 = [&](hipStream_t st) { hipLaunchKernelGGL(k_distance_topk<ENGINE_F32>, dim3(blocks.size()), dim3(256), 0, st, dP, dB); };
  auto small = [&](hipStream_t st) { hipLaunchKernelGGL(k_small<5000>, dim3(512), dim3(256), 0, st, dchain, dout, hops); };
  auto tiny = [&](hipStream_t st) { hipLaunchKernelGGL(k_small<256>, dim3(512), dim3(256), 0, st, dchain, dout, hops); };
  auto few = [&](hipStream_t st) { hipLaunchKernelGGL(k_small<5000>, dim3(128), dim3(256), 0, st, dchain, dout, hops); };
  auto timed = [&](const char* name, auto body) {
    for (int rep = 0; rep < 2; rep++) {
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(t0, s_main));
      CK(hipStreamSynchronize(s_main));
      auto w0 = std::chrono::steady_clock::now();
      body();
      CK(hipDeviceSynchronize());
      double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - w0).count();
      if (rep) printf("%-62s %8.1f us per step\n", name, us / steps);
    }
  };
  printf("priority range: lowest %d, highest %d\n", lo_p, hi_p);
  // the small kernel alone
  timed("small kernel alone, back to back", [&] { for (int i = 0; i < steps; i++) small(s_main); });
  timed("A distance kernel only", [&] { for (int i = 0; i < steps; i++) k1(s_main); });
  timed("A2 distance kernels on two streams in turn, nothing else", [&] { for (int i = 0; i < steps; i++) k1(i & 1 ? s_eq : s_main); });
  timed("B distance + small alternating on one stream", [&] { for (int i = 0; i < steps; i++) { k1(s_main); small(s_main); } });
  timed("C two streams, equal priority", [&] {
    for (int i = 0; i < steps; i++) { k1(s_main); CK(hipEventRecord(ev[i], s_main)); CK(hipStreamWaitEvent(s_eq, ev[i], 0)); small(s_eq); } });
  timed("D two streams, distance high / small low priority", [&] {
    for (int i = 0; i < steps; i++) { k1(s_hi); CK(hipEventRecord(ev[i], s_hi)); CK(hipStreamWaitEvent(s_lo, ev[i], 0)); small(s_lo); } });
  timed("E two streams, events recorded and waited, no small kernel", [&] {
    for (int i = 0; i < steps; i++) { k1(s_hi); CK(hipEventRecord(ev[i], s_hi)); CK(hipStreamWaitEvent(s_lo, ev[i], 0)); } });
  timed("F as D, small kernel with 1 KiB of LDS", [&] {
    for (int i = 0; i < steps; i++) { k1(s_hi); CK(hipEventRecord(ev[i], s_hi)); CK(hipStreamWaitEvent(s_lo, ev[i], 0)); tiny(s_lo); } });
  timed("G as D, small kernel of 128 workgroups only", [&] {
    for (int i = 0; i < steps; i++) { k1(s_hi); CK(hipEventRecord(ev[i], s_hi)); CK(hipStreamWaitEvent(s_lo, ev[i], 0)); few(s_lo); } });
  timed("H small kernels free-running on their own stream (no events)", [&] {
    for (int i = 0; i < steps; i++) { k1(s_hi); small(s_lo); } });
  // arrangements built on start/stop events that ride on the distance kernel's dispatch packet
  std::vector<hipEvent_t> es(steps), ee(steps), et(steps);
  for (int i = 0; i < steps; i++) { CK(hipEventCreate(&es[i])); CK(hipEventCreate(&ee[i])); CK(hipEventCreateWithFlags(&et[i], hipEventDisableTiming)); }
  auto k1x = [&](hipStream_t st, hipEvent_t a, hipEvent_t b) {
    hipExtLaunchKernelGGL(k_distance_topk<ENGINE_F32>, dim3(blocks.size()), dim3(256), 0, st, a, b, 0, dP, dB); };
  hipStream_t lane[2] = {s_main, s_eq};
  timed("I two lanes alternate; distance waits for the other lane's distance; tiny follows on the lane", [&] {
    for (int i = 0; i < steps; i++) { hipStream_t st = lane[i & 1];
      if (i > 0) CK(hipStreamWaitEvent(st, ee[i - 1], 0));
      k1x(st, es[i], ee[i]); tiny(st); } });
  timed("I' as I without the tiny kernel", [&] {
    for (int i = 0; i < steps; i++) { hipStream_t st = lane[i & 1];
      if (i > 0) CK(hipStreamWaitEvent(st, ee[i - 1], 0));
      k1x(st, es[i], ee[i]); } });
  timed("J distance kernels on one stream (stop event only); tiny on a second stream waits for it", [&] {
    for (int i = 0; i < steps; i++) { k1x(s_main, nullptr, ee[i]); CK(hipStreamWaitEvent(s_eq, ee[i], 0)); tiny(s_eq); } });
  timed("K as J with start and stop events", [&] {
    for (int i = 0; i < steps; i++) { k1x(s_main, es[i], ee[i]); CK(hipStreamWaitEvent(s_eq, ee[i], 0)); tiny(s_eq); } });
  timed("L as K; distance i also waits for tail i-2 (event recorded after the tiny kernel)", [&] {
    for (int i = 0; i < steps; i++) {
      if (i >= 2) CK(hipStreamWaitEvent(s_main, et[i - 2], 0));
      k1x(s_main, es[i], ee[i]); CK(hipStreamWaitEvent(s_eq, ee[i], 0)); tiny(s_eq); CK(hipEventRecord(et[i], s_eq)); } });
  auto run_j = [&](const char* name, auto launch_small) {
    timed(name, [&] {
      for (int i = 0; i < steps; i++) { k1x(s_main, nullptr, ee[i]); CK(hipStreamWaitEvent(s_eq, ee[i], 0)); launch_small(s_eq); } });
  };
  run_j("J2 as J, small kernel 3.5 KiB LDS, few registers", [&](hipStream_t st) { hipLaunchKernelGGL((k_small<896, 0>), dim3(256), dim3(256), 0, st, dchain, dout, hops); });
  run_j("J3 as J, small kernel 1 KiB LDS, 64 registers", [&](hipStream_t st) { hipLaunchKernelGGL((k_small<256, 64>), dim3(256), dim3(256), 0, st, dchain, dout, hops); });
  run_j("J4 as J, small kernel 1 KiB LDS, 104 registers", [&](hipStream_t st) { hipLaunchKernelGGL((k_small<256, 104>), dim3(256), dim3(256), 0, st, dchain, dout, hops); });
  run_j("J5 as J, small kernel 1 KiB LDS, 128 registers", [&](hipStream_t st) { hipLaunchKernelGGL((k_small<256, 128>), dim3(256), dim3(256), 0, st, dchain, dout, hops); });
  run_j("J6 as J, small kernel 1 KiB LDS, 144 registers", [&](hipStream_t st) { hipLaunchKernelGGL((k_small<256, 144>), dim3(256), dim3(256), 0, st, dchain, dout, hops); });
  run_j("J7 as J, small kernel 3.5 KiB LDS, 104 registers", [&](hipStream_t st) { hipLaunchKernelGGL((k_small<896, 104>), dim3(256), dim3(256), 0, st, dchain, dout, hops); });
  run_j("J8 as J, small kernel 6 KiB LDS, few registers", [&](hipStream_t st) { hipLaunchKernelGGL((k_small<1536, 0>), dim3(256), dim3(256), 0, st, dchain, dout, hops); });
  run_j("J9 as J, one workgroup of 256 threads, 512 B LDS", [&](hipStream_t st) { hipLaunchKernelGGL((k_small<128, 0>), dim3(1), dim3(256), 0, st, dchain, dout, hops); });
  timed("M as K, start/stop events and nothing else", [&] {
    for (int i = 0; i < steps; i++) { k1x(s_main, es[i], ee[i]); } });
  timed("N as M, stop events only", [&] {
    for (int i = 0; i < steps; i++) { k1x(s_main, nullptr, ee[i]); } });
  { float ms = 0; double sum = 0; for (int i = 0; i < steps; i++) { CK(hipEventElapsedTime(&ms, es[i], ee[i])); sum += ms; }
    printf("  (start->stop of the last arrangement with both: %.1f us per launch)\n", sum / steps * 1e3); }
  return 0;
}
