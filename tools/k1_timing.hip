// Diagnostic: per-workgroup clocks of k_distance_topk on a synthetic 8192x8192 pair.
#define SFM_K1_STAMPS 1
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include "../sfm_amd/csrc/bf_kernels.hpp"
using namespace sfm;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
int main(int argc, char** argv) {
  int na = argc > 1 ? atoi(argv[1]) : 8192, nb = argc > 2 ? atoi(argv[2]) : 8192, chunk = argc > 3 ? atoi(argv[3]) : 1024;
  int reps = argc > 4 ? atoi(argv[4]) : 5;
  int engine = argc > 5 ? atoi(argv[5]) : 0;   // 0: fp32 matrix instructions, 1: split fp16
  int na_pad = (na + IMG_PAD - 1) / IMG_PAD * IMG_PAD, nb_pad = (nb + IMG_PAD - 1) / IMG_PAD * IMG_PAD;
  std::vector<float> A((size_t)na * 128), B((size_t)nb * 128);
  unsigned s = 12345; auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (s >> 8) * (1.0f / 16777216.0f); };
  for (auto& v : A) v = rnd() * 0.125f; for (auto& v : B) v = rnd() * 0.125f;
  float *dx, *dy, *dxn, *dyn;
  CK(hipMalloc(&dx, (size_t)na_pad * 512)); CK(hipMalloc(&dy, (size_t)nb_pad * 512));
  CK(hipMalloc(&dxn, (na_pad + 2) * 4)); CK(hipMalloc(&dyn, (nb_pad + 2) * 4));
  CK(hipMemset(dxn, 0, (na_pad + 2) * 4)); CK(hipMemset(dyn, 0, (nb_pad + 2) * 4));
  CK(hipMemset(dx, 0, (size_t)na_pad * 512)); CK(hipMemset(dy, 0, (size_t)nb_pad * 512));
  CK(hipMemcpy(dx, A.data(), (size_t)na * 512, hipMemcpyHostToDevice));
  CK(hipMemcpy(dy, B.data(), (size_t)nb * 512, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_norms, dim3((na_pad + NORM_ROWS_PER_BLOCK - 1) / NORM_ROWS_PER_BLOCK), dim3(256), 0, 0, dx, dxn, na, na_pad);
  hipLaunchKernelGGL(k_norms, dim3((nb_pad + NORM_ROWS_PER_BLOCK - 1) / NORM_ROWS_PER_BLOCK), dim3(256), 0, 0, dy, dyn, nb, nb_pad);
  uint16_t *hx, *hy; float *sx, *sy;
  CK(hipMalloc(&hx, (size_t)na_pad * 512)); CK(hipMalloc(&hy, (size_t)nb_pad * 512));
  CK(hipMalloc(&sx, 8)); CK(hipMalloc(&sy, 8));
  hipLaunchKernelGGL(k_split, dim3((na_pad * 16 + 255) / 256), dim3(256), 0, 0, dx, dxn, hx, sx, na_pad);
  hipLaunchKernelGGL(k_split, dim3((nb_pad * 16 + 255) / 256), dim3(256), 0, 0, dy, dyn, hy, sy, nb_pad);
  int nchunks = (nb_pad + chunk - 1) / chunk, nrb = na_pad / BM;
  uint32_t *rowpart, *colpart;
  size_t rw = (size_t)nchunks * ROWPART_FIELDS * na_pad, cw = (size_t)nrb * 3 * nb_pad;
  CK(hipMalloc(&rowpart, rw * 4)); CK(hipMalloc(&colpart, cw * 4));
  PairDev P{}; P.x = dx; P.xn = dxn; P.y = dy; P.yn = dyn; P.rowpart = rowpart; P.colpart = colpart;
  P.xh = hx; P.yh = hy; P.xscale = sx; P.yscale = sy;
  P.na = na; P.nb = nb; P.na_pad = na_pad; P.nb_pad = nb_pad; P.nchunks = nchunks; P.chunk_cols = chunk;
  std::vector<BlockDev> blocks;
  for (int c = 0; c < nchunks; c++) for (int rb = 0; rb < nrb; rb++) {
    BlockDev b{}; b.pair = 0; b.row0 = rb * BM; b.col0 = c * chunk; b.ncols = std::min(chunk, nb_pad - c * chunk); b.chunk = c; b.xrows = engine ? (const void*)(hx + (size_t)rb * BM * 256) : (const void*)(dx + (size_t)rb * BM * DIM); blocks.push_back(b);
  }
  PairDev* dP; BlockDev* dB; unsigned long long* dS;
  size_t nblk = blocks.size();
  CK(hipMalloc(&dP, sizeof(P))); CK(hipMalloc(&dB, sizeof(BlockDev) * nblk)); CK(hipMalloc(&dS, nblk * 96));
  CK(hipMemcpy(dP, &P, sizeof(P), hipMemcpyHostToDevice)); CK(hipMemcpy(dB, blocks.data(), sizeof(BlockDev) * nblk, hipMemcpyHostToDevice));
  CK(hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &dS, sizeof(dS)));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<unsigned long long> st(nblk * 12);
  for (int rep = 0; rep < reps; rep++) {
    CK(hipEventRecord(e0, 0));
    if (engine) hipLaunchKernelGGL(k_distance_topk<ENGINE_F16X3>, dim3(nblk), dim3(256), 0, 0, dP, dB);
    else hipLaunchKernelGGL(k_distance_topk<ENGINE_F32>, dim3(nblk), dim3(256), 0, 0, dP, dB);
    CK(hipEventRecord(e1, 0));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipMemcpy(st.data(), dS, nblk * 96, hipMemcpyDeviceToHost));
    unsigned long long t0 = ~0ull, t1 = 0, r0 = ~0ull, r1 = 0;
    std::vector<double> dur, start;
    for (size_t b = 0; b < nblk; b++) { t0 = std::min(t0, st[12*b]); t1 = std::max(t1, st[12*b+2]); r0 = std::min(r0, st[12*b+1]); r1 = std::max(r1, st[12*b+3]); }
    for (size_t b = 0; b < nblk; b++) { dur.push_back((double)(st[12*b+2] - st[12*b])); start.push_back((double)(st[12*b] - t0)); }
    std::vector<double> pro, mer;
    for (size_t b = 0; b < nblk; b++) { pro.push_back((double)(st[12*b+4] - st[12*b])); mer.push_back((double)(st[12*b+2] - st[12*b+5])); }
    std::sort(pro.begin(), pro.end()); std::sort(mer.begin(), mer.end());
    { std::vector<double> d; for (size_t b = 0; b < nblk; b++) d.push_back((double)(st[12*b+11]-st[12*b]));
      std::sort(d.begin(), d.end()); printf("   start->block descriptor in registers med %.0f min %.0f max %.0f\n", d[nblk/2], d.front(), d.back()); }
    { std::vector<double> a, b2, c; for (size_t b = 0; b < nblk; b++) { a.push_back((double)(st[12*b+6]-st[12*b])); b2.push_back((double)(st[12*b+7]-st[12*b+6])); c.push_back((double)(st[12*b+8]-st[12*b+7])); }
      std::sort(a.begin(), a.end()); std::sort(b2.begin(), b2.end()); std::sort(c.begin(), c.end());
      printf("   start->A issued med %.0f | A landed med %.0f | A read+scaled med %.0f\n", a[nblk/2], b2[nblk/2], c[nblk/2]); }
    printf("   prologue med %.0f max %.0f | row merge med %.0f max %.0f cycles\n", pro[nblk/2], pro.back(), mer[nblk/2], mer.back());
    { // clock seen by each workgroup: cycle counter against the 100 MHz real-time counter, same compute unit
      std::vector<double> ghz, us;
      for (size_t b = 0; b < nblk; b++) { const double rt = (double)(st[12*b+3] - st[12*b+1]); if (rt > 0) { ghz.push_back((double)(st[12*b+2] - st[12*b]) / rt * 0.1); us.push_back(rt / 100.0); } }
      std::sort(ghz.begin(), ghz.end()); std::sort(us.begin(), us.end());
      if (!ghz.empty()) printf("   per workgroup: cycle counter / real time = %.3f GHz (median; min %.3f max %.3f), lifetime %.1f us median\n", ghz[ghz.size()/2], ghz.front(), ghz.back(), us[us.size()/2]); }
    { std::vector<double> dw, bw; for (size_t b = 0; b < nblk; b++) { dw.push_back((double)(st[12*b+9] >> 32)); bw.push_back((double)(st[12*b+9] & 0xFFFFFFFFull)); }
      std::sort(dw.begin(), dw.end()); std::sort(bw.begin(), bw.end());
      printf("   wave 0, summed over the kernel: matrix sections of the tile pairs med %.0f max %.0f | folding accumulators into the lists med %.0f max %.0f cycles\n", dw[nblk/2], dw.back(), bw[nblk/2], bw.back()); }
    { std::vector<double> hd, wt; for (size_t b = 0; b < nblk; b++) { hd.push_back((double)(st[12*b+10] >> 32)); wt.push_back((double)(st[12*b+10] & 0xFFFFFFFFull)); }
      std::sort(hd.begin(), hd.end()); std::sort(wt.begin(), wt.end());
      printf("   wave 0, summed over the stages: stage entry to first tile pair med %.0f | stage-end waits med %.0f cycles\n", hd[nblk/2], wt[nblk/2]); }
    std::sort(dur.begin(), dur.end()); std::sort(start.begin(), start.end());
    double real_us = (r1 - r0) / 100.0;   // 100 MHz
    printf("rep %d: event %.1f us | stamps: span %llu cyc, real %.1f us -> clock %.3f GHz | WG dur min %.0f med %.0f p90 %.0f max %.0f | start med %.0f p90 %.0f max %.0f | blocks %zu\n",
           rep, ms * 1e3, t1 - t0, real_us, (t1 - t0) / real_us / 1e3, dur.front(), dur[nblk/2], dur[nblk*9/10], dur.back(), start[nblk/2], start[nblk*9/10], start.back(), nblk);
  }
  double tiles = (double)(na / 32) * (nb / 32);
  printf("matrix-instruction cycles per SIMD if perfectly packed: %.0f\n",
         tiles * (engine ? 24 * 32 + 64 : 65 * 64) / 1024);
  return 0;
}
