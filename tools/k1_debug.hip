// Debug harness: runs k_norms + k_distance_topk on one pair read from files and checks that
// every row's two nearest columns (float64 on the host) appear in the row's candidate lists,
// and the same for columns.  Usage: k1_debug A.bin na B.bin nb chunk_cols
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#include "../sfm_amd/csrc/bf_kernels.hpp"
using namespace sfm;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
static std::vector<float> readf(const char* p, size_t n) { std::vector<float> v(n); FILE* f = fopen(p, "rb"); if (!f || fread(v.data(), 4, n, f) != n) { printf("read %s failed\n", p); exit(1);} fclose(f); return v; }
int main(int argc, char** argv) {
  int na = atoi(argv[2]), nb = atoi(argv[4]), chunk = atoi(argv[5]);
  auto A = readf(argv[1], (size_t)na * 128), B = readf(argv[3], (size_t)nb * 128);
  int na_pad = (na + IMG_PAD - 1) / IMG_PAD * IMG_PAD, nb_pad = (nb + IMG_PAD - 1) / IMG_PAD * IMG_PAD;
  float *dx, *dy, *dxn, *dyn;
  CK(hipMalloc(&dx, (size_t)na_pad * 512)); CK(hipMalloc(&dy, (size_t)nb_pad * 512));
  CK(hipMalloc(&dxn, (na_pad + 1) * 4)); CK(hipMalloc(&dyn, (nb_pad + 1) * 4));
  CK(hipMemset(dx, 0, (size_t)na_pad * 512)); CK(hipMemset(dy, 0, (size_t)nb_pad * 512));
  CK(hipMemset(dxn, 0, (na_pad + 1) * 4)); CK(hipMemset(dyn, 0, (nb_pad + 1) * 4));
  CK(hipMemcpy(dx, A.data(), (size_t)na * 512, hipMemcpyHostToDevice));
  CK(hipMemcpy(dy, B.data(), (size_t)nb * 512, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_norms, dim3((na_pad + 3) / 4), dim3(256), 0, 0, dx, dxn, na, na_pad);
  hipLaunchKernelGGL(k_norms, dim3((nb_pad + 3) / 4), dim3(256), 0, 0, dy, dyn, nb, nb_pad);
  int nchunks = (nb_pad + chunk - 1) / chunk, nrb = na_pad / BM;
  uint32_t *rowpart, *colpart;
  size_t rw = (size_t)nchunks * ROWPART_FIELDS * na_pad, cw = (size_t)nrb * 3 * nb_pad;
  CK(hipMalloc(&rowpart, rw * 4)); CK(hipMalloc(&colpart, cw * 4));
  CK(hipMemset(rowpart, 0xEE, rw * 4)); CK(hipMemset(colpart, 0xEE, cw * 4));
  PairDev P{}; P.x = dx; P.xn = dxn; P.y = dy; P.yn = dyn; P.rowpart = rowpart; P.colpart = colpart;
  P.na = na; P.nb = nb; P.na_pad = na_pad; P.nb_pad = nb_pad; P.nchunks = nchunks; P.chunk_cols = chunk;
  std::vector<BlockDev> blocks;
  for (int c = 0; c < nchunks; c++) for (int rb = 0; rb < nrb; rb++) {
    BlockDev b{}; b.pair = 0; b.row0 = rb * BM; b.col0 = c * chunk; b.ncols = std::min(chunk, nb_pad - c * chunk); b.chunk = c; blocks.push_back(b);
  }
  PairDev* dP; BlockDev* dB;
  CK(hipMalloc(&dP, sizeof(P))); CK(hipMalloc(&dB, sizeof(BlockDev) * blocks.size()));
  CK(hipMemcpy(dP, &P, sizeof(P), hipMemcpyHostToDevice)); CK(hipMemcpy(dB, blocks.data(), sizeof(BlockDev) * blocks.size(), hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_distance_topk, dim3(blocks.size()), dim3(256), 0, 0, dP, dB);
  CK(hipDeviceSynchronize());
  std::vector<uint32_t> rp(rw), cp(cw);
  CK(hipMemcpy(rp.data(), rowpart, rw * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(cp.data(), colpart, cw * 4, hipMemcpyDeviceToHost));
  auto d2 = [&](int i, int j) { double s = 0; for (int k = 0; k < 128; k++) { double t = (double)A[(size_t)i * 128 + k] - B[(size_t)j * 128 + k]; s += t * t; } return s; };
  int bad_rows = 0, bad_cols = 0;
  for (int i = 0; i < na; i++) {
    std::vector<std::pair<double, int>> v; for (int j = 0; j < nb; j++) v.push_back({d2(i, j), j});
    std::partial_sort(v.begin(), v.begin() + std::min(2, nb), v.end());
    for (int t = 0; t < std::min(2, nb); t++) {
      bool found = false;
      for (int c = 0; c < nchunks && !found; c++) for (int e = 0; e < NCAND; e++) {
        uint32_t key = rp[((size_t)c * ROWPART_FIELDS + e) * na_pad + i], idx = rp[((size_t)c * ROWPART_FIELDS + NCAND + e) * na_pad + i];
        if (key != SENT && (int)idx == v[t].second) found = true;
      }
      if (!found && bad_rows++ < 10) {
        printf("ROW %d: rank-%d column %d (d2 %.8f) missing. chunk lists:\n", i, t, v[t].second, v[t].first);
        int c = v[t].second / chunk;
        for (int e = 0; e < NCAND; e++) { uint32_t key = rp[((size_t)c * ROWPART_FIELDS + e) * na_pad + i], idx = rp[((size_t)c * ROWPART_FIELDS + NCAND + e) * na_pad + i];
          float kv; uint32_t kk = key & ROW_MASK; std::memcpy(&kv, &kk, 4); printf("   key %08x val %.8f t %u idx %u true d2 %.8f\n", key, kv, key & 0x1FF, idx, idx < (uint32_t)nb ? d2(i, idx) : -1.0); }
        uint32_t bk = rp[((size_t)c * ROWPART_FIELDS + 2 * NCAND) * na_pad + i]; float bv; uint32_t bb = bk & ROW_MASK; std::memcpy(&bv, &bb, 4); printf("   bound %08x %.8f\n", bk, bv);
      }
    }
  }
  for (int j = 0; j < nb; j++) {
    std::vector<std::pair<double, int>> v; for (int i = 0; i < na; i++) v.push_back({d2(i, j), i});
    std::partial_sort(v.begin(), v.begin() + std::min(2, na), v.end());
    for (int t = 0; t < std::min(2, na); t++) {
      bool found = false;
      for (int rb = 0; rb < nrb && !found; rb++) for (int k = 0; k < 3; k++) {
        uint32_t key = cp[((size_t)rb * 3 + k) * nb_pad + j]; if (key == SENT) continue;
        int row = rb * BM + (int)col_key_row(key & ~COL_MASK);
        if (row == v[t].second) found = true;
      }
      if (!found && bad_cols++ < 10) printf("COL %d: rank-%d row %d (d2 %.8f) missing\n", j, t, v[t].second, v[t].first);
    }
  }
  printf("rows with a missing top-2 column: %d ; columns with a missing top-2 row: %d\n", bad_rows, bad_cols);
  return 0;
}
