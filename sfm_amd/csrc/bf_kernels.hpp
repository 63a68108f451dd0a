// Device code of the brute-force descriptor matcher for gfx950 (MI355X).
//
// One image pair (X = first image, Y = second image) is handled as the
// product matrix S[i][j] = |x_i - y_j|^2 that is never stored:
//
//   k_distance_topk   S in 128 x chunk blocks on the fp32 matrix cores
//                     (v_mfma_f32_32x32x2_f32), squared distances in the form
//                     |x|^2 + |y|^2 - 2 x.y, and from every accumulator tile the
//                     smallest few entries of each row AND of each column,
//                     kept as 32-bit keys (distance bits with the low mantissa
//                     bits replaced by a local index).  Output: short candidate
//                     lists per row and per column.
//   k_rerank          merges the candidate lists of one query, recomputes the
//                     best four exactly as the reference does (src/metric.hpp:
//                     15-22: sequential fp32 sum of squared differences, then
//                     sqrt), applies the top-2 / ratio rule of
//                     src/feature_matcher/brute_force_matcher.cc:18-38, and
//                     proves from the list bounds that no other column could
//                     have changed the answer; queries where that proof fails
//                     are queued.
//   k_rescan          answers queued queries by the exact full scan.
//   k_cross_compact   cross-check of brute_force_matcher.cc:64-80 and ordered
//                     compaction into the output arrays.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sfm {

constexpr int DIM = 128;             // floats per descriptor
constexpr int BM = 128;              // rows of S per workgroup (4 waves x 32 rows)
constexpr int SN = 64;               // columns of S per LDS stage (2 tiles of 32)
constexpr int IMG_PAD = 128;         // images are padded to a multiple of this many rows
constexpr uint32_t SENT = 0xFFFFFFFFu;   // "empty" key, larger than every real key
constexpr int ROW_IDX_BITS = 9;      // row lists: low bits = 32-column tile number inside the chunk
constexpr uint32_t ROW_MASK = ~((1u << ROW_IDX_BITS) - 1u);
constexpr int MAX_CHUNK_COLS = 32 << ROW_IDX_BITS;   // 16384
constexpr int COL_IDX_BITS = 7;      // column lists: low bits = row inside the 128-row block (permuted)
constexpr uint32_t COL_MASK = ~((1u << COL_IDX_BITS) - 1u);
constexpr float PAD_NORM = 1e30f;    // squared norm given to padding rows: never a near neighbour
constexpr int NCAND = 4;             // candidates re-ranked exactly per query
constexpr int ROWPART_FIELDS = 2 * NCAND + 1;  // keys, column indices, bound key

constexpr int LDS_STAGE_BYTES = SN * DIM * 4;                 // 32 KiB
constexpr int LDS_COLBUF_WORDS = 4 * 2 * 3 * 32;              // per parity: wave, tile, k, lane
constexpr int LDS_OFF_COLBUF = 2 * LDS_STAGE_BYTES;
constexpr int LDS_OFF_NAH = LDS_OFF_COLBUF + 2 * LDS_COLBUF_WORDS * 4;
constexpr int LDS_TOTAL = LDS_OFF_NAH + BM * 4;
constexpr int ROWBUF_STRIDE = 97;    // 96 keys per row + 1 word so that rows fall on different banks

// One image pair as the kernels see it.
struct PairDev {
  const float* x;        // [na_pad][128], padding rows zero
  const float* xn;       // [na_pad] squared norms (padding: PAD_NORM), then [na_pad] = max real norm
  const float* y;
  const float* yn;
  uint32_t* rowpart;     // [nchunks][ROWPART_FIELDS][na_pad]
  uint32_t* colpart;     // [na_pad/BM][3][nb_pad]
  int32_t* fwd_best;     // [na_pad] best column of each row, -1 if the ratio test fails
  float* fwd_dist;       // [na_pad] its exact distance
  int32_t* rev_best;     // [nb_pad]
  float* rev_dist;       // [nb_pad]
  int32_t na, nb, na_pad, nb_pad;
  int32_t nchunks, chunk_cols;
  int32_t out_slot;      // index of this pair in the batch's output arrays
  int32_t pad_;
};

// One workgroup of k_distance_topk.
struct BlockDev {
  int32_t pair;
  int32_t row0;      // first row of the 128-row block
  int32_t col0;      // first column of the chunk
  int32_t ncols;     // columns in the chunk (multiple of SN)
  int32_t chunk;     // chunk number (selects the row-partial slab)
  int32_t pad_[3];
};

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t med3u(uint32_t a, uint32_t b, uint32_t c) {
  uint32_t r;
  asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}

// Insert x into the ascending triple (a <= b <= c), dropping the largest of the four.
__device__ __forceinline__ void insert3(uint32_t& a, uint32_t& b, uint32_t& c, uint32_t x) {
  c = med3u(b, c, x);
  b = med3u(a, b, x);
  a = a < x ? a : x;
}

// Squared norms of the rows of one image, PAD_NORM for padding rows, and the
// largest real norm in xn[n_pad].  One wave per row.
__global__ __launch_bounds__(256) void k_norms(const float* __restrict__ x, float* __restrict__ xn, int n,
                                               int n_pad) {
  int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  int l = threadIdx.x & 63;
  if (row >= n_pad) return;
  if (row >= n) {
    if (l == 0) xn[row] = PAD_NORM;
    return;
  }
  float2 v = reinterpret_cast<const float2*>(x + (size_t)row * DIM)[l];
  float s = v.x * v.x + v.y * v.y;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if (l == 0) {
    xn[row] = s;
    atomicMax(reinterpret_cast<unsigned int*>(xn + n_pad), __float_as_uint(s));
  }
}

// ---------------------------------------------------------------------------
// k_distance_topk
//
// Workgroup = 256 threads = 4 waves; wave w owns rows row0+32w .. +31 of the
// block and keeps them as MFMA A-operands in 64 registers for the whole
// kernel (lane (r, h) holds x[row r][4(2s+h)+e] for s<16, e<4; the k order
// inside the contraction is permuted the same way for both operands).  The Y
// columns of the chunk stream through LDS in stages of 64 rows x 512 B,
// loaded by global_load_lds (no registers), double buffered, one barrier per
// stage.  LDS rows are linear; 16-byte chunk c of row R is stored at chunk
// position c ^ (R & 15) so that the ds_read_b128 of an operand fragment
// (32 rows, same chunk) touches every bank group once.
//
// Accumulators start at -|x_i|^2/2, so after the K loop
//     g = |y_j|^2 - 2*acc = |x_i|^2 + |y_j|^2 - 2 x_i.y_j .
// In the 32x32 accumulator layout lane (c, h) holds column c and rows
// (reg&3) + 8(reg>>2) + 4h.  Column minima are therefore lane-local; row
// minima are spread over the 32 lanes of a half-wave, so every lane keeps a
// private 3-entry list per row over the columns it sees (c, c+32, ...) and the
// lists are merged once, at the end of the chunk.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void stage_load(const float* __restrict__ ysrc, char* ldsbuf, int w, int l) {
  // Wave w fills rows 16w .. 16w+15 of the stage: 8 instructions of 1 KiB (2 rows) each.
  const int pc = l & 31;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const int q = w * 8 + i;
    const int R = 2 * q + (l >> 5);
    const int lc = pc ^ (R & 15);
    const float* src = ysrc + (size_t)R * DIM + lc * 4;
    __builtin_amdgcn_global_load_lds(
        (const __attribute__((address_space(1))) void*)src,
        (__attribute__((address_space(3))) void*)(ldsbuf + q * 1024), 16, 0, 0);
  }
}

__global__ __launch_bounds__(256, 2) void k_distance_topk(const PairDev* __restrict__ pairs,
                                                          const BlockDev* __restrict__ blocks) {
  __shared__ __attribute__((aligned(1024))) char smem[LDS_TOTAL];

  const BlockDev blk = blocks[blockIdx.x];
  const PairDev& P = pairs[blk.pair];
  const int tid = threadIdx.x;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l = tid & 63;
  const int r = l & 31;
  const int h = l >> 5;

  float* nah = reinterpret_cast<float*>(smem + LDS_OFF_NAH);
  uint32_t* colbuf = reinterpret_cast<uint32_t*>(smem + LDS_OFF_COLBUF);

  const float* ybase = P.y + (size_t)blk.col0 * DIM;
  const int nstages = blk.ncols / SN;

  // first stage on its way before anything else
  stage_load(ybase, smem, w, l);

  // A operands: 16 x 16 bytes per lane
  float a[64];
  {
    const f32x4* xrow = reinterpret_cast<const f32x4*>(P.x + (size_t)(blk.row0 + 32 * w + r) * DIM);
#pragma unroll
    for (int s = 0; s < 16; s++) {
      f32x4 v = xrow[2 * s + h];
      a[4 * s + 0] = v.x;
      a[4 * s + 1] = v.y;
      a[4 * s + 2] = v.z;
      a[4 * s + 3] = v.w;
    }
  }
  if (tid < BM) nah[tid] = -0.5f * P.xn[blk.row0 + tid];

  // private row lists: 16 rows x 3 keys
  uint32_t rl0[16], rl1[16], rl2[16];
#pragma unroll
  for (int i = 0; i < 16; i++) rl0[i] = rl1[i] = rl2[i] = SENT;

  const uint32_t lowc = (uint32_t)((w << 5) | (h << 4));
  const int c0 = h ^ (r & 15);
  const float* ynp = P.yn + blk.col0 + r;
  float nb_next0 = ynp[0], nb_next1 = ynp[32];

  __syncthreads();  // stage 0 landed (the barrier's fence drains the LDS-DMA), nah visible

  for (int g = 0; g < nstages; g++) {
    const int cur = g & 1;
    char* buf = smem + cur * LDS_STAGE_BYTES;
    const float nb0 = nb_next0, nb1 = nb_next1;
    if (g + 1 < nstages) {
      stage_load(ybase + (size_t)(g + 1) * SN * DIM, smem + (cur ^ 1) * LDS_STAGE_BYTES, w, l);
      nb_next0 = ynp[(g + 1) * SN];
      nb_next1 = ynp[(g + 1) * SN + 32];
    }
    // column lists of the previous stage: merge the four waves' lists, one thread per column
    if (g > 0 && w == 0) {
      const uint32_t* cb = colbuf + (cur ^ 1) * LDS_COLBUF_WORDS;
      const int ct = l >> 5, c = l & 31;
      uint32_t k0 = cb[((0 * 2 + ct) * 3 + 0) * 32 + c];
      uint32_t k1 = cb[((0 * 2 + ct) * 3 + 1) * 32 + c];
      uint32_t k2 = cb[((0 * 2 + ct) * 3 + 2) * 32 + c];
#pragma unroll
      for (int wv = 1; wv < 4; wv++)
#pragma unroll
        for (int k = 0; k < 3; k++) insert3(k0, k1, k2, cb[((wv * 2 + ct) * 3 + k) * 32 + c]);
      uint32_t* dst = P.colpart + (size_t)(blk.row0 / BM) * 3 * P.nb_pad + blk.col0 + (g - 1) * SN + l;
      dst[0] = k0;
      dst[(size_t)P.nb_pad] = k1;
      dst[2 * (size_t)P.nb_pad] = k2;
    }

#pragma unroll
    for (int ct = 0; ct < 2; ct++) {
      const float nbj = ct == 0 ? nb0 : nb1;
      const uint32_t t = (uint32_t)(g * 2 + ct);
      f32x16 acc;
      {
        const f32x4* np = reinterpret_cast<const f32x4*>(nah + 32 * w + 4 * h);
#pragma unroll
        for (int q = 0; q < 4; q++) {
          f32x4 v = np[2 * q];
          acc[4 * q + 0] = v.x;
          acc[4 * q + 1] = v.y;
          acc[4 * q + 2] = v.z;
          acc[4 * q + 3] = v.w;
        }
      }
      const char* rowp = buf + (32 * ct + r) * (DIM * 4);
#pragma unroll
      for (int s = 0; s < 16; s++) {
        f32x4 b = *reinterpret_cast<const f32x4*>(rowp + ((c0 ^ (2 * s)) << 4));
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[4 * s + 0], b.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[4 * s + 1], b.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[4 * s + 2], b.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[4 * s + 3], b.w, acc, 0, 0, 0);
      }
      uint32_t cl0 = SENT, cl1 = SENT, cl2 = SENT;
#pragma unroll
      for (int reg = 0; reg < 16; reg++) {
        float gq = fmaxf(fmaf(acc[reg], -2.0f, nbj), 0.0f);
        uint32_t gb = __float_as_uint(gq);
        insert3(rl0[reg], rl1[reg], rl2[reg], (gb & ROW_MASK) | t);
        insert3(cl0, cl1, cl2, (gb & COL_MASK) | (uint32_t)reg);
      }
      cl0 |= lowc;
      cl1 |= lowc;
      cl2 |= lowc;
      uint32_t p0 = __shfl_xor(cl0, 32), p1 = __shfl_xor(cl1, 32), p2 = __shfl_xor(cl2, 32);
      insert3(cl0, cl1, cl2, p0);
      insert3(cl0, cl1, cl2, p1);
      insert3(cl0, cl1, cl2, p2);
      if (h == 0) {
        uint32_t* cb = colbuf + cur * LDS_COLBUF_WORDS + ((w * 2 + ct) * 3) * 32 + r;
        cb[0] = cl0;
        cb[32] = cl1;
        cb[64] = cl2;
      }
    }
    __syncthreads();  // next stage landed; everyone done with this one; colbuf[cur] complete
  }

  // column lists of the last stage
  if (w == 0) {
    const int cur = (nstages - 1) & 1;
    const uint32_t* cb = colbuf + cur * LDS_COLBUF_WORDS;
    const int ct = l >> 5, c = l & 31;
    uint32_t k0 = cb[((0 * 2 + ct) * 3 + 0) * 32 + c];
    uint32_t k1 = cb[((0 * 2 + ct) * 3 + 1) * 32 + c];
    uint32_t k2 = cb[((0 * 2 + ct) * 3 + 2) * 32 + c];
#pragma unroll
    for (int wv = 1; wv < 4; wv++)
#pragma unroll
      for (int k = 0; k < 3; k++) insert3(k0, k1, k2, cb[((wv * 2 + ct) * 3 + k) * 32 + c]);
    uint32_t* dst = P.colpart + (size_t)(blk.row0 / BM) * 3 * P.nb_pad + blk.col0 + (nstages - 1) * SN + l;
    dst[0] = k0;
    dst[(size_t)P.nb_pad] = k1;
    dst[2 * (size_t)P.nb_pad] = k2;
  }

  // row lists: through LDS (the stage buffers are free now), one thread per row
  uint32_t* rowbuf = reinterpret_cast<uint32_t*>(smem);
#pragma unroll
  for (int reg = 0; reg < 16; reg++) {
    const int R = 32 * w + (reg & 3) + 8 * (reg >> 2) + 4 * h;
    uint32_t* d = rowbuf + R * ROWBUF_STRIDE + r * 3;
    d[0] = rl0[reg];
    d[1] = rl1[reg];
    d[2] = rl2[reg];
  }
  __syncthreads();
  if (tid < BM) {
    const uint32_t* src = rowbuf + tid * ROWBUF_STRIDE;
    uint32_t v0 = SENT, v1 = SENT, v2 = SENT, v3 = SENT, v4 = SENT;
    uint32_t i0 = 0, i1 = 0, i2 = 0, i3 = 0, i4 = 0;
    uint32_t lanebound = SENT;
    for (int p = 0; p < 32; p++) {
#pragma unroll
      for (int k = 0; k < 3; k++) {
        const uint32_t key = src[p * 3 + k];
        if (k == 2) lanebound = lanebound < key ? lanebound : key;
        if (key < v4) {
          v4 = key;
          i4 = ((key & ~ROW_MASK) << 5) + (uint32_t)p;
          if (v4 < v3) { uint32_t tv = v3; v3 = v4; v4 = tv; uint32_t ti = i3; i3 = i4; i4 = ti; }
          if (v3 < v2) { uint32_t tv = v2; v2 = v3; v3 = tv; uint32_t ti = i2; i2 = i3; i3 = ti; }
          if (v2 < v1) { uint32_t tv = v1; v1 = v2; v2 = tv; uint32_t ti = i1; i1 = i2; i2 = ti; }
          if (v1 < v0) { uint32_t tv = v0; v0 = v1; v1 = tv; uint32_t ti = i0; i0 = i1; i1 = ti; }
        }
      }
    }
    const uint32_t bound = lanebound < v4 ? lanebound : v4;
    uint32_t* dst = P.rowpart + (size_t)blk.chunk * ROWPART_FIELDS * P.na_pad + blk.row0 + tid;
    const size_t st = (size_t)P.na_pad;
    dst[0 * st] = v0;
    dst[1 * st] = v1;
    dst[2 * st] = v2;
    dst[3 * st] = v3;
    dst[4 * st] = (uint32_t)blk.col0 + i0;
    dst[5 * st] = (uint32_t)blk.col0 + i1;
    dst[6 * st] = (uint32_t)blk.col0 + i2;
    dst[7 * st] = (uint32_t)blk.col0 + i3;
    dst[8 * st] = bound;
  }
}

// ---------------------------------------------------------------------------
// Exact arithmetic of the reference.
// ---------------------------------------------------------------------------

// src/metric.hpp:15-21 without the sqrt: one fp32 accumulator, index order,
// subtract / multiply / add each rounded on its own (no fused multiply-add).
__device__ __forceinline__ float ref_sumsq(const float* __restrict__ q, const float* __restrict__ c) {
#pragma clang fp contract(off)
  const f32x4* q4 = reinterpret_cast<const f32x4*>(q);
  const f32x4* c4 = reinterpret_cast<const f32x4*>(c);
  float sum = 0.0f;
#pragma unroll 8
  for (int k = 0; k < DIM / 4; k++) {
    f32x4 u = q4[k], v = c4[k];
    float t;
    t = u.x - v.x; sum = sum + t * t;
    t = u.y - v.y; sum = sum + t * t;
    t = u.z - v.z; sum = sum + t * t;
    t = u.w - v.w; sum = sum + t * t;
  }
  return sum;
}

// Running state of the scan at brute_force_matcher.cc:18-33, in a form that
// does not depend on visiting order: the smallest distance, the lowest index
// attaining it, and the second smallest distance counted with multiplicity.
// (The reference scans j upward with strict '<', which yields exactly that.)
struct Top2 {
  float d1; int32_t i1; float d2;
  float r2;   // second smallest squared sum, for the completeness proof
  float r1;
};
__device__ __forceinline__ void top2_init(Top2& t) {
  t.d1 = 3.402823466e+38f; t.i1 = -1; t.d2 = 3.402823466e+38f;
  t.r1 = __builtin_inff(); t.r2 = __builtin_inff();
}
__device__ __forceinline__ void top2_add(Top2& t, float d, float rsq, int32_t idx) {
  if (d < t.d1 || (d == t.d1 && t.i1 >= 0 && idx < t.i1)) {
    // new minimum (or same value at a lower index): old minimum becomes the second
    t.d2 = t.d1; t.d1 = d; t.i1 = idx;
  } else if (d < t.d2) {
    t.d2 = d;
  }
  if (rsq < t.r1) { t.r2 = t.r1; t.r1 = rsq; }
  else if (rsq < t.r2) t.r2 = rsq;
}
__device__ __forceinline__ void top2_merge(Top2& t, const Top2& o) {
  // fold o's two values in; o.d2 has no index of its own and never needs one
  if (o.i1 >= 0) {
    float od2 = o.d2;
    if (o.d1 < t.d1 || (o.d1 == t.d1 && (t.i1 < 0 || o.i1 < t.i1))) {
      float s = t.d1 < od2 ? t.d1 : od2;   // second = min(old minimum, o's second)
      t.d2 = s < t.d2 ? s : t.d2;
      // when t had a real minimum equal to o.d1 the old minimum is the second: covered by s
      t.d1 = o.d1; t.i1 = o.i1;
    } else {
      float s = o.d1 < t.d2 ? o.d1 : t.d2;
      t.d2 = s;
    }
  }
  // squared sums: merge two sorted pairs
  float a1 = t.r1, a2 = t.r2, b1 = o.r1, b2 = o.r2;
  float m1 = a1 < b1 ? a1 : b1;
  float hi = a1 < b1 ? b1 : a1;
  float lo2 = a2 < b2 ? a2 : b2;
  t.r1 = m1;
  t.r2 = hi < lo2 ? hi : lo2;
}

// Lower bound on the reference's squared sum of any column that is NOT among a
// query's candidates, given the smallest key such a column can have.
//   g  = computed |x|^2+|y|^2-2x.y; key value <= g (truncation and clamp only lower it);
//   |g - t| <= 3.1*gamma128*(nx+ny) + u*g  (norm sums, the 128-term fma chain started at
//              -nx/2, the final fma), t = exact squared distance, u = 2^-24;
//   reference sum r >= t*(1 - 131u).
// The constants below are rounded up generously.
__device__ __forceinline__ double excluded_lower_bound(uint32_t boundkey, uint32_t mask, float nq, float nomax) {
  const double u = 5.9604644775390625e-8;
  double bv = (double)__uint_as_float(boundkey & mask);
  return bv * (1.0 - 160.0 * u) - 450.0 * u * ((double)nq + (double)nomax);
}

// Running top-5 of (key, index) pairs, ascending by key; branch-free insertion.
struct Top5 {
  uint32_t v0, v1, v2, v3, v4;
  uint32_t i0, i1, i2, i3, i4;
};
__device__ __forceinline__ void top5_init(Top5& t) {
  t.v0 = t.v1 = t.v2 = t.v3 = t.v4 = SENT;
  t.i0 = t.i1 = t.i2 = t.i3 = t.i4 = 0;
}
__device__ __forceinline__ void cswap(uint32_t& va, uint32_t& ia, uint32_t& vb, uint32_t& ib) {
  // afterwards va <= vb
  const bool s = vb < va;
  const uint32_t tv = s ? vb : va, ti = s ? ib : ia;
  vb = s ? va : vb;
  ib = s ? ia : ib;
  va = tv;
  ia = ti;
}
__device__ __forceinline__ void top5_put(Top5& t, uint32_t key, uint32_t idx) {
  const bool s = key < t.v4;
  t.v4 = s ? key : t.v4;
  t.i4 = s ? idx : t.i4;
  cswap(t.v3, t.i3, t.v4, t.i4);
  cswap(t.v2, t.i2, t.v3, t.i3);
  cswap(t.v1, t.i1, t.v2, t.i2);
  cswap(t.v0, t.i0, t.v1, t.i1);
}

constexpr int RR_QUERIES = 64;       // queries per workgroup of k_rerank
constexpr int RR_SQ_STRIDE = 132;    // words per row of squares (128 + 4: conflict-free b128 reads)

// k_rerank: grid (ceil(max queries / 64), pairs, directions), 256 threads.
// Direction 0: queries are rows of X, lists come from rowpart, candidates are rows of Y.
// Direction 1: queries are rows of Y, lists come from colpart, candidates are rows of X.
//   A. four lanes per query merge the query's partial lists (a quarter each, then two
//      shuffle rounds) into the best five keys and the bound key;
//   B. the workgroup computes the 128 squared differences of every (query, candidate)
//      with coalesced row loads, through LDS; one lane per (query, candidate) then adds
//      them up in index order and takes the square root: src/metric.hpp:15-22;
//   C. one lane per query applies brute_force_matcher.cc:26-38 to its four candidates
//      and checks the completeness bound;
//   D. queries that failed the check are re-answered by an exact scan of every
//      candidate row, by the whole workgroup.
__global__ __launch_bounds__(256) void k_rerank(const PairDev* __restrict__ pairs, double ratio,
                                                unsigned long long* __restrict__ rescanned) {
  __shared__ __attribute__((aligned(16))) float sq[128 * RR_SQ_STRIDE];
  __shared__ uint32_t cand[RR_QUERIES * NCAND];
  __shared__ uint32_t bnd[RR_QUERIES];
  __shared__ float res_r[RR_QUERIES * NCAND];
  __shared__ int unsafe_q[RR_QUERIES];
  __shared__ int n_unsafe;

  const PairDev& P = pairs[blockIdx.y];
  const int dir = blockIdx.z;
  const int nq = dir == 0 ? P.na : P.nb;
  const int no = dir == 0 ? P.nb : P.na;
  const int q0 = blockIdx.x * RR_QUERIES;
  if (q0 >= nq) return;
  const float* qbase = dir == 0 ? P.x : P.y;
  const float* obase = dir == 0 ? P.y : P.x;
  const int tid = threadIdx.x;
  if (tid == 0) n_unsafe = 0;

  // ---- A
  {
    const int ql = tid >> 2, sub = tid & 3;
    const int q = q0 + ql;
    Top5 t;
    top5_init(t);
    uint32_t bound = SENT;
    if (q < nq) {
      if (dir == 0) {
        const size_t st = (size_t)P.na_pad;
        for (int c = sub; c < P.nchunks; c += 4) {
          const uint32_t* src = P.rowpart + (size_t)c * ROWPART_FIELDS * st + q;
          uint32_t k[NCAND], ix[NCAND];
#pragma unroll
          for (int e = 0; e < NCAND; e++) {
            k[e] = src[e * st];
            ix[e] = src[(NCAND + e) * st];
          }
          const uint32_t b = src[2 * NCAND * st];
#pragma unroll
          for (int e = 0; e < NCAND; e++) top5_put(t, k[e], ix[e]);
          bound = bound < b ? bound : b;
        }
      } else {
        const size_t st = (size_t)P.nb_pad;
        const int nrb = P.na_pad / BM;
        for (int rb0 = sub; rb0 < nrb; rb0 += 16) {
          uint32_t k[4][3];
#pragma unroll
          for (int u = 0; u < 4; u++) {
            const int rb = rb0 + 4 * u;
            const uint32_t* src = P.colpart + (size_t)rb * 3 * st + q;
#pragma unroll
            for (int e = 0; e < 3; e++) k[u][e] = rb < nrb ? src[e * st] : SENT;
          }
#pragma unroll
          for (int u = 0; u < 4; u++) {
            const int rb = rb0 + 4 * u;
#pragma unroll
            for (int e = 0; e < 3; e++) {
              const uint32_t key = k[u][e];
              const uint32_t lo = key & ~COL_MASK;
              const uint32_t reg = lo & 15u, hh = (lo >> 4) & 1u, ww = lo >> 5;
              top5_put(t, key, (uint32_t)rb * BM + 32u * ww + (reg & 3u) + 8u * (reg >> 2) + 4u * hh);
            }
            bound = bound < k[u][2] ? bound : k[u][2];   // a list's last key bounds what it left out
          }
        }
      }
    }
#pragma unroll
    for (int o = 1; o < 4; o <<= 1) {
      Top5 u;
      u.v0 = __shfl_xor(t.v0, o); u.v1 = __shfl_xor(t.v1, o); u.v2 = __shfl_xor(t.v2, o);
      u.v3 = __shfl_xor(t.v3, o); u.v4 = __shfl_xor(t.v4, o);
      u.i0 = __shfl_xor(t.i0, o); u.i1 = __shfl_xor(t.i1, o); u.i2 = __shfl_xor(t.i2, o);
      u.i3 = __shfl_xor(t.i3, o); u.i4 = __shfl_xor(t.i4, o);
      const uint32_t ob = __shfl_xor(bound, o);
      top5_put(t, u.v0, u.i0); top5_put(t, u.v1, u.i1); top5_put(t, u.v2, u.i2);
      top5_put(t, u.v3, u.i3); top5_put(t, u.v4, u.i4);
      bound = bound < ob ? bound : ob;
    }
    bound = bound < t.v4 ? bound : t.v4;
    const uint32_t ck = sub == 0 ? t.v0 : sub == 1 ? t.v1 : sub == 2 ? t.v2 : t.v3;
    const uint32_t ci = sub == 0 ? t.i0 : sub == 1 ? t.i1 : sub == 2 ? t.i2 : t.i3;
    cand[tid] = (ck != SENT && (int)ci < no) ? ci : SENT;
    if (sub == 0) bnd[ql] = bound;
  }
  __syncthreads();

  // ---- B
  {
    const int w = tid >> 6, l = tid & 63, half = l >> 5, j = l & 31;
#pragma unroll 1
    for (int pass = 0; pass < 2; pass++) {
#pragma unroll 4
      for (int it = 0; it < 16; it++) {
#pragma clang fp contract(off)
        const int pl = w * 32 + it * 2 + half;      // (query, candidate) inside this pass
        const int pi = pass * 128 + pl;
        const uint32_t ci = cand[pi];
        if (ci != SENT) {
          const f32x4 u = reinterpret_cast<const f32x4*>(qbase + (size_t)(q0 + (pi >> 2)) * DIM)[j];
          const f32x4 v = reinterpret_cast<const f32x4*>(obase + (size_t)ci * DIM)[j];
          f32x4 d;
          d.x = u.x - v.x; d.y = u.y - v.y; d.z = u.z - v.z; d.w = u.w - v.w;
          d.x = d.x * d.x; d.y = d.y * d.y; d.z = d.z * d.z; d.w = d.w * d.w;
          *reinterpret_cast<f32x4*>(sq + pl * RR_SQ_STRIDE + 4 * j) = d;
        }
      }
      __syncthreads();
      if (tid < 128) {
#pragma clang fp contract(off)
        const int pi = pass * 128 + tid;
        float sum = __builtin_inff();
        if (cand[pi] != SENT) {
          const f32x4* s4 = reinterpret_cast<const f32x4*>(sq + tid * RR_SQ_STRIDE);
          sum = 0.0f;
#pragma unroll 8
          for (int k = 0; k < DIM / 4; k++) {
            const f32x4 v = s4[k];
            sum = sum + v.x;
            sum = sum + v.y;
            sum = sum + v.z;
            sum = sum + v.w;
          }
        }
        res_r[pi] = sum;
      }
      __syncthreads();
    }
  }

  // ---- C
  if (tid < RR_QUERIES && q0 + tid < nq) {
    const int q = q0 + tid;
    Top2 t;
    top2_init(t);
#pragma unroll
    for (int c = 0; c < NCAND; c++) {
      const uint32_t ci = cand[tid * NCAND + c];
      if (ci != SENT) {
        const float rsq = res_r[tid * NCAND + c];
        top2_add(t, sqrtf(rsq), rsq, (int32_t)ci);
      }
    }
    const float* qn = dir == 0 ? P.xn : P.yn;
    const float* on = dir == 0 ? P.yn : P.xn;
    const int opad = dir == 0 ? P.nb_pad : P.na_pad;
    const uint32_t bound = bnd[tid];
    bool safe = no <= NCAND || bound == SENT;
    if (!safe) {
      const uint32_t mask = dir == 0 ? ROW_MASK : COL_MASK;
      safe = (double)t.r2 <= excluded_lower_bound(bound, mask, qn[q], on[opad]);
    }
    const bool accept = (double)t.d1 < ratio * (double)t.d2;   // brute_force_matcher.cc:35
    int32_t* best = dir == 0 ? P.fwd_best : P.rev_best;
    float* dist = dir == 0 ? P.fwd_dist : P.rev_dist;
    best[q] = accept ? t.i1 : -1;
    dist[q] = t.d1;
    if (!safe) unsafe_q[atomicAdd(&n_unsafe, 1)] = q;
  }
  __syncthreads();

  // ---- D
  const int nu = n_unsafe;
  if (nu == 0) return;
  if (tid == 0) atomicAdd(rescanned, (unsigned long long)nu);
  float* qrow = sq;                                        // 128 floats
  Top2* part = reinterpret_cast<Top2*>(sq + 128);          // 256 partial states
  for (int e = 0; e < nu; e++) {
    const int q = unsafe_q[e];
    __syncthreads();
    if (tid < DIM) qrow[tid] = qbase[(size_t)q * DIM + tid];
    __syncthreads();
    Top2 t;
    top2_init(t);
    for (int jj = tid; jj < no; jj += 256) {
      const float rsq = ref_sumsq(qrow, obase + (size_t)jj * DIM);
      top2_add(t, sqrtf(rsq), rsq, jj);
    }
    part[tid] = t;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
      if (tid < s) {
        Top2 a = part[tid];
        top2_merge(a, part[tid + s]);
        part[tid] = a;
      }
      __syncthreads();
    }
    if (tid == 0) {
      const Top2 r = part[0];
      const bool accept = (double)r.d1 < ratio * (double)r.d2;
      int32_t* best = dir == 0 ? P.fwd_best : P.rev_best;
      float* dist = dir == 0 ? P.fwd_dist : P.rev_dist;
      best[q] = accept ? r.i1 : -1;
      dist[q] = r.d1;
    }
  }
}

// Cross-check (brute_force_matcher.cc:64-80) and ordered compaction.  A forward
// match (i -> j) is kept when the reverse direction matched j back to i, or
// when j has no surviving reverse match and i == 0 (the reference's map_[j]
// default-inserts 0).  One workgroup of 1024 threads per pair; every thread owns
// eight consecutive rows per sweep, so output order is ascending row order.
__global__ __launch_bounds__(1024) void k_cross_compact(const PairDev* __restrict__ pairs, int cross_check,
                                                        int32_t* __restrict__ out_counts,
                                                        int32_t* __restrict__ out_lhs,
                                                        int32_t* __restrict__ out_rhs,
                                                        float* __restrict__ out_dist, int64_t out_stride) {
  __shared__ int wsum[16];
  __shared__ int base_s;
  const PairDev& P = pairs[blockIdx.x];
  const int tid = threadIdx.x, l = tid & 63, wv = tid >> 6;
  const size_t ob = (size_t)P.out_slot * (size_t)out_stride;
  if (tid == 0) base_s = 0;
  __syncthreads();
  for (int i0 = 0; i0 < P.na; i0 += 8192) {
    const int ib = i0 + tid * 8;     // fwd arrays are padded to a multiple of 128 rows: vector loads stay in bounds
    int j[8];
    float d[8];
    bool keep[8];
    if (ib < P.na) {
      const int4 a = *reinterpret_cast<const int4*>(P.fwd_best + ib);
      const int4 b = *reinterpret_cast<const int4*>(P.fwd_best + ib + 4);
      const float4 da = *reinterpret_cast<const float4*>(P.fwd_dist + ib);
      const float4 db = *reinterpret_cast<const float4*>(P.fwd_dist + ib + 4);
      j[0] = a.x; j[1] = a.y; j[2] = a.z; j[3] = a.w; j[4] = b.x; j[5] = b.y; j[6] = b.z; j[7] = b.w;
      d[0] = da.x; d[1] = da.y; d[2] = da.z; d[3] = da.w; d[4] = db.x; d[5] = db.y; d[6] = db.z; d[7] = db.w;
    }
    int rbv[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const int i = ib + k;
      keep[k] = ib < P.na && i < P.na && j[k] >= 0;
      rbv[k] = (keep[k] && cross_check) ? P.rev_best[j[k]] : 0;
    }
    int cnt = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const int i = ib + k;
      if (keep[k] && cross_check) keep[k] = rbv[k] == i || (rbv[k] < 0 && i == 0);
      cnt += keep[k] ? 1 : 0;
    }
    // exclusive scan of cnt over the workgroup
    int incl = cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int v = __shfl_up(incl, o);
      if (l >= o) incl += v;
    }
    if (l == 63) wsum[wv] = incl;
    __syncthreads();
    int before = base_s;
    for (int k = 0; k < wv; k++) before += wsum[k];
    int pos = before + incl - cnt;
#pragma unroll
    for (int k = 0; k < 8; k++) {
      if (keep[k]) {
        const size_t o = ob + (size_t)pos++;
        out_lhs[o] = ib + k;
        out_rhs[o] = j[k];
        out_dist[o] = d[k];
      }
    }
    __syncthreads();
    if (tid == 0) {
      int s = base_s;
      for (int k = 0; k < 16; k++) s += wsum[k];
      base_s = s;
    }
    __syncthreads();
  }
  if (tid == 0) out_counts[P.out_slot] = base_s;
}

}  // namespace sfm
