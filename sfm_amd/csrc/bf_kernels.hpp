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

struct QueueEntry { int32_t pair; int32_t dir_q; };  // dir_q = dir<<30 | query

// DIR 0: queries are rows of X (lists from rowpart), candidates are rows of Y.
// DIR 1: queries are rows of Y (lists from colpart), candidates are rows of X.
// Four lanes per query, one candidate each.
template <int DIR>
__global__ __launch_bounds__(256) void k_rerank(const PairDev* __restrict__ pairs, double ratio,
                                                QueueEntry* __restrict__ queue, uint32_t* __restrict__ qcount) {
  const PairDev& P = pairs[blockIdx.y];
  const int nq = DIR == 0 ? P.na : P.nb;
  const int no = DIR == 0 ? P.nb : P.na;
  const int q = blockIdx.x * 64 + (threadIdx.x >> 2);
  const int cl = threadIdx.x & 3;
  if (q >= nq) return;   // whole groups of four leave together

  uint32_t v0 = SENT, v1 = SENT, v2 = SENT, v3 = SENT, v4 = SENT;
  uint32_t i0 = 0, i1 = 0, i2 = 0, i3 = 0, i4 = 0;
  uint32_t bound = SENT;
  auto put = [&](uint32_t key, uint32_t idx) {
    if (key < v4) {
      v4 = key; i4 = idx;
      if (v4 < v3) { uint32_t tv = v3; v3 = v4; v4 = tv; uint32_t ti = i3; i3 = i4; i4 = ti; }
      if (v3 < v2) { uint32_t tv = v2; v2 = v3; v3 = tv; uint32_t ti = i2; i2 = i3; i3 = ti; }
      if (v2 < v1) { uint32_t tv = v1; v1 = v2; v2 = tv; uint32_t ti = i1; i1 = i2; i2 = ti; }
      if (v1 < v0) { uint32_t tv = v0; v0 = v1; v1 = tv; uint32_t ti = i0; i0 = i1; i1 = ti; }
    }
  };
  if (DIR == 0) {
    const size_t st = (size_t)P.na_pad;
    for (int c = 0; c < P.nchunks; c++) {
      const uint32_t* src = P.rowpart + (size_t)c * ROWPART_FIELDS * st + q;
#pragma unroll
      for (int e = 0; e < NCAND; e++) put(src[e * st], src[(NCAND + e) * st]);
      const uint32_t b = src[2 * NCAND * st];
      bound = bound < b ? bound : b;
    }
  } else {
    const size_t st = (size_t)P.nb_pad;
    const int nrb = P.na_pad / BM;
    for (int rb = 0; rb < nrb; rb++) {
      const uint32_t* src = P.colpart + (size_t)rb * 3 * st + q;
#pragma unroll
      for (int k = 0; k < 3; k++) {
        const uint32_t key = src[k * st];
        const uint32_t lo = key & ~COL_MASK;
        const uint32_t reg = lo & 15u, hh = (lo >> 4) & 1u, ww = lo >> 5;
        const uint32_t row = (uint32_t)rb * BM + 32u * ww + (reg & 3u) + 8u * (reg >> 2) + 4u * hh;
        put(key, row);
        if (k == 2) bound = bound < key ? bound : key;
      }
    }
  }
  bound = bound < v4 ? bound : v4;

  const uint32_t ck = cl == 0 ? v0 : cl == 1 ? v1 : cl == 2 ? v2 : v3;
  const uint32_t ci = cl == 0 ? i0 : cl == 1 ? i1 : cl == 2 ? i2 : i3;
  const float* qbase = DIR == 0 ? P.x : P.y;
  const float* obase = DIR == 0 ? P.y : P.x;
  Top2 t;
  top2_init(t);
  if (ck != SENT && (int)ci < no) {
    float rsq = ref_sumsq(qbase + (size_t)q * DIM, obase + (size_t)ci * DIM);
    top2_add(t, sqrtf(rsq), rsq, (int32_t)ci);
  }
#pragma unroll
  for (int o = 1; o < 4; o <<= 1) {
    Top2 u;
    u.d1 = __shfl_xor(t.d1, o); u.i1 = __shfl_xor(t.i1, o); u.d2 = __shfl_xor(t.d2, o);
    u.r1 = __shfl_xor(t.r1, o); u.r2 = __shfl_xor(t.r2, o);
    top2_merge(t, u);
  }
  if (cl == 0) {
    const float* qn = DIR == 0 ? P.xn : P.yn;
    const float* on = DIR == 0 ? P.yn : P.xn;
    const int opad = DIR == 0 ? P.nb_pad : P.na_pad;
    bool safe = no <= NCAND || bound == SENT;
    if (!safe) {
      const uint32_t mask = DIR == 0 ? ROW_MASK : COL_MASK;
      safe = (double)t.r2 <= excluded_lower_bound(bound, mask, qn[q], on[opad]);
    }
    const bool accept = (double)t.d1 < ratio * (double)t.d2;   // brute_force_matcher.cc:35
    int32_t* best = DIR == 0 ? P.fwd_best : P.rev_best;
    float* dist = DIR == 0 ? P.fwd_dist : P.rev_dist;
    best[q] = accept ? t.i1 : -1;
    dist[q] = t.d1;
    if (!safe) {
      uint32_t pos = atomicAdd(qcount, 1u);
      queue[pos].pair = blockIdx.y;
      queue[pos].dir_q = (DIR << 30) | q;
    }
  }
}

// Exact full scan for queued queries: one workgroup per query, threads take
// columns j = tid, tid+256, ...; partial states merge order-independently.
__global__ __launch_bounds__(256) void k_rescan(const PairDev* __restrict__ pairs, double ratio,
                                                const QueueEntry* __restrict__ queue,
                                                const uint32_t* __restrict__ qcount) {
  __shared__ float qrow[DIM];
  __shared__ Top2 part[256];
  const uint32_t n = *qcount;
  for (uint32_t e = blockIdx.x; e < n; e += gridDim.x) {
    const QueueEntry qe = queue[e];
    const PairDev& P = pairs[qe.pair];
    const int dir = (qe.dir_q >> 30) & 1;
    const int q = qe.dir_q & 0x3FFFFFFF;
    const int no = dir == 0 ? P.nb : P.na;
    const float* qbase = dir == 0 ? P.x : P.y;
    const float* obase = dir == 0 ? P.y : P.x;
    __syncthreads();
    if (threadIdx.x < DIM) qrow[threadIdx.x] = qbase[(size_t)q * DIM + threadIdx.x];
    __syncthreads();
    Top2 t;
    top2_init(t);
    for (int j = threadIdx.x; j < no; j += 256) {
      float rsq = ref_sumsq(qrow, obase + (size_t)j * DIM);
      top2_add(t, sqrtf(rsq), rsq, j);
    }
    part[threadIdx.x] = t;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
      if ((int)threadIdx.x < s) {
        Top2 a = part[threadIdx.x];
        top2_merge(a, part[threadIdx.x + s]);
        part[threadIdx.x] = a;
      }
      __syncthreads();
    }
    if (threadIdx.x == 0) {
      Top2 r = part[0];
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
// default-inserts 0).  One workgroup of 1024 threads per pair.
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
  for (int i0 = 0; i0 < P.na; i0 += 1024) {
    const int i = i0 + tid;
    int j = -1;
    float d = 0.f;
    bool keep = false;
    if (i < P.na) {
      j = P.fwd_best[i];
      d = P.fwd_dist[i];
      keep = j >= 0;
      if (keep && cross_check) {
        const int rb = P.rev_best[j];
        keep = rb == i || (rb < 0 && i == 0);
      }
    }
    const unsigned long long m = __ballot(keep);
    const int within = __popcll(m & ((1ull << l) - 1ull));
    if (l == 0) wsum[wv] = __popcll(m);
    __syncthreads();
    int before = base_s;
    for (int k = 0; k < wv; k++) before += wsum[k];
    if (keep) {
      const size_t o = ob + (size_t)(before + within);
      out_lhs[o] = i;
      out_rhs[o] = j;
      out_dist[o] = d;
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
